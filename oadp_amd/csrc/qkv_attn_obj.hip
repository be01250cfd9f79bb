// qkv_attn_obj.hip — the fused ln_1 + in_proj + attention kernel on a 208-row tile, two forms of one kernel template:
//   * objects mode (QUAD = false): LayerNorm-folded QKV in-projection + self-attention of ONE crop's 197 tokens AND its
//     object token as one persistent kernel for gfx950 (192 < L + 1 <= 200 rows per crop) — described first;
//   * QUAD = true: FOUR images of L <= 50 tokens per tile, plain self-attention per image (encode_image at 224^2 / patch 32,
//     blocks mode [REF oadp/oake/globals.py:57; oadp/oake/blocks.py:129]): the same K loop and tile-end phases, 16 tasks
//     (image, query tile) instead of 13 query tiles, no object token.  Batch 256 x 12 heads = 768 tiles = three exact rounds
//     of 256 CUs (qkv_attn.hip's three-image 160-row tile: 1032 = 4.03).  See "QUAD mode" below.
//
//   reference ops  Hooks.residual_attention_block_forward_pre + the block's own attention
//                  [REF oadp/oake/objects.py:223-247: y attends over ln_1(cat([x[1:], y])) with the -100 * mask bias;
//                  x over ln_1(x)]; SURVEY.md §8 A11, A15c-e
//
// What it replaces per layer and pass of 128 crops: gemm_pp_kernel<EPI_T16_BIAS_LN> over the T + n rows (qkv [T + n, 3C]:
// 116 MB written) + attention_head_kernel (reads it back; 42.8 us, VALU-issue-bound) = 131.5 us.  Here a tile is
// (crop, head): BM = 208 rows = the crop's 197 token rows + its object-token row (gathered from row T + crop of the same
// matrix) + 10 padding rows, BN = 192 = the head's q | k | v.  128 crops x 12 heads = 1536 tiles = exactly six rounds of
// 256 CUs.
//   * K loop: gemm_pp_kernel's four-phase schedule at 208 x 192 (13 MFMA row tiles: row group 0 takes 7, group 1 six; wave
//     tile 112 | 96 x 48; one fragment set: 84 + 40 registers), 4 LDS-DMA waves, 3-slot ring of 51 200-byte stages.  The
//     folded weight rows are permuted so that every column wave owns 16 q, 16 k and 16 v columns of the head
//     (launch_permute_qkv_obj): all eight compute waves then hold one V tile column each.
//   * Tile end: q and k -> LDS (the ring slot the tile's last K-tile has left: two regions of (L + 1) x 128-byte rows,
//     50 688 of 51 200 bytes); v stays in registers, packed.  S phase: 13 query tiles over the twelve waves (one 16-query
//     task each; compute wave 4 takes the 13th tile IN THE SAME instruction stream as its own: one set of K / V
//     fragment reads feeds both): S^T = K Q^T for all 13 key tiles in registers, one-pass softmax
//     (attention_head_kernel's), the object token's query column with its own key rules (not the CLS key; -100 * mask on
//     the patch keys; itself).  Then v -> LDS over the q region, PV through the transpose read, O out as half lines.
//   * Five barriers per tile end; nothing pending across the K loop.
//
// Measured (tools/qkv_attn_trace.py 128 197, profiles/r05/qkv_attn_obj_trace_v3b.txt): a tile takes 37-39 k cycles: K loop
// 23-25 k (13 K-tiles x ~1.85 k; 1.34 k is the MFMA time), q | k write 2.0 k, S phase 6.0 k, v write 1.0 k, PV 4.5 k.
// The S and PV phases are bound by the SIMDs' issue slots and the LDS port, not by one wave's dependent chain: three
// tasks on a SIMD take 4.4 k cycles together (the oldest wave is served first and reports 3.2 k), the SIMD with
// the 13th tile 5.9 k.  Splitting the 13th tile over all twelve waves by key tile (every wave + 1 / 13 of a task; built:
// profiles/r05/qkv_attn_obj_v4_split_13th_tile.hip.txt, parity-green) is no faster (40.3 k per tile: 13 partial tasks cost
// more than one whole) — kept on compute wave 4.  Objects step (A/B, three interleaved rounds,
// profiles/r05/ab_fuse_qkv_attn_objects_v3b.log): 92.4 -> 93.4 images/s on two lanes, 83.9 -> 87.9 on one.
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

constexpr int BK = 64;
constexpr int kRowBytes = BK * 2;
constexpr int LBM = 208, LBN = 192;
constexpr int kLStage = (LBM + LBN) * kRowBytes;  // 51 200
constexpr int kLNStage = 3;
constexpr int kLEpi = kLNStage * kLStage;         // 153 600: bias[192] | colsum[192] | rowstat[208] | key bias[208]
constexpr int kLBias = kLEpi, kLColsum = kLEpi + 1024, kLRowstat = kLEpi + 2048, kLMbias = kLEpi + 2048 + LBM * 8;
constexpr int kLWalkTab = kLMbias + LBM * 4;      // 158 144 (the key bias is 16-bit: half of its slot)
constexpr int kLWalkN = 256;                      // this block's first 256 tiles, decoded once at entry: (group << 8) | head
constexpr int kLLdsBytes = kLWalkTab + kLWalkN * 4;  // 159 168
constexpr int kLNKT = 13;                         // key / query tiles of 16
constexpr int kRowParts = 16;
constexpr float kLog2e = 1.4426950408889634f;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct QkvAttnObjParams {
  const float* bias;      // [H * 192] folded bias in the kernel's column order
  const float* colsum;    // [H * 192]
  const float2* rowpart;  // [T + n, 16]
  int nparts;
  float inv_k;
  int n_img, L, H, T;     // crops, tokens per crop, heads, T = n_img * L = first object-token row
  const void* mask;       // [n_img, L - 1], 1 = background
  int mask_f16;
  int walk_hq, walk_gq;   // tile walk (walk_decode): head blocks of walk_hq heads x group blocks of walk_gq groups; 0 = group-major
  unsigned long long* trace;
};

// Tile walk.  A tile is (group, head) — group = a crop (objects) or four images (QUAD).  The flat order t -> (group, head)
// decides what an XCD's 32 co-resident blocks stream through its 4 MiB L2 (XCD x owns a contiguous range of t, block
// b / 8 of it takes every 32nd tile): group-major (t = group * H + head) puts 2.67 groups x ALL heads side by side, i.e.
// the whole folded in-projection (3 C^2 = 3.5 MB) passes every XCD's L2 once per round of 32 tiles — measured reads
// 104 MB per launch for 23 MB algorithmic (globals: 8 XCDs x 3 rounds x 3.5 MB + x).  Blocked: group blocks of GQ groups
// (outer), inside them head blocks of HQ heads, inside those (group, head) — one round of an XCD is then GQ groups x HQ
// heads: HQ x 295 KB of W beside GQ x (rows x 1.5 KB) of x, and the x rows of a group block stay in the L2 while its H / HQ
// head blocks pass.  HQ x GQ = 32 = the blocks per XCD at 256 CUs.  [REF oadp/oake/globals.py:57; objects.py:223-247: the
// reference's attention has no such order; this is placement only, the results do not depend on it]
__device__ __forceinline__ void walk_decode(int t, int G, int H, int HQ, int GQ, int& grp, int& head) {
  if (HQ <= 0 || H % HQ != 0) {
    grp = t / H;
    head = t - grp * H;
    return;
  }
  const int per_gb = GQ * H;
  const int gb = t / per_gb;
  int r = t - gb * per_gb;
  const int left = G - gb * GQ;
  const int gl = left < GQ ? left : GQ;  // groups of this block (the last one may be short)
  const int per_hb = gl * HQ;
  const int hb = r / per_hb;
  r -= hb * per_hb;
  const int g = r / HQ;
  grp = gb * GQ + g;
  head = hb * HQ + (r - g * HQ);
}

template <typename T>
struct ObjTask {
  typename T16<T>::vec8 pf[7];  // P of the task's 16 queries over 13 key tiles, packed as PV operand fragments
  float inv;
};

// scores + softmax of query tile q0 .. q0 + 15 (tiles 0 .. 11: token queries only) against all keys; NQ = 2: the wave
// also takes the 13th tile (rows 192 .. 207: the last tokens, the object token's row L, padding) in the SAME instruction
// stream — a task is latency-bound (LDS read -> MFMA -> exp chains of one wave), so the second tile rides on the first
// one's K fragments for a few hundred cycles instead of doubling the phase
template <typename T, int NQ>
__device__ __forceinline__ void obj_task_s(ObjTask<T> (&st)[NQ], const char* qs, const char* ks, const char* mbias,
                                           int q0, int L, int tid_) {
  typedef typename T16<T>::vec8 vec8;
  int atid = tid_;
  asm volatile("" : "+v"(atid));
  const int fr = atid & 15, g = (atid & 63) >> 4;
  constexpr int kObjQ0 = 16 * (kLNKT - 1);
  vec8 qf[NQ][2];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int row = (t == 0 ? q0 : kObjQ0) + fr;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[t][kk] = *reinterpret_cast<const vec8*>(qs + row * kRowBytes + (((kk * 4 + g) ^ sw) << 4));
  }
  const int ksw = (fr >> 1) & 7;
  f32x4 sacc[NQ][kLNKT];
#pragma unroll
  for (int kt = 0; kt < kLNKT; ++kt) {
    const vec8 kf0 = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * kRowBytes + ((g ^ ksw) << 4));
    const vec8 kf1 = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * kRowBytes + (((4 + g) ^ ksw) << 4));
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      sacc[t][kt] = T16<T>::mfma(kf0, qf[t][0], f32x4{0.f, 0.f, 0.f, 0.f});
      sacc[t][kt] = T16<T>::mfma(kf1, qf[t][1], sacc[t][kt]);
    }
    if (kt & 1) __builtin_amdgcn_sched_barrier(0);  // (at most two tiles' K fragments in flight)
  }
  // keys that only pad: the last key tile alone has any.  A token's query takes keys 0 .. L - 1; the object token's
  // query (row L) also its own key L
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int lim = t == 1 && kObjQ0 + fr == L ? L + 1 : L;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sacc[t][kLNKT - 1][r] = 16 * (kLNKT - 1) + 4 * g + r < lim ? sacc[t][kLNKT - 1][r] : -1e30f;
  }
  if (NQ == 2) {
    // the object token's key rules [REF oadp/oake/objects.py:206-213,232-247] — not the CLS key, -100 * mask on the patch
    // keys — as ONE more k step of the score product: A = the key's bias in k = 0 (mbias: 16-bit, exact: 0 / -100 /
    // -60000 = "not a key"), B = 1 in k = 0 of the object token's query column, 0 elsewhere.  13 MFMAs instead of
    // 52 x (compare, add, two selects) per lane.
    vec8 one;
#pragma unroll
    for (int j = 0; j < 8; ++j) one[j] = to16<T>(0.f);
    if (g == 0 && kObjQ0 + fr == L) one[0] = to16<T>(1.f);
    // mbias is [key % 16][key / 16]: the lane's 13 values in two 16-byte reads, no branch
    const vec8 bt0 = *reinterpret_cast<const vec8*>(mbias + fr * 32);
    const vec8 bt1 = *reinterpret_cast<const vec8*>(mbias + fr * 32 + 16);
#pragma unroll
    for (int kt = 0; kt < kLNKT; ++kt) {
      vec8 bk;
#pragma unroll
      for (int j = 0; j < 8; ++j) bk[j] = to16<T>(0.f);
      const T b = kt < 8 ? bt0[kt & 7] : bt1[kt & 7];
      bk[0] = g == 0 ? b : to16<T>(0.f);
      sacc[NQ - 1][kt] = T16<T>::mfma(bk, one, sacc[NQ - 1][kt]);
    }
  }
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < kLNKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[t][kt][r]);
    mx = rows16_max(mx);
    // (two scores per instruction: v_pk_fma_f32 / v_pk_add_f32 — the phase is bound by the SIMDs' issue slots)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float nb = -mx * kLog2e;
    const f32x2 nb2 = f32x2{nb, nb}, l2 = f32x2{kLog2e, kLog2e};
    f32x2 sum2 = f32x2{0.f, 0.f};
#pragma unroll
    for (int ksx = 0; ksx < 7; ++ksx) {
      vec8 p8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kt = 2 * ksx + h;
        if (kt < kLNKT) {
          const f32x2 a = __builtin_elementwise_fma(f32x2{sacc[t][kt][0], sacc[t][kt][1]}, l2, nb2);
          const f32x2 b = __builtin_elementwise_fma(f32x2{sacc[t][kt][2], sacc[t][kt][3]}, l2, nb2);
          const f32x2 ea = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          const f32x2 eb = f32x2{__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
          sum2 += ea;
          sum2 += eb;
          p8[4 * h + 0] = to16<T>(ea[0]);
          p8[4 * h + 1] = to16<T>(ea[1]);
          p8[4 * h + 2] = to16<T>(eb[0]);
          p8[4 * h + 3] = to16<T>(eb[1]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) p8[4 * h + j] = to16<T>(0.f);
        }
      }
      st[t].pf[ksx] = p8;
    }
    const float sum = sum2[0] + sum2[1];
    st[t].inv = __builtin_amdgcn_rcpf(rows16_sum(sum));
  }
}

// O^T = V^T P^T of the task's tile(s), out as half lines (attention_head_kernel's output path)
template <typename T, int NQ>
__device__ __forceinline__ void obj_task_pv(const ObjTask<T> (&st)[NQ], const char* vs, int q0, int L, char* out_rows,
                                            char* out_y, int C, int tid_) {
  typedef typename T16<T>::vec8 vec8;
  typedef s16x4 __attribute__((address_space(3))) * lds4_t;
  int atid = tid_;
  asm volatile("" : "+v"(atid));
  const int fr = atid & 15, g = (atid & 63) >> 4;
  constexpr int kObjQ0 = 16 * (kLNKT - 1);
  f32x4 oacc[NQ][4];
#pragma unroll
  for (int t = 0; t < NQ; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int row0 = 4 * g + (fr >> 2);  // + 32 ks (+ 16): the same swizzle
  const int vsw = (row0 >> 1) & 7, c4 = (fr & 3) * 4;
#pragma unroll
  for (int ksx = 0; ksx < 7; ++ksx) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const char* p0 = vs + (row0 + 32 * ksx) * kRowBytes + (((dt * 2 + (c4 >> 3)) ^ vsw) << 4) + (c4 & 4) * 2;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
      s16x4 hi = s16x4{0, 0, 0, 0};  // (the 13th key tile has no partner: P is 0 there)
      if (2 * ksx + 1 < kLNKT) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 16 * kRowBytes));
      s16x8 both;
      both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
      both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
#pragma unroll
      for (int t = 0; t < NQ; ++t) oacc[t][dt] = T16<T>::mfma(__builtin_bit_cast(vec8, both), st[t].pf[ksx], oacc[t][dt]);
    }
    __builtin_amdgcn_sched_barrier(0);  // (one key step's V fragments in flight)
  }
  // lane (fr, g) holds d = 16 dt + 4 g + i of query row fr: permlane16_swap(tile a, tile b) gives every lane 8
  // consecutive d of tile (g & 1 ? b : a) from column 8 (g >> 1) on (attention_head.inc)
  const unsigned ooff = (unsigned)(16 * (g >> 1) + 32 * (g & 1));
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int q = (t == 0 ? q0 : kObjQ0) + fr;
    char* dst = q < L ? out_rows + (size_t)q * (size_t)(2 * C) : out_y;
    const float inv = st[t].inv;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4 oa = oacc[t][2 * half], ob = oacc[t][2 * half + 1];
      const uint2 xa = pack4<T>(oa[0] * inv, oa[1] * inv, oa[2] * inv, oa[3] * inv);
      const uint2 xb = pack4<T>(ob[0] * inv, ob[1] * inv, ob[2] * inv, ob[3] * inv);
      const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(xa.x, xb.x, false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(xa.y, xb.y, false, false);
      if (q <= L) store16_policy<1>(dst + ooff + 64u * half, u32x4_t{s0[0], s1[0], s0[1], s1[1]});
    }
  }
}

// ---- QUAD mode: four images of L <= 50 tokens per tile (rows 0 .. 4 L - 1), plain self-attention per image ----
// 16 tasks: task tk = (image tk >> 2 of the tile, query tile tk & 3) against the image's <= 64 keys (4 key tiles, the
// padded keys masked).  Wave w takes task w, the DMA waves (idle through the q | k and v writes) also task w + 4: four
// tasks on every SIMD.
template <typename T>
struct QuadTask {
  typename T16<T>::vec8 pf[2];
  float inv;
};

// NT tasks tk, tk + 4, .. in ONE instruction stream (a task this small is one wave's dependent chain LDS read -> MFMA ->
// row maximum -> exp -> row sum: two of them interleaved take little longer than one)
template <typename T, int NT>
__device__ __forceinline__ void quad_task_s(QuadTask<T> (&st)[NT], const char* qs, const char* ks, int tk, int L, int tid_) {
  typedef typename T16<T>::vec8 vec8;
  int atid = tid_;
  asm volatile("" : "+v"(atid));
  const int fr = atid & 15, g = (atid & 63) >> 4;
  vec8 qf[NT][2];
  f32x4 sacc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r0 = ((tk + 4 * t) >> 2) * L;  // the image's first row of the tile
    const int row = r0 + 16 * (tk & 3) + fr;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[t][kk] = *reinterpret_cast<const vec8*>(qs + row * kRowBytes + (((kk * 4 + g) ^ sw) << 4));
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int r0 = ((tk + 4 * t) >> 2) * L;
      const int ksw = ((r0 + fr) >> 1) & 7;  // (+ 16 kt: the same swizzle)
      const vec8 kf0 = *reinterpret_cast<const vec8*>(ks + (r0 + kt * 16 + fr) * kRowBytes + ((g ^ ksw) << 4));
      const vec8 kf1 = *reinterpret_cast<const vec8*>(ks + (r0 + kt * 16 + fr) * kRowBytes + (((4 + g) ^ ksw) << 4));
      sacc[t][kt] = T16<T>::mfma(kf0, qf[t][0], f32x4{0.f, 0.f, 0.f, 0.f});
      sacc[t][kt] = T16<T>::mfma(kf1, qf[t][1], sacc[t][kt]);
    }
  float mx[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    mx[t] = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if ((kt + 1) * 16 > L) {  // (wave-uniform) a key tile with keys past the image: rows of the next image, or none
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[t][kt][r] = kt * 16 + 4 * g + r < L ? sacc[t][kt][r] : -1e30f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[t] = fmaxf(mx[t], sacc[t][kt][r]);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) mx[t] = rows16_max(mx[t]);
  float sum[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float nb = -mx[t] * kLog2e;
    sum[t] = 0.f;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      vec8 p8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e = __builtin_amdgcn_exp2f(fmaf(sacc[t][2 * ks2 + (j >> 2)][j & 3], kLog2e, nb));
        sum[t] += e;
        p8[j] = to16<T>(e);
      }
      st[t].pf[ks2] = p8;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) st[t].inv = __builtin_amdgcn_rcpf(rows16_sum(sum[t]));
}

// out_tile: the tile's first output row at the head's columns; rows_left: rows of the matrix from the tile's first on
template <typename T, int NT>
__device__ __forceinline__ void quad_task_pv(const QuadTask<T> (&st)[NT], const char* vs, int tk, int L, char* out_tile,
                                             int rows_left, int C, int tid_) {
  typedef typename T16<T>::vec8 vec8;
  typedef s16x4 __attribute__((address_space(3))) * lds4_t;
  int atid = tid_;
  asm volatile("" : "+v"(atid));
  const int fr = atid & 15, g = (atid & 63) >> 4;
  f32x4 oacc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int c4 = (fr & 3) * 4;
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row0 = ((tk + 4 * t) >> 2) * L + 4 * g + (fr >> 2);  // + 32 ks (+ 16): the same swizzle
        const int vsw = (row0 >> 1) & 7;
        const char* p0 = vs + (row0 + 32 * ks2) * kRowBytes + (((dt * 2 + (c4 >> 3)) ^ vsw) << 4) + (c4 & 4) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 16 * kRowBytes));
        s16x8 both;
        both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
        both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
        oacc[t][dt] = T16<T>::mfma(__builtin_bit_cast(vec8, both), st[t].pf[ks2], oacc[t][dt]);
      }
  const int q = 16 * (tk & 3) + fr;
  const unsigned ooff = (unsigned)(16 * (g >> 1) + 32 * (g & 1));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r0 = ((tk + 4 * t) >> 2) * L;
    const bool live = q < L && r0 + q < rows_left;
    char* dst = out_tile + (size_t)(r0 + q) * (size_t)(2 * C);
    const float inv = st[t].inv;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4 oa = oacc[t][2 * half], ob = oacc[t][2 * half + 1];
      const uint2 xa = pack4<T>(oa[0] * inv, oa[1] * inv, oa[2] * inv, oa[3] * inv);
      const uint2 xb = pack4<T>(ob[0] * inv, ob[1] * inv, ob[2] * inv, ob[3] * inv);
      const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(xa.x, xb.x, false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(xa.y, xb.y, false, false);
      if (live) store16_policy<1>(dst + ooff + 64u * half, u32x4_t{s0[0], s1[0], s0[1], s1[1]});
    }
  }
}

#define QO_PIN() __builtin_amdgcn_sched_barrier(0)
#define QO_BAR()                  \
  do {                            \
    QO_PIN();                     \
    __builtin_amdgcn_s_barrier(); \
    QO_PIN();                     \
  } while (0)

#define QO_STAMP(role_, tile_, k_)                                                                            \
  do {                                                                                                        \
    if (p.trace != nullptr && (tid & 63) == 0 && blockIdx.x < 64 && (tile_) < 6)                              \
      p.trace[(((size_t)blockIdx.x * 3 + (role_)) * 6 + (tile_)) * 8 + (k_)] = __builtin_readcyclecounter();  \
  } while (0)

// one compute wave: row group ROW0 .. ROW0 + 16 MI - 1 (group 0: MI = 7; group 1, one phase behind: MI = 6), columns
// 48 wn .. + 47 = (q, k, v) x 16 head-dim columns 16 wn ..
template <typename T, bool QUAD, int MI, int ROW0, bool LATE>
__device__ __forceinline__ void obj_compute_wave(char* smem, int tid, int wid, const QkvAttnObjParams& p, T* out, int C,
                                                 int nk, int my_tiles, int xb, int xslot, int per_xcd) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NI = 3, TN = 48;
  const int lane = tid & 63;
  const int wn = wid & 3;
  const int L = p.L, H = p.H;
  const int region = (QUAD ? 4 * L : L + 1) * kRowBytes;
  const int rows_live = QUAD ? 4 * L : L + 1;  // tile rows that hold tokens
  const int frow = lane & 15, fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int a_base = (ROW0 + frow) * kRowBytes;
  const int b_base = LBM * kRowBytes + (wn * TN + frow) * kRowBytes;
  const int koff0 = ((0 * 4 + fg) ^ fsw) << 4;
  const int koff1 = ((1 * 4 + fg) ^ fsw) << 4;
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  QO_BAR();  // B0
  int c_buf = 0;
  const bool stamp = (wid & 3) == 0;  // waves 0 and 4: measurement (tools/qkv_attn_trace.py)
  for (int ti = 0; ti < my_tiles; ++ti) {
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 0);
    if (LATE) QO_BAR();  // group 1 runs one phase behind group 0
    for (int kt = 0; kt < nk; ++kt) {
      vec8 af[MI], bf[NI];
      const char* st = smem + c_buf * kLStage;
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff0);
#pragma unroll
      for (int i = 0; i < NI; ++i) bf[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      QO_BAR();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][OAKE_NI_AT(mi, ni, NI)] = T16<T>::mfma(bf[OAKE_NI_AT(mi, ni, NI)], af[mi], acc[mi][OAKE_NI_AT(mi, ni, NI)]);
      __builtin_amdgcn_s_setprio(0);
      QO_BAR();
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff1);
#pragma unroll
      for (int i = 0; i < NI; ++i) bf[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff1);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      QO_BAR();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][OAKE_NI_AT(mi, ni, NI)] = T16<T>::mfma(bf[OAKE_NI_AT(mi, ni, NI)], af[mi], acc[mi][OAKE_NI_AT(mi, ni, NI)]);
      __builtin_amdgcn_s_setprio(0);
      const bool last = kt == nk - 1;
      if (!last) c_buf = c_buf == kLNStage - 1 ? 0 : c_buf + 1;
      if (!LATE || !last) QO_BAR();
    }
    // ---------------- tile end ----------------
    // c_buf = the slot of the tile's last K-tile (every wave has read its fragments of it): [Q | K] regions of L + 1 rows
    char* const qs = smem + c_buf * kLStage;
    char* const ks = qs + region;
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 1);
    uint2 vpk[MI];
    {
      QO_PIN();
      int etid = tid;
      asm volatile("" : "+v"(etid));
      const int er = etid & 15, eg = (etid & 63) >> 4;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef const __attribute__((address_space(3))) f32x2* lds_f2_t;
      f32x2 rs[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) rs[mi] = *(lds_f2_t)(smem + kLRowstat + (ROW0 + mi * 16 + er) * 8);
      const int d0 = wn * 16 + 4 * eg;  // head-dim column of the lane's four values (the same for q, k, v)
      const int coff = (d0 & 7) << 1, cch = d0 >> 3;
      // column tile by column tile (q, k, v): one tile's constants in registers at a time (the accumulators are live)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float4 b4 = *reinterpret_cast<const float4*>(smem + kLBias + (wn * TN + ni * 16 + 4 * eg) * 4);
        const float4 c4 = *reinterpret_cast<const float4*>(smem + kLColsum + (wn * TN + ni * 16 + 4 * eg) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int R = ROW0 + mi * 16 + er;
          const f32x2 r = rs[mi];
          const f32x4 a = acc[mi][ni];
          const uint2 pk = pack4<T>(fmaf(a[0], r[0], fmaf(r[1], c4.x, b4.x)), fmaf(a[1], r[0], fmaf(r[1], c4.y, b4.y)),
                                    fmaf(a[2], r[0], fmaf(r[1], c4.z, b4.z)), fmaf(a[3], r[0], fmaf(r[1], c4.w, b4.w)));
          acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ni == 2) {
            vpk[mi] = pk;
          } else if (R < rows_live) {
            *reinterpret_cast<uint2*>(qs + ni * region + R * kRowBytes + ((cch ^ ((R >> 1) & 7)) << 4) + coff) = pk;
          }
        }
      }
    }
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 2);
    QO_BAR();  // X2: q and k of the crop's head are in LDS
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 3);
    constexpr int NQ = LATE ? 2 : 1;  // (row group 1 is compiled for two tiles; only wave 4 takes the 13th)
    ObjTask<T> task[QUAD ? 1 : NQ];
    QuadTask<T> qtask[1];
    if constexpr (QUAD) {
      quad_task_s<T, 1>(qtask, qs, ks, wid, L, tid);
    } else if (LATE && wid == 4) {
      obj_task_s<T, NQ>(task, qs, ks, smem + kLMbias, wid * 16, L, tid);
    } else {
      obj_task_s<T, 1>(reinterpret_cast<ObjTask<T>(&)[1]>(task[0]), qs, ks, smem + kLMbias, wid * 16, L, tid);
    }
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 4);
    QO_BAR();  // X3: every wave is done with q and k
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 5);
    {
      int etid = tid;
      asm volatile("" : "+v"(etid));
      const int er = etid & 15, eg = (etid & 63) >> 4;
      const int d0 = wn * 16 + 4 * eg;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int R = ROW0 + mi * 16 + er;
        if (R < rows_live) *reinterpret_cast<uint2*>(qs + R * kRowBytes + (((d0 >> 3) ^ ((R >> 1) & 7)) << 4) + ((d0 & 7) << 1)) = vpk[mi];
      }
    }
    QO_BAR();  // X4: v is in LDS (over the q region)
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 6);
    {
      // (group, head) of the tile: decoded once per block at entry into an LDS table (three runtime divisions per
      // decode: ~400 cycles in front of every wave's output stores otherwise — what made the blocked walk LOSE 0.6-1 %
      // although it halves the kernel's fabric reads, profiles/r06/ab_qkv_walk_*.log)
      int img, head;
      if (ti < kLWalkN) {
        const int pk = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + kLWalkTab + ti * 4));
        img = pk >> 8;
        head = pk & 255;
      } else {
        walk_decode(xb + xslot + ti * per_xcd, QUAD ? (p.n_img + 3) / 4 : p.n_img, H, p.walk_hq, p.walk_gq, img, head);
      }
      if constexpr (QUAD) {
        const int row_first = img * 4 * L;  // (img: the tile's group of four images)
        quad_task_pv<T, 1>(qtask, qs, wid, L, reinterpret_cast<char*>(out + (size_t)row_first * C + head * kHeadDim),
                        p.T - row_first, C, tid);
      } else {
        char* orow = reinterpret_cast<char*>(out + (size_t)img * L * C + head * kHeadDim);
        char* oy = reinterpret_cast<char*>(out + (size_t)(p.T + img) * C + head * kHeadDim);
        if (LATE && wid == 4) {
          obj_task_pv<T, NQ>(task, qs, wid * 16, L, orow, oy, C, tid);
        } else {
          obj_task_pv<T, 1>(reinterpret_cast<const ObjTask<T>(&)[1]>(task[0]), qs, wid * 16, L, orow, oy, C, tid);
        }
      }
    }
    QO_BAR();  // X5: the slot goes back to the ring
    if (stamp) QO_STAMP(LATE ? 1 : 0, ti, 7);
    c_buf = c_buf == kLNStage - 1 ? 0 : c_buf + 1;
  }
}

template <typename T, bool QUAD>
__global__ __launch_bounds__(768) void qkv_attn_obj_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                           T* __restrict__ out, int K, QkvAttnObjParams p) {
  constexpr int NW = 8, NL = 4;
  constexpr int kPieces = (LBM + LBN) / 8;  // 50 pieces of 1 KiB per K-tile: 26 of A rows, 24 of W rows
  constexpr int kAPieces = LBM / 8;
  constexpr int NPLMAX = 13;                // DMA waves 0 / 1 issue 13 pieces per K-tile, waves 2 / 3 twelve
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L, H = p.H;
  const int C = H * kHeadDim;
  const int nx = 8;
  const int ntiles = (QUAD ? (p.n_img + 3) / 4 : p.n_img) * H;
  const int xcd = blockIdx.x % nx, xslot = blockIdx.x / nx;
  const int per_xcd = gridDim.x / nx;
  const int q_ = ntiles / nx, r_ = ntiles % nx;
  const int xb = xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
  const int xc = xcd < r_ ? q_ + 1 : q_;
  const int my_tiles = xslot < xc ? (xc - xslot + per_xcd - 1) / per_xcd : 0;
  if (my_tiles == 0) return;
  const int nk = K / BK;
  const int total = my_tiles * nk;
  if (wid == 0) {  // this block's tile list, decoded by the lanes of one wave (visible to every wave behind barrier B0)
    for (int i = lane; i < (my_tiles < kLWalkN ? my_tiles : kLWalkN); i += 64) {
      int g_, h_;
      walk_decode(xb + xslot + i * per_xcd, QUAD ? (p.n_img + 3) / 4 : p.n_img, H, p.walk_hq, p.walk_gq, g_, h_);
      *reinterpret_cast<int*>(smem + kLWalkTab + i * 4) = (g_ << 8) | h_;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the table is written before this wave arrives at B0
  }

  if (wid >= NW) {
    // ================= DMA wave =================
    const int lw = wid - NW;
    const int npl = lw < 2 ? 13 : 12;
    const int region = (QUAD ? 4 * L : L + 1) * kRowBytes;
    // per-lane byte offsets of the wave's pieces from the (wave-uniform) A / W base: 13 registers, not 13 address pairs
    unsigned src[NPLMAX];
    const char* const a_bytes = reinterpret_cast<const char*>(A);
    const char* const w_bytes = reinterpret_cast<const char*>(W);
    auto row_of = [&](int img, int rr) {  // global row of tile row rr: the crop's tokens, then its object token (and padding)
      if (QUAD) {  // (img: the group of four images; rows past the matrix: any row, their results are not stored)
        const int m = img * 4 * L + rr;
        return m < p.T ? m : p.T - 1;
      }
      return rr < L ? img * L + rr : p.T + img;
    };
    // (group, head) of the tile being staged and of its predecessor — the consumer position is at most one tile behind the
    // producer cursor — so the decode (three runtime divisions) runs once per tile in this wave, not five times
    int dec_tile = -1, dec_img = 0, dec_head = 0, prev_img = 0, prev_head = 0;
    auto set_src = [&](int tile_i) {
      int img, head;
      walk_decode(xb + xslot + tile_i * per_xcd, QUAD ? (p.n_img + 3) / 4 : p.n_img, H, p.walk_hq, p.walk_gq, img, head);
      prev_img = dec_img; prev_head = dec_head;
      dec_tile = tile_i; dec_img = img; dec_head = head;
#pragma unroll
      for (int j = 0; j < NPLMAX; ++j) {
        int ii = lw + NL * j;
        ii = ii < kPieces ? ii : lw;  // (the 13th piece of waves 2 / 3 does not exist: never issued)
        const int rr = 8 * ii + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rr >> 1) & 7);
        if (ii < kAPieces)
          src[j] = (unsigned)row_of(img, rr) * (unsigned)(K * 2) + chunk * 16;
        else
          src[j] = (unsigned)(head * LBN + rr - LBM) * (unsigned)(K * 2) + chunk * 16;
      }
    };
    int s_g = 0, s_kt = 0, s_tile = 0, s_buf = 0;
    int d_kt = 0, d_tile = 0;
#define QO_STAGE(j0_, j1_)                                                                                  \
  do {                                                                                                      \
    if (s_g < total) {                                                                                      \
      char* _base = smem + s_buf * kLStage;                                                                 \
      const unsigned _koff = (unsigned)s_kt * (BK * 2);                                                     \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j) if (_j < npl)                                \
          __builtin_amdgcn_global_load_lds(                                                                 \
              (gbl_ptr_t)((lw + NL * _j < kAPieces ? a_bytes : w_bytes) + (size_t)(src[_j] + _koff)),       \
              (lds_ptr_t)(_base + (lw + NL * _j) * 1024), 16, 0, 0);                                        \
    }                                                                                                       \
  } while (0)
#define QO_ADVANCE()                                  \
  do {                                                \
    if (s_g < total) {                                \
      ++s_g;                                          \
      s_buf = s_buf == kLNStage - 1 ? 0 : s_buf + 1;  \
      if (++s_kt == nk) {                             \
        s_kt = 0;                                     \
        ++s_tile;                                     \
        if (s_tile < my_tiles) set_src(s_tile);       \
      }                                               \
    }                                                 \
  } while (0)
#define QO_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
#define QO_WAIT_NEWEST_TILE()          \
  do {                                 \
    if (lw < 2) QO_VMCNT(13);          \
    else QO_VMCNT(12);                 \
  } while (0)
    set_src(0);
    QO_STAGE(0, NPLMAX);
    QO_ADVANCE();
    QO_STAGE(0, NPLMAX);
    QO_ADVANCE();
    if (total >= 2) QO_WAIT_NEWEST_TILE(); else QO_VMCNT(0);
    QO_BAR();  // B0
    constexpr int RPW = LBM / NL;  // 52 rows per DMA wave, one per lane
    float st_rstd = 0.f, st_shift = 0.f;
    auto tile_of = [&](int tile_i, int& img, int& head) {  // tile_i = dec_tile or dec_tile - 1 (set_src has run for both)
      img = tile_i == dec_tile ? dec_img : prev_img;
      head = tile_i == dec_tile ? dec_head : prev_head;
    };
    for (int g = 0; g < total; ++g) {
      if (d_kt == 0 && lane < RPW) {
        int img, head;
        tile_of(d_tile, img, head);
        const int m = row_of(img, lw * RPW + lane);
        const float4* pp = reinterpret_cast<const float4*>(p.rowpart + (size_t)m * kRowParts);
        float4 v[kRowParts / 2];
#pragma unroll
        for (int i = 0; i < kRowParts / 2; ++i) v[i] = 2 * i < p.nparts ? pp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < kRowParts / 2; ++i) {
          s1 += v[i].x;
          s2 += v[i].y;
          if (2 * i + 1 < p.nparts) {
            s1 += v[i].z;
            s2 += v[i].w;
          }
        }
        const float mean = s1 * p.inv_k;
        const float var = fmaxf(s2 * p.inv_k - mean * mean, 0.f);
        st_rstd = rsqrtf(var + 1e-5f);
        st_shift = -mean * st_rstd;
      }
      QO_STAGE(0, 4);
      QO_BAR();
      QO_STAGE(4, 7);
      QO_BAR();
      if (d_kt == 1 && lane < RPW) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) f32x2* lds_f2w_t;
        *(lds_f2w_t)(smem + kLRowstat + (lw * RPW + lane) * 8) = f32x2{st_rstd, st_shift};
      }
      if (!QUAD && d_kt == 1 && lw == 2) {
        int img, head;
        tile_of(d_tile, img, head);
        // the object token's key rules as one additive row: -60000 = not a key (the CLS row), -100 * mask on the
        // patch keys [REF oadp/oake/objects.py:206-213], 0 on its own key (row L)
        for (int key = lane; key < LBM; key += 64) {
          float b = -60000.f;  // not a key of the object token (the CLS row; rows past L are masked by position)
          if (key >= 1 && key < L) {
            const size_t mi = (size_t)img * (L - 1) + key - 1;
            b = -100.0f * (p.mask_f16 ? (float)reinterpret_cast<const f16_t*>(p.mask)[mi]
                                      : reinterpret_cast<const float*>(p.mask)[mi]);
          } else if (key >= L) {
            b = 0.f;
          }
          *reinterpret_cast<T*>(smem + kLMbias + (key & 15) * 32 + (key >> 4) * 2) = to16<T>(b);  // [key % 16][key / 16]
        }
      }
      QO_STAGE(7, 10);
      QO_BAR();
      QO_STAGE(10, NPLMAX);
      if (d_kt == 1) {
        int img, head;
        tile_of(d_tile, img, head);
        int n = head * LBN + 4 * lane;
        n = n + 4 <= (head + 1) * LBN ? n : (head + 1) * LBN - 4;
        if (lw == 0)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.bias + n), (lds_ptr_t)(smem + kLBias), 16, 0, 0);
        if (lw == 1)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.colsum + n), (lds_ptr_t)(smem + kLColsum), 16, 0, 0);
      }
      const bool tile_end = d_kt == nk - 1;
      const int win_buf = g % kLNStage;  // the slot of flat K-tile g
      if (++d_kt == nk) {
        d_kt = 0;
        ++d_tile;
      }
      const bool newer = g + 2 < total;
      QO_ADVANCE();
      if (newer) QO_WAIT_NEWEST_TILE(); else QO_VMCNT(0);
      QO_BAR();
      if (tile_end) {
        char* const qs = smem + win_buf * kLStage;
        char* const ks = qs + region;
        QO_BAR();  // X2
        const bool dstamp = lw == 0;  // DMA wave 0: measurement (role 2 of tools/qkv_attn_trace.py)
        const int dti = d_tile - 1;
        if (dstamp) QO_STAMP(2, dti, 0);
        ObjTask<T> task[1];
        QuadTask<T> qtask[2];
        if constexpr (QUAD) {
          quad_task_s<T, 2>(qtask, qs, ks, NW + lw, L, tid);  // (tasks 8 + lw and 12 + lw)
        } else {
          obj_task_s<T, 1>(task, qs, ks, smem + kLMbias, (NW + lw) * 16, L, tid);
        }
        if (dstamp) QO_STAMP(2, dti, 1);
        QO_BAR();  // X3
        if (dstamp) QO_STAMP(2, dti, 2);
        QO_BAR();  // X4
        if (dstamp) QO_STAMP(2, dti, 3);
        int img, head;
        tile_of(d_tile - 1, img, head);
        if constexpr (QUAD) {
          const int row_first = img * 4 * L;
          char* ot = reinterpret_cast<char*>(out + (size_t)row_first * C + head * kHeadDim);
          quad_task_pv<T, 2>(qtask, qs, NW + lw, L, ot, p.T - row_first, C, tid);
        } else {
          char* orow = reinterpret_cast<char*>(out + (size_t)img * L * C + head * kHeadDim);
          char* oy = reinterpret_cast<char*>(out + (size_t)(p.T + img) * C + head * kHeadDim);
          obj_task_pv<T, 1>(task, qs, (NW + lw) * 16, L, orow, oy, C, tid);
        }
        if (dstamp) QO_STAMP(2, dti, 4);
        QO_BAR();  // X5
      }
    }
#undef QO_STAGE
#undef QO_ADVANCE
#undef QO_VMCNT
#undef QO_WAIT_NEWEST_TILE
    return;
  }
  // ================= compute waves =================
  if (wid < 4)
    obj_compute_wave<T, QUAD, 7, 0, false>(smem, tid, wid, p, out, C, nk, my_tiles, xb, xslot, per_xcd);
  else
    obj_compute_wave<T, QUAD, 6, 112, true>(smem, tid, wid, p, out, C, nk, my_tiles, xb, xslot, per_xcd);
}

// rows of the folded in-projection in the kernel's column order: out row h * 192 + 48 wn + 16 m + j  <-  in row
// m * C + 64 h + 16 wn + j   (m = q, k, v; wn = column wave)
__device__ __forceinline__ int obj_src_row(int ro, int C) {
  const int h = ro / 192, rem = ro - h * 192, wn = rem / 48, r2 = rem - wn * 48, m = r2 >> 4, j = r2 & 15;
  return m * C + h * kHeadDim + wn * 16 + j;
}
__global__ void permute_qkv_obj_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int C, int chunks_per_row) {
  const long total = (long)3 * C * chunks_per_row;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ro = (int)(i / chunks_per_row), ch = (int)(i - (long)ro * chunks_per_row);
    out[i] = in[(long)obj_src_row(ro, C) * chunks_per_row + ch];
  }
}
__global__ void permute_qkv_obj_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * C) out[i] = in[obj_src_row(i, C)];
}

}  // namespace

bool qkv_attn_obj_supported(int L, int heads, int width, int n_img) {
  // (13 key tiles; the q and k regions of L + 1 rows share ONE ring slot: 2 (L + 1) 128 <= 51 200, i.e. L <= 199)
  return L + 1 > 16 * (kLNKT - 1) && L + 1 <= 16 * kLNKT && 2 * (L + 1) * kRowBytes <= kLStage && heads >= 1 && width == heads * kHeadDim && width % BK == 0 &&
         width / BK >= 3 && n_img >= 1;
}

bool qkv_attn_quad_supported(int L, int heads, int width, int n_img) {
  // (four images per tile: 4 L rows of q and of k in ONE ring slot, <= 64 keys per image)
  return L >= 1 && 2 * 4 * L * kRowBytes <= kLStage && heads >= 1 && width == heads * kHeadDim && width % BK == 0 && width / BK >= 3 &&
         n_img >= 1;
}

hipError_t launch_permute_qkv_obj(const void* w, const float* bias, const float* colsum, void* wp, float* biasp,
                                  float* colsump, int width, hipStream_t s) {
  const int chunks = width * 2 / 16;
  hipLaunchKernelGGL(permute_qkv_obj_rows_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const uint4*>(w),
                     reinterpret_cast<uint4*>(wp), width, chunks);
  const int blocks = (3 * width + 255) / 256;
  hipLaunchKernelGGL(permute_qkv_obj_vec_kernel, dim3(blocks), dim3(256), 0, s, bias, biasp, width);
  hipLaunchKernelGGL(permute_qkv_obj_vec_kernel, dim3(blocks), dim3(256), 0, s, colsum, colsump, width);
  return hipGetLastError();
}

template <typename T, bool QUAD>
static hipError_t qkv_attn_obj_launch_t(const void* x, const void* wp, const float* biasp, const float* colsump,
                                        const float* rowpart, int nparts, const void* mask, int mask_dtype, void* out,
                                        int n_img, int L, int heads, const LaunchOpts* opts, hipStream_t s,
                                        unsigned long long* trace) {
  static DynLdsAttr attr;
  auto kern = qkv_attn_obj_kernel<T, QUAD>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), kLLdsBytes); e != hipSuccess) return e;
  int num_cu = 0;
  if (hipError_t e = device_cu_count(&num_cu); e != hipSuccess) return e;
  if (opts && opts->cu_count > 0 && opts->cu_count < num_cu) num_cu = opts->cu_count;
  QkvAttnObjParams p{};
  p.bias = biasp; p.colsum = colsump; p.rowpart = reinterpret_cast<const float2*>(rowpart); p.nparts = nparts;
  const int C = heads * kHeadDim;
  p.inv_k = 1.0f / (float)C;
  p.n_img = n_img; p.L = L; p.H = heads; p.T = n_img * L;
  p.mask = mask; p.mask_f16 = mask_dtype == DT_F16 ? 1 : 0;
  // head blocks of qkv_walk heads (LaunchOpts: 0 = group-major), group blocks of 32 / qkv_walk groups
  const int hq = opts ? opts->qkv_walk : kQkvWalkDefault;
  p.walk_hq = hq > 0 && heads % hq == 0 ? hq : 0;
  p.walk_gq = p.walk_hq > 0 ? (32 / p.walk_hq > 0 ? 32 / p.walk_hq : 1) : 0;
  p.trace = trace;
  const int ntiles = (QUAD ? (n_img + 3) / 4 : n_img) * heads;
  int grid = (num_cu / 8) * 8;
  if (grid < 8) grid = 8;
  const int need = ((ntiles + 7) / 8) * 8;
  if (grid > need) grid = need;
  OAKE_LAUNCH(kern, dim3(grid), dim3(768), kLLdsBytes, s, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(wp),
              reinterpret_cast<T*>(out), C, p);
  return hipGetLastError();
}

hipError_t launch_qkv_attn_obj(int dtype16, const void* x, const void* wp, const float* biasp, const float* colsump,
                               const float* rowpart, int nparts, const void* mask, int mask_dtype, void* out, int n_img,
                               int L, int heads, const LaunchOpts* opts, hipStream_t s, unsigned long long* trace) {
  if (!qkv_attn_obj_supported(L, heads, heads * kHeadDim, n_img) || nparts < 1 || !mask ||
      (mask_dtype != DT_F16 && mask_dtype != DT_F32))
    return hipErrorInvalidValue;
  if (dtype16 == DT_BF16)
    return qkv_attn_obj_launch_t<bf16_t, false>(x, wp, biasp, colsump, rowpart, nparts, mask, mask_dtype, out, n_img, L, heads, opts, s, trace);
  return qkv_attn_obj_launch_t<f16_t, false>(x, wp, biasp, colsump, rowpart, nparts, mask, mask_dtype, out, n_img, L, heads, opts, s, trace);
}

hipError_t launch_qkv_attn_quad(int dtype16, const void* x, const void* wp, const float* biasp, const float* colsump,
                                const float* rowpart, int nparts, void* out, int n_img, int L, int heads,
                                const LaunchOpts* opts, hipStream_t s, unsigned long long* trace) {
  if (!qkv_attn_quad_supported(L, heads, heads * kHeadDim, n_img) || nparts < 1) return hipErrorInvalidValue;
  if (dtype16 == DT_BF16)
    return qkv_attn_obj_launch_t<bf16_t, true>(x, wp, biasp, colsump, rowpart, nparts, nullptr, DT_F32, out, n_img, L, heads, opts, s, trace);
  return qkv_attn_obj_launch_t<f16_t, true>(x, wp, biasp, colsump, rowpart, nparts, nullptr, DT_F32, out, n_img, L, heads, opts, s, trace);
}

}  // namespace oake
