// qkv_attn.hip — LayerNorm-folded QKV in-projection + multi-head self-attention of SHORT sequences (L <= 53:
// encode_image at 224^2 / patch 32 and blocks mode, L = 50) as ONE persistent kernel for gfx950.
//
//   reference ops  ln_1 -> attn.in_proj (q | k | v) -> softmax(q k^T / 8) v      [REF oadp/oake/globals.py:57,
//                  oadp/oake/blocks.py:129: clip's ResidualAttentionBlock.attention; SURVEY.md §8 A15c-e]
//
// What it replaces: gemm_pp_kernel<EPI_T16_BIAS_LN> (writes qkv [T, 3C], 59 MB per layer at batch 256) followed by
// attention_pair_kernel (reads it back, writes att [T, C]): 118 MB of HBM-side round trip and one launch per layer.
// Here the qkv values never leave the CU.
//
//   * A tile is (image group, head): BM = 160 residual rows = 3 images x L tokens (150 of 160 rows at L = 50; the
//     rows behind them belong to the next group and are computed for nothing: 6.7 % padding MFMAs), BN = 192
//     columns = this head's q | k | v (the folded weight rows are permuted head-major at load time:
//     launch_permute_qkv).  256 images x 12 heads = 86 x 12 = 1032 tiles, persistent blocks, XCD-contiguous order
//     with the 12 heads of a group running on one XCD (the A rows come from that XCD's L2).
//   * K loop: the production GEMM's schedule (gemm.hip, gemm_pp_kernel, two long phases per K-tile): 8 compute
//     waves as two row groups (wave tile 80 x 48 = 5 x 3 MFMA tiles) one phase apart + 4 LDS-DMA waves staging
//     K-tile g + 2 into a 3-slot ring, LayerNorm statistics summed from the producer's row slices by the DMA waves.
//   * Tile end: both groups write their accumulators — affine'd (rstd * acc + (-mean rstd) * colsum + bias'), rounded
//     to 16 bits, exactly the values the unfused path stores — into LDS as three per-image [V | K | Q] regions of
//     L x 128 B rows (16-byte chunks XOR-swizzled by (row >> 1) & 7 as everywhere): images 0 and 1 in the ring slot
//     the tile's LAST K-tile has just left (its next user, K-tile 2 of the next tile, is not staged before the
//     attention is done: the DMA waves take the same three barriers), image 2 in the 25 KB behind the ring.  Then
//     six waves run attention_pair_kernel's body — (image, query half): S^T = K Q^T, softmax over the keys in
//     registers, O^T = V^T P^T through the transpose read — and store O (150 x 64 x 16 bit per tile) as full lines.
//   * Nothing of a tile is pending across the K loop, so the accumulators are dead during the attention phase
//     (the kernel stays inside the 168-register budget of three waves per SIMD).
//
// Barrier schedule (workgroup barriers: all 12 waves take every one).  Per tile of nk K-tiles:
//     B1(0) B2(0) B1(1) B2(1) ... B1(nk-1) B2(nk-1) X2 X3
//   group 0: loads(kt) | B1 | MFMA(kt) | B2 ...            after B2(nk-1): window write | X2 | attention task | X3
//   group 1: B1(0); loads(kt) | B2 | MFMA(kt) | B1(kt+1)    after MFMA(nk-1): window write | X2 | attention task | X3
//   DMA    : stage 1st half of K-tile g+2 | B1 | 2nd half, wait for K-tile g+1 | B2    after the tile's last: X2 | attention task | X3
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

constexpr int BK = 64;
constexpr int kRowBytes = BK * 2;  // 128
constexpr int QBM = 160, QBN = 192;
constexpr int kStage = (QBM + QBN) * kRowBytes;  // 45 056
constexpr int kNStage = 3;
constexpr int kEpi = kNStage * kStage;           // 135 168: bias[192] | colsum[192] | rowstat[160] (1 KiB slots)
constexpr int kEpiBias = kEpi, kEpiColsum = kEpi + 1024, kEpiRowstat = kEpi + 2048;
constexpr int kSpare = kEpi + 2048 + QBM * 8;    // 138 496: the third image's [V | K | Q] regions
constexpr int kLdsBytes = 160 * 1024;
constexpr int kRowParts = 16;                    // float2 slots per row of rowpart (gemm.hip)
constexpr int kMaxL = 53;                        // 3 L <= 160 rows

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct QkvAttnParams {
  const float* bias;      // [H * 192] folded bias, head-major (q | k | v per head)
  const float* colsum;    // [H * 192]
  const float2* rowpart;  // [M, 16] (sum x, sum x^2) slices of the residual rows
  int nparts;
  float inv_k;
  int n_img, L, H, groups, ipt;  // images, tokens per image, heads, image groups (tiles per head), images per group
  unsigned long long* trace;     // measurement (tools/qkv_attn_trace.py): [block < 64][role 3][tile < 6][8] cycle stamps, or nullptr
};

// Attention of 16 queries of one (image, head): attention_pair_kernel's body on one query tile, Q / K / V from the LDS
// window (regions of L x 128-byte rows, chunk-swizzled).  S^T = K Q^T, softmax over the keys in registers, O^T = V^T P^T
// through the transpose read; O goes back through the wave's own 16 Q rows (nobody else reads them as queries; as padded
// KEYS of another wave they are masked) and leaves as full lines.
template <typename T>
__device__ __forceinline__ void attn_task16(char* vs, int region, int q0, int L, T* obase, int ldo, bool store_ok, int tid_) {
  typedef typename T16<T>::vec8 vec8;
  char* const ks = vs + region;
  char* const qs = ks + region;
  int atid = tid_;
  asm volatile("" : "+v"(atid));  // (lane coordinates re-derived here: not kept in VGPRs across the K loop)
  const int fr = atid & 15, g = (atid & 63) >> 4;
  const int sw = (fr >> 1) & 7;
  vec8 qf[2], kf[4][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
    qf[kk] = *reinterpret_cast<const vec8*>(qs + (q0 + fr) * kRowBytes + (((kk * 4 + g) ^ sw) << 4));
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      kf[kt][kk] = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * kRowBytes + (((kk * 4 + g) ^ sw) << 4));
  f32x4 sacc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) sacc[kt] = T16<T>::mfma(kf[kt][kk], qf[kk], sacc[kt]);
  // the V fragments are requested before the softmax runs: their LDS latency hides under it
  vec8 vf[2][4];
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2) {
    const int row0 = 32 * ks2 + 4 * g + (fr >> 2);  // and row0 + 16: same swizzle
    const int vsw = (row0 >> 1) & 7;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int c4 = (fr & 3) * 4;
      const char* p0 = vs + row0 * kRowBytes + (((dt * 2 + (c4 >> 3)) ^ vsw) << 4) + (c4 & 4) * 2;
      typedef s16x4 __attribute__((address_space(3))) * lds4_t;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 16 * kRowBytes));
      s16x8 both;
      both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
      both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
      vf[ks2][dt] = __builtin_bit_cast(vec8, both);
    }
  }
  constexpr float kLog2e = 1.4426950408889634f;
  float mx = -1e30f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if ((kt + 1) * 16 > L) {  // (wave-uniform) a key tile with padded keys: at L = 50 only the last one
#pragma unroll
      for (int i = 0; i < 4; ++i) sacc[kt][i] = kt * 16 + 4 * g + i < L ? sacc[kt][i] : -1e30f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, sacc[kt][i]);
  }
  mx = rows16_max(mx);
  const float nb = -mx * kLog2e;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][i], kLog2e, nb));
      sacc[kt][i] = e;
      sum += e;
    }
  const float inv = 1.0f / rows16_sum(sum);
  vec8 pf[2];
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2) {
    vec8 p8;
#pragma unroll
    for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[2 * ks2 + (j >> 2)][j & 3]);
    pf[ks2] = p8;
  }
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = T16<T>::mfma(vf[ks2][dt], pf[ks2], oacc[dt]);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const f32x4 o = oacc[dt];
    if (q0 + fr < L)
      *reinterpret_cast<uint2*>(qs + (q0 + fr) * kRowBytes + (((dt * 2 + (g >> 1)) ^ sw) << 4) + (g & 1) * 8) =
          pack4<T>(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int ln = atid & 63;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = q0 + (ln >> 3) + 8 * i;
    const uint4 v = *reinterpret_cast<const uint4*>(qs + row * kRowBytes + (((ln & 7) ^ ((row >> 1) & 7)) << 4));
    if (store_ok && row < L) aux_store16(obase + (size_t)row * ldo + (ln & 7) * 8, v);
  }
}

template <typename T>
__global__ __launch_bounds__(768) void qkv_attn_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                       T* __restrict__ out, int M, int K, QkvAttnParams p) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = 8, NL = 4;
  constexpr int MI = 5, NI = 3, TM = 80, TN = 48;
  constexpr int NINST = (QBM + QBN) / 8;  // 44 pieces of 1 KiB per K-tile
  constexpr int NPL = NINST / NL;         // 11 per DMA wave: 5 of A rows, 6 of W rows
  constexpr int kAPieces = QBM / 8 / NL;  // 5
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.L, H = p.H;
  const int region = L * kRowBytes;  // one matrix of one image

  // this block's tiles: XCD x owns logical tiles [xb, xb + xc), block b / 8 of it takes xb + b / 8 + i * (blocks per XCD)
  const int nx = 8;
  const int ntiles = p.groups * H;
  const int xcd = blockIdx.x % nx, xslot = blockIdx.x / nx;
  const int per_xcd = gridDim.x / nx;
  const int q_ = ntiles / nx, r_ = ntiles % nx;
  const int xb = xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
  const int xc = xcd < r_ ? q_ + 1 : q_;
  const int my_tiles = xslot < xc ? (xc - xslot + per_xcd - 1) / per_xcd : 0;
  if (my_tiles == 0) return;
  const int nk = K / BK;
  const int total = my_tiles * nk;
#define QA_PIN() __builtin_amdgcn_sched_barrier(0)
#define QA_BAR()                  \
  do {                            \
    QA_PIN();                     \
    __builtin_amdgcn_s_barrier(); \
    QA_PIN();                     \
  } while (0)
#define QA_TILE(i_, m0_, n0_, grp_, head_)                  \
  const int _t##m0_ = xb + xslot + (i_) * per_xcd;          \
  const int grp_ = _t##m0_ / H, head_ = _t##m0_ - grp_ * H; \
  const int m0_ = grp_ * p.ipt * L, n0_ = head_ * QBN;      \
  (void)m0_;                                                \
  (void)n0_

#define QA_STAMP(role_, tile_, k_)                                                                         \
  do {                                                                                                     \
    if (p.trace != nullptr && lane == 0 && blockIdx.x < 64 && (tile_) < 6)                                 \
      p.trace[(((size_t)blockIdx.x * 3 + (role_)) * 6 + (tile_)) * 8 + (k_)] = __builtin_readcyclecounter(); \
  } while (0)

  if (wid >= NW) {
    // ================= DMA wave =================
    const int lw = wid - NW;
    const char* src[NPL];
    auto set_src = [&](int tile_i) {
      QA_TILE(tile_i, m0, n0, grp, head);
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int rr = 8 * (lw + NL * j) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rr >> 1) & 7);
        if (j < kAPieces) {
          int gr = m0 + rr;
          gr = gr < M ? gr : M - 1;
          src[j] = reinterpret_cast<const char*>(A + (size_t)gr * K) + chunk * 16;
        } else {
          src[j] = reinterpret_cast<const char*>(W + (size_t)(n0 + rr - QBM) * K) + chunk * 16;
        }
      }
    };
    int s_g = 0, s_kt = 0, s_tile = 0, s_buf = 0;  // producer cursor
    int d_kt = 0, d_tile = 0;                      // consumer position (group 0)
#define QA_STAGE(j0_, j1_)                                                                              \
  do {                                                                                                  \
    if (s_g < total) {                                                                                  \
      char* _base = smem + s_buf * kStage;                                                              \
      const size_t _koff = (size_t)s_kt * (BK * 2);                                                     \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j)                                          \
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[_j] + _koff),                                \
                                           (lds_ptr_t)(_base + (lw + NL * _j) * 1024), 16, 0, 0);       \
    }                                                                                                   \
  } while (0)
#define QA_ADVANCE()                               \
  do {                                             \
    if (s_g < total) {                             \
      ++s_g;                                       \
      s_buf = s_buf == kNStage - 1 ? 0 : s_buf + 1; \
      if (++s_kt == nk) {                          \
        s_kt = 0;                                  \
        ++s_tile;                                  \
        if (s_tile < my_tiles) set_src(s_tile);    \
      }                                            \
    }                                              \
  } while (0)
#define QA_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
    constexpr int Q2 = (2 * NPL + 3) / 4;
    set_src(0);
    QA_STAGE(0, NPL);
    QA_ADVANCE();
    QA_STAGE(0, NPL);
    QA_ADVANCE();
    if (total >= 2) QA_VMCNT(NPL); else QA_VMCNT(0);
    QA_BAR();  // B0: flat K-tile 0 published
    constexpr int RPW = QBM / NL;  // 40 rows per DMA wave, one per lane
    float st_rstd = 0.f, st_shift = 0.f;
    for (int g = 0; g < total; ++g) {
      if (d_kt == 0 && lw == 0) QA_STAMP(2, d_tile, 0);
      if (d_kt == 0 && lane < RPW) {
        QA_TILE(d_tile, m0, n0, grp, head);
        int m = m0 + lw * RPW + lane;
        m = m < M ? m : M - 1;
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
      QA_STAGE(0, Q2);
      QA_BAR();  // B1(g)
      if (d_kt == 1 && lane < RPW) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) f32x2* lds_f2w_t;
        *(lds_f2w_t)(smem + kEpiRowstat + (lw * RPW + lane) * 8) = f32x2{st_rstd, st_shift};
      }
      QA_STAGE(Q2, NPL);
      if (d_kt == 1) {  // the tile's bias / colsum block (the previous tile's epilogue ended at its X3)
        QA_TILE(d_tile, m0, n0, grp, head);
        int n = n0 + 4 * lane;
        n = n + 4 <= n0 + QBN ? n : n0 + QBN - 4;
        if (lw == 0)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.bias + n), (lds_ptr_t)(smem + kEpiBias), 16, 0, 0);
        if (lw == 1)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(p.colsum + n), (lds_ptr_t)(smem + kEpiColsum), 16, 0, 0);
      }
      const bool tile_end = d_kt == nk - 1;
      if (++d_kt == nk) {
        d_kt = 0;
        ++d_tile;
      }
      const bool newer = g + 2 < total;
      QA_ADVANCE();
      if (newer) QA_VMCNT(NPL); else QA_VMCNT(0);  // flat K-tile g + 1 landed
      QA_BAR();  // B2(g)
      if (tile_end) {  // window write | attention | done: the slot of K-tile g is the window until X3
        if (lw == 0) QA_STAMP(2, d_tile - 1, 1);
        QA_BAR();  // X2: the window is complete; twelve waves = 3 images x 4 query tiles of 16
        {
          QA_TILE(d_tile - 1, m0, n0, grp, head);
          const int img = 2, q0 = lw * 16;  // (waves 0-7: images 0 and 1; the DMA waves: image 2)
          const int img_g = grp * p.ipt + img;
          if (q0 < L)
            attn_task16<T>(smem + kSpare, region, q0, L, out + (size_t)img_g * L * (H * kHeadDim) + head * kHeadDim,
                           H * kHeadDim, img < p.ipt && img_g < p.n_img, tid);
        }
        QA_BAR();  // X3
        if (lw == 0) QA_STAMP(2, d_tile - 1, 5);
      }
    }
#undef QA_STAGE
#undef QA_ADVANCE
#undef QA_VMCNT
    return;
  }

  // ================= compute wave =================
  const int wm = wid >> 2, wn = wid & 3;
  const bool late = wid >= 4;
  const int frow = lane & 15, fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int a_base = (wm * TM + frow) * kRowBytes;
  const int b_base = QBM * kRowBytes + (wn * TN + frow) * kRowBytes;
  const int koff0 = ((0 * 4 + fg) ^ fsw) << 4;
  const int koff1 = ((1 * 4 + fg) ^ fsw) << 4;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  QA_BAR();  // B0
  int c_buf = 0;
  const int role = wid == 0 ? 0 : (wid == 4 ? 1 : 3);
  for (int ti = 0; ti < my_tiles; ++ti) {
    if (role < 3) QA_STAMP(role, ti, 0);
    if (late) QA_BAR();  // B1(0): group 1 runs one phase behind group 0
    for (int kt = 0; kt < nk; ++kt) {
      vec8 af[MI], bf[NI], af1[MI], bf1[NI];
      {
        const char* st = smem + c_buf * kStage;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff0);
#pragma unroll
        for (int i = 0; i < NI; ++i) bf[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff0);
#pragma unroll
        for (int i = 0; i < MI; ++i) af1[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff1);
#pragma unroll
        for (int i = 0; i < NI; ++i) bf1[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff1);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      QA_BAR();  // group 0: B1(kt); group 1: B2(kt)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T16<T>::mfma(bf[ni], af[mi], acc[mi][ni]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T16<T>::mfma(bf1[ni], af1[mi], acc[mi][ni]);
      __builtin_amdgcn_s_setprio(0);
      const bool last = kt == nk - 1;
      if (!last) c_buf = c_buf == kNStage - 1 ? 0 : c_buf + 1;
      if (!late || !last) QA_BAR();  // group 0: B2(kt); group 1: B1(kt + 1)
    }
    // ---------------- tile end ----------------
    // c_buf = the slot of the tile's last K-tile: every wave has read its fragments of it (group 0 before B1, group 1
    // before B2 of that K-tile) -> it holds images 0 and 1 of the [V | K | Q] window, the spare region image 2
    char* const win01 = smem + c_buf * kStage;
    char* const win2 = smem + kSpare;
    if (role < 3) QA_STAMP(role, ti, 1);
    // (no barrier here: group 0 writes its rows under group 1's last MFMA phase — every wave has read its fragments of
    // the slot by B2 of that K-tile —, group 1 right after its own MFMAs)
    {
      QA_PIN();
      int etid = tid;  // (re-derived behind an opaque asm: hipcc would otherwise keep every epilogue address in a VGPR
      asm volatile("" : "+v"(etid));  // across the K loop — gemm.hip)
      const int er = etid & 15, eg = (etid & 63) >> 4;
      // every epilogue constant first (one LDS round trip for all of them), then arithmetic and stores back to back
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef const __attribute__((address_space(3))) f32x2* lds_f2_t;
      float4 b4[NI], c4[NI];
      f32x2 rs[MI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        b4[ni] = *reinterpret_cast<const float4*>(smem + kEpiBias + (wn * TN + ni * 16 + 4 * eg) * 4);
        c4[ni] = *reinterpret_cast<const float4*>(smem + kEpiColsum + (wn * TN + ni * 16 + 4 * eg) * 4);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) rs[mi] = *(lds_f2_t)(smem + kEpiRowstat + (wm * TM + mi * 16 + er) * 8);
      // column part of the address: matrix region (V | K | Q) + byte inside the 16-byte chunk; chunk index for the swizzle
      int coff[NI], cch[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int cn = wn * TN + ni * 16;
        const int mat = cn >> 6, d0 = (cn & 63) + 4 * eg;
        coff[ni] = (2 - mat) * region + ((d0 & 7) << 1);
        cch[ni] = d0 >> 3;
      }
      const int img_stride = 3 * region;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int R = wm * TM + mi * 16 + er;
        // image of the row and row inside it, without integer multiplies (v_mul_lo_u32 is a quarter-rate instruction)
        const bool ge1 = R >= L, ge2 = R >= 2 * L;
        const int ri = R - (ge1 ? L : 0) - (ge2 ? L : 0);
        char* rowp = (ge2 ? win2 : (ge1 ? win01 + img_stride : win01)) + ri * kRowBytes;
        const int sw = (ri >> 1) & 7;
        const f32x2 r = rs[mi];
        if (R < 3 * L) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const f32x4 a = acc[mi][ni];
            const uint2 pk = pack4<T>(fmaf(a[0], r[0], fmaf(r[1], c4[ni].x, b4[ni].x)), fmaf(a[1], r[0], fmaf(r[1], c4[ni].y, b4[ni].y)),
                                      fmaf(a[2], r[0], fmaf(r[1], c4[ni].z, b4[ni].z)), fmaf(a[3], r[0], fmaf(r[1], c4[ni].w, b4[ni].w)));
            *reinterpret_cast<uint2*>(rowp + coff[ni] + ((cch[ni] ^ sw) << 4)) = pk;
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (role < 3) QA_STAMP(role, ti, 2);
    QA_BAR();  // X2: the window is complete; twelve waves = 3 images x 4 query tiles of 16
    if (role < 3) QA_STAMP(role, ti, 3);
    {
      QA_TILE(ti, m0, n0, grp, head);
      const int img = wid >> 2, q0 = (wid & 3) * 16;  // (the DMA waves take image 2)
      const int img_g = grp * p.ipt + img;
      if (q0 < L)
        attn_task16<T>(win01 + img * 3 * region, region, q0, L, out + (size_t)img_g * L * (H * kHeadDim) + head * kHeadDim,
                       H * kHeadDim, img < p.ipt && img_g < p.n_img, tid);
    }
    if (role < 3) QA_STAMP(role, ti, 4);
    QA_BAR();  // X3: the window slot goes back to the ring
    if (role < 3) QA_STAMP(role, ti, 5);
    c_buf = c_buf == kNStage - 1 ? 0 : c_buf + 1;
  }
#undef QA_TILE
#undef QA_STAMP
#undef QA_BAR
#undef QA_PIN
}

// rows of the folded in-projection, head-major: out row h * 192 + 64 m + j  <-  in row m * C + 64 h + j   (m = q, k, v)
__global__ void permute_qkv_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int C, int chunks_per_row) {
  const long total = (long)3 * C * chunks_per_row;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ro = (int)(i / chunks_per_row), ch = (int)(i - (long)ro * chunks_per_row);
    const int h = ro / 192, rem = ro - h * 192, m = rem >> 6, j = rem & 63;
    const int ri = m * C + h * kHeadDim + j;
    out[i] = in[(long)ri * chunks_per_row + ch];
  }
}
__global__ void permute_qkv_vec_kernel(const float* __restrict__ in, float* __restrict__ out, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * C) {
    const int h = i / 192, rem = i - h * 192, m = rem >> 6, j = rem & 63;
    out[i] = in[m * C + h * kHeadDim + j];
  }
}

}  // namespace

bool qkv_attn_supported(int L, int heads, int width, int n_img) {
  return L >= 1 && L <= kMaxL && heads >= 1 && width == heads * kHeadDim && width % BK == 0 && width / BK >= 3 &&
         n_img >= 1;
}

hipError_t launch_permute_qkv(int dtype16, const void* w, const float* bias, const float* colsum, void* wp, float* biasp,
                              float* colsump, int width, hipStream_t s) {
  (void)dtype16;
  const int chunks = width * 2 / 16;
  hipLaunchKernelGGL(permute_qkv_rows_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const uint4*>(w),
                     reinterpret_cast<uint4*>(wp), width, chunks);
  const int blocks = (3 * width + 255) / 256;
  hipLaunchKernelGGL(permute_qkv_vec_kernel, dim3(blocks), dim3(256), 0, s, bias, biasp, width);
  hipLaunchKernelGGL(permute_qkv_vec_kernel, dim3(blocks), dim3(256), 0, s, colsum, colsump, width);
  return hipGetLastError();
}

template <typename T>
static hipError_t qkv_attn_launch_t(const void* x, const void* wp, const float* biasp, const float* colsump,
                                    const float* rowpart, int nparts, void* out, int n_img, int L, int heads,
                                    const LaunchOpts* opts, hipStream_t s, unsigned long long* trace) {
  static DynLdsAttr attr;
  auto kern = qkv_attn_kernel<T>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), kLdsBytes); e != hipSuccess) return e;
  int num_cu = 0;
  if (hipError_t e = device_cu_count(&num_cu); e != hipSuccess) return e;
  if (opts && opts->cu_count > 0 && opts->cu_count < num_cu) num_cu = opts->cu_count;
  QkvAttnParams p{};
  p.bias = biasp; p.colsum = colsump; p.rowpart = reinterpret_cast<const float2*>(rowpart); p.nparts = nparts;
  const int C = heads * kHeadDim;
  p.inv_k = 1.0f / (float)C;
  p.n_img = n_img; p.L = L; p.H = heads;
  p.ipt = 3;
  p.trace = trace;
  p.groups = (n_img + p.ipt - 1) / p.ipt;
  const int ntiles = p.groups * heads;
  int grid = (num_cu / 8) * 8;
  if (grid < 8) grid = 8;
  const int need = ((ntiles + 7) / 8) * 8;
  if (grid > need) grid = need;
  OAKE_LAUNCH(kern, dim3(grid), dim3(768), kLdsBytes, s, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(wp),
              reinterpret_cast<T*>(out), n_img * L, C, p);
  return hipGetLastError();
}

hipError_t launch_qkv_attn(int dtype16, const void* x, const void* wp, const float* biasp, const float* colsump,
                           const float* rowpart, int nparts, void* out, int n_img, int L, int heads,
                           const LaunchOpts* opts, hipStream_t s, unsigned long long* trace) {
  if (!qkv_attn_supported(L, heads, heads * kHeadDim, n_img) || nparts < 1) return hipErrorInvalidValue;
  if (dtype16 == DT_BF16)
    return qkv_attn_launch_t<bf16_t>(x, wp, biasp, colsump, rowpart, nparts, out, n_img, L, heads, opts, s, trace);
  return qkv_attn_launch_t<f16_t>(x, wp, biasp, colsump, rowpart, nparts, out, n_img, L, heads, opts, s, trace);
}

}  // namespace oake
