// attn_out.hip — multi-head self-attention + out_proj + residual of one transformer block in ONE kernel, for
// short sequences (L <= 64: encode_image at 224^2 / patch 32 and blocks mode, L = 50) on a 16-bit residual stream.
//
// Replaces two launches of the layer — `attention` (a pure memory round trip at L = 50: 59 MB of q/k/v in, 19.7 MB
// of attention output out) and `gemm_out_proj` (N = 768: 240 tiles for 256 CUs, one tile per CU, residual epilogue
// and end-of-kernel drain fully exposed) — and the `att` buffer between them:
//     x[n] += softmax(q k^T) v  W_out^T + b_out          [REF clip model.py ResidualAttentionBlock.attention, as called
//                                                         by encode_image: oadp/oake/globals.py:57, blocks.py:129]
// One workgroup per image (batch 256 = 256 CUs), 8 waves (two per SIMD, 256 registers each):
//   * out_proj's K dimension is (head, d): its sum splits by head, so head h's contribution
//     O_h [L x 64] . W_out[:, 64 h .. 64 h + 63]^T is accumulated as soon as O_h exists.  A step = 2 heads: for the
//     attention part the 8 waves are 2 heads x 4 query tiles of 16 rows (S^T = K Q^T as in attention_pair_kernel:
//     softmax in registers, P as the B operand of the PV product through a permuted key enumeration, V by
//     transposing LDS reads); for out_proj each wave owns 96 output columns — one full 128-byte line of every row
//     (columns 64 w ..) and half of one of the last four lines (columns 512 + 32 w ..) — x all 64 (padded) rows:
//     4 x 6 accumulator tiles = 96 registers for the whole image.
//   * software pipeline: while a wave issues the MFMAs of step s (4 groups of 24, 32 k-columns each) it runs the
//     attention of step s + 1 between the groups — the softmax's VALU chain fills the issue slots the matrix pipe
//     leaves — so there is ONE workgroup barrier per step (it publishes O of step s + 1 and frees the K / V stage).
//   * the attention output never leaves the chip and is never transposed: a lane's PV accumulators are, as they
//     stand, two B-operand fragments of the out_proj MFMA for a permuted k enumeration (k = 16 (i >> 2) + 4 g +
//     (i & 3) within a 32-column block), written to LDS as 1-KiB fragments (ds_write_b128, linear) and read back by
//     all 8 waves (ds_read_b128, linear: no swizzle, no conflicts).  W_out is stored ONCE at load time in exactly
//     that fragment order (permute_out_w_kernel): every A-operand fetch is one fully coalesced 1-KiB load per wave
//     straight into registers — no LDS, no address arithmetic beyond an immediate offset, no lane swaps — issued two
//     groups ahead of its use (sched_barrier keeps hipcc from sinking the loads back to their use).
//   * K / V of a step's two heads arrive by LDS-DMA into a two-stage ring (source-side XOR swizzle as in gemm.hip /
//     attention_pair_kernel), two steps ahead of the attention that reads them.
//   * epilogue = EPI_RESID16 of gemm.hip: + bias + residual row (16-bit, in place; lane-swapped full 128-byte lines,
//     written through, for the wave's own line) and the (sum x, sum x^2) of every 64-column slice into rowpart — the
//     LayerNorm statistics the LN-folded c_fc GEMM that follows consumes; the four slices whose
//     halves belong to two waves are added up through LDS in a fixed order.
// Arithmetic per image: out_proj 4 x 6 x 24 = 576 MFMAs per wave (rows padded 50 -> 64) + 16 per wave and step for
// the attention: 672 x 8 waves x 16 cycles / 4 SIMDs = 21.5 k cycles; W_out streams from the XCD's L2 at 1.18 MB
// per image = 18.4 k cycles of the CU's 64 B/clk vector-memory path; q/k/v 230 KB per image from the fabric.
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

constexpr int kAoWaves = 8;
constexpr int kAoHeads = 12;
constexpr int kAoC = kAoHeads * 64;                       // 768
constexpr int kAoStepHeads = 2;                           // 8 waves = 2 heads x 4 query tiles
constexpr int kAoSteps = kAoHeads / kAoStepHeads;         // 6
constexpr int kAoStepGroups = 2 * kAoStepHeads;           // 32-column k groups of out_proj per step: 4
constexpr int kAoGroups = 2 * kAoHeads;                   // ... per image: 24
constexpr int kAoNT = 6;                                  // 16-column tiles per wave: 4 (its line) + 2 (its half line)
constexpr int kAoRegion = 64 * 128;                       // K or V of one head: 64 rows x 128 B (rows >= L zero)
constexpr int kAoStage = kAoStepHeads * 2 * kAoRegion;    // 32 KB
constexpr int kAoObuf = kAoStepHeads * 4 * 2 * 1024;      // a step's O fragments: [head][query tile][kk] x 1 KiB
constexpr int kAoStages = 3;                              // K / V ring: a unit's rows arrive two steps before they are read
constexpr float kLog2eAo = 1.4426950408889634f;

// first output column of tile nt of wave w: its own line (tiles 0..3) or its half of line 8 + (w >> 1) (tiles 4, 5)
__host__ __device__ constexpr int ao_col0(int w, int nt) { return nt < 4 ? 64 * w + 32 * (nt >> 1) : 512 + 32 * w; }

// W_out [C, C] (row n = output feature, K contiguous) -> the A-operand fragments of attn_out_kernel, in fetch order:
// fragment (wave w, group G = 2 head + kk, column tile nt) is 64 lanes x 8 values = 1 KiB;
//   lane (r = l & 15, g = l >> 4), value i:  n = ao_col0(w, nt) + 8 (r >> 2) + 4 (nt & 1) + (r & 3)
//                                            k = 32 G + 16 (i >> 2) + 4 g + (i & 3)
// (n: the pair interleave of gemm.hip's 16-bit epilogues — a lane's accumulators of tiles 2t, 2t+1 are 8 consecutive
// columns; k: the enumeration in which a lane's PV accumulators are B fragments as they stand.)
template <typename T>
__global__ void permute_out_w_kernel(const T* __restrict__ w, T* __restrict__ wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one output value
  if (idx >= kAoC * kAoC) return;
  const int i = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
  const int nt = frag % kAoNT, G = (frag / kAoNT) % kAoGroups, wv = frag / (kAoNT * kAoGroups);
  const int r = lane & 15, g = lane >> 4;
  const int n = ao_col0(wv, nt) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
  const int k = 32 * G + 16 * (i >> 2) + 4 * g + (i & 3);
  wp[idx] = w[(size_t)n * kAoC + k];
}

// TRACE (measurement builds of the kernel only, oake_debug_attn_out_trace): s_memtime stamps of every wave of the
// first kAoTraceBlocks workgroups at the phase boundaries, trace[(block * 8 + wave) * 64 + point]
constexpr int kAoTraceBlocks = 4;
template <typename T, bool TRACE>
__global__ __launch_bounds__(kAoWaves * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_out_kernel(const T* __restrict__ qkv, const T* __restrict__ wperm, const float* __restrict__ bias,
                     T* __restrict__ x, float2* __restrict__ rowpart, int L, unsigned long long* __restrict__ trace) {
  typedef typename T16<T>::vec8 vec8;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  // Separate LDS objects, not one dynamic array: hipcc guards every DS read that MAY alias an LDS-DMA in flight with
  // a vmcnt wait for it, and it tells accesses apart by the LDS variable they belong to (the alias scopes the
  // module-LDS lowering attaches).  With one array every O-fragment read of a step waited ~3.7 k cycles for the K / V
  // rows requested at the step's start (tools/attn_out_trace.py); with the ring's stages as distinct variables (and
  // the step loop fully unrolled, so that each access names its stage statically) only real hazards wait.
  __shared__ __attribute__((aligned(16))) char kv0[kAoStage], kv1[kAoStage], kv2[kAoStage];  // 3 x 32 KB
  __shared__ __attribute__((aligned(16))) char ob0[kAoObuf], ob1[kAoObuf];                  // 2 x 16 KB
  __shared__ float2 stat[64 * kAoWaves];                                                     // 4 KB
  auto kvp = [&](int u) -> char* { return u % kAoStages == 0 ? kv0 : (u % kAoStages == 1 ? kv1 : kv2); };
  auto obp = [&](int u) -> char* { return (u & 1) ? ob1 : ob0; };
  constexpr int C = kAoC;
  constexpr size_t ld = 3 * C;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int fsw = (fr >> 1) & 7;
  const int img = blockIdx.x;
  const T* base = qkv + (size_t)img * L * ld;
  auto stamp = [&](int point) {
    if constexpr (TRACE) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (blockIdx.x < kAoTraceBlocks && lane == 0) trace[(blockIdx.x * kAoWaves + wid) * 64 + point] = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  // attention role of this wave within a step: head 2 u + ah, query rows 16 amt .. 16 amt + 15
  const int ah = wid >> 2, amt = wid & 3;
  // LDS-DMA role: matrix (head dh of the step, K or V), rows 32 dhalf .. 32 dhalf + 31
  const int dmat = wid >> 1, dh = dmat >> 1, dv = dmat & 1, dhalf = wid & 1;

  // (addresses = a wave-uniform base + a 32-bit lane offset, so that the requests take the SGPR-base form: per-piece
  // 64-bit lane addresses, precomputed by hipcc for every step, were what pushed the kernel over 256 registers)
  const char* qkv_b = reinterpret_cast<const char*>(base);
  const unsigned drow = lane >> 3;  // row of the lane within an 8-row piece; the swizzle alternates with the piece's parity
  const unsigned doff[2] = {drow * (unsigned)(ld * 2) + (((lane & 7) ^ ((drow >> 1) & 7)) << 4),
                            drow * (unsigned)(ld * 2) + (((lane & 7) ^ ((4 + (drow >> 1)) & 7)) << 4)};
  auto dma_piece = [&](int u, int i) {  // 8 rows x 128 B of this wave's K / V matrix of unit u -> stage u % 3
    char* dst = kvp(u) + dmat * kAoRegion;
    const int jr = dhalf * 4 + i;
    const char* src = qkv_b + ((dv ? 2 * C : C) + (kAoStepHeads * u + dh) * kHeadDim) * 2 + (size_t)jr * 8 * ld * 2;
    if ((int)drow < L - jr * 8)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + doff[i & 1]), (lds_ptr_t)(dst + jr * 1024), 16, 0,
                                       OAKE_STREAM_AUX);
  };
  auto dma_step = [&](int u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(u, i);
  };
  vec8 qraw[2];  // Q rows as fetched (full lines); the lane swap into fragments happens where they are used
  unsigned qoff[2];
  {
    const int sw_row = fr & 7, sw_col = (fr & 8) * 4 + g * 8;
    int ra = amt * 16 + sw_row, rb = ra + 8;
    ra = ra < L ? ra : L - 1;
    rb = rb < L ? rb : L - 1;
    qoff[0] = (unsigned)ra * (unsigned)(ld * 2) + sw_col * 2;
    qoff[1] = (unsigned)rb * (unsigned)(ld * 2) + sw_col * 2;
  }
  auto load_q = [&](int u) {  // B operand of S^T = K Q^T: Q[16 amt + fr][32 kk + 8 g .. +8), fetched as full lines
    const char* qb = qkv_b + (kAoStepHeads * u + ah) * kHeadDim * 2;
    qraw[0] = stream_load16(reinterpret_cast<const vec8*>(qb + qoff[0]));
    qraw[1] = stream_load16(reinterpret_cast<const vec8*>(qb + qoff[1]));
  };
  // uniform base + 32-bit lane offset: the fetches take the SGPR-base addressing form (no 64-bit VGPR address each)
  const char* wbase = reinterpret_cast<const char*>(wperm) + (size_t)wid * kAoGroups * kAoNT * 1024;
  const unsigned wlane = lane * 16u;
  vec8 wf[kAoNT];  // a group's A fragments; fragment nt of the NEXT group is requested as soon as its MFMAs are issued
  auto load_w1 = [&](int G, int nt) {
    wf[nt] = *reinterpret_cast<const vec8*>(wbase + (size_t)(G * kAoNT + nt) * 1024 + wlane);
  };

  // ---- the attention of unit u (head 2 u + ah, query tile amt), cut into slices ------------------------------------
  // A slice is a handful of instructions placed between two 4-MFMA blocks of out_proj (sched_barrier on both sides):
  // the LDS reads of slice j are consumed in slice j + 1, and no more than two fragments are in flight — the kernel
  // lives at the 256-register limit (96 accumulators + 24 + 16 operand registers of out_proj).
  f32x4 sacc[4], oacc[4];
  vec8 pf[2], qf[2], kf[2], vf[2];
  float inv = 0.f, smx = 0.f, ssum = 0.f;
  auto k_read = [&](int u, int kt) {
    const char* ks = kvp(u) + (ah * 2) * kAoRegion;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      kf[kk] = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * 128 + (((kk * 4 + g) ^ fsw) << 4));
  };
  auto v_read = [&](int u, int dt) {  // V^T fragments of d-tile dt for both key halves (transposing reads)
    const char* vs = kvp(u) + (ah * 2 + 1) * kAoRegion;
    typedef s16x4 __attribute__((address_space(3))) * lds4_t;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const int row0 = 32 * ks2 + 4 * g + (fr >> 2);  // and row0 + 16: same swizzle
      const int vsw = (row0 >> 1) & 7;
      const int c4 = (fr & 3) * 4;
      const char* p0 = vs + row0 * 128 + (((dt * 2 + (c4 >> 3)) ^ vsw) << 4) + (c4 & 4) * 2;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 16 * 128));
      s16x8 both;
      both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
      both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
      vf[ks2] = __builtin_bit_cast(vec8, both);
    }
  };
  // slices of S^T[key][query] = K Q^T (j = 0 .. 4), of the softmax (0 .. 3), of O^T = V^T P^T (0 .. 4), of the store (0)
  auto qk_slice = [&](int u, int j) {
    if (j == 0) {
      qf[0] = swap_piece(qraw[0], qraw[1], true);
      qf[1] = swap_piece(qraw[1], qraw[0], false);
      k_read(u, 0);
    } else {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = T16<T>::mfma(kf[0], qf[0], a);
      sacc[j - 1] = T16<T>::mfma(kf[1], qf[1], a);
      if (j < 4) k_read(u, j);
    }
  };
  auto softmax_slice = [&](int j) {  // over the keys of a query: 16 in-register values + the four 16-lane rows
    if (j == 0) {
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sc = sacc[kt][i];
          sc = kt * 16 + 4 * g + i < L ? sc : -1e30f;  // padded keys
          sacc[kt][i] = sc;
          mx = fmaxf(mx, sc);
        }
      smx = rows16_max(mx);
    } else if (j == 1 || j == 2) {
      const float nb = -smx * kLog2eAo;
      float sum = j == 1 ? 0.f : ssum;
#pragma unroll
      for (int kt = 2 * (j - 1); kt < 2 * j; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][i], kLog2eAo, nb));
          sacc[kt][i] = p;
          sum += p;
        }
      ssum = sum;
    } else {
      inv = __builtin_amdgcn_rcpf(rows16_sum(ssum));  // (1 ulp; the product is rounded to 16 bits next)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        vec8 p8;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) p8[jj] = to16<T>(sacc[2 * ks2 + (jj >> 2)][jj & 3]);
        pf[ks2] = p8;
      }
    }
  };
  auto pv_slice = [&](int u, int j) {
    if (j == 0) {
      v_read(u, 0);
    } else {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = T16<T>::mfma(vf[0], pf[0], a);
      oacc[j - 1] = T16<T>::mfma(vf[1], pf[1], a);
      if (j < 4) v_read(u, j);
    }
  };
  auto attn_store = [&](int u) {  // the lane's accumulators = O[query fr][d = 16 dt + 4 g + j]: two B fragments
    char* ob = obp(u) + ((ah * 4 + amt) * 2) * 1024 + lane * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const f32x4 a = oacc[2 * kk], b = oacc[2 * kk + 1];
      const uint2 lo = pack4<T>(a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv);
      const uint2 hi = pack4<T>(b[0] * inv, b[1] * inv, b[2] * inv, b[3] * inv);
      *reinterpret_cast<uint4*>(ob + kk * 1024) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  };
  // slice `nt` (0 .. 5; -1 = before the group's first MFMA block) of part `i` of unit u's attention
  auto attn_slice = [&](int u, int i, int nt) {
    if (i == 0) {
      if (nt >= -1 && nt <= 3) qk_slice(u, nt + 1);
    } else if (i == 1) {
      if (nt >= 0 && nt <= 3) softmax_slice(nt);
    } else if (i == 2) {
      if (nt >= -1 && nt <= 3) pv_slice(u, nt + 1);
    } else if (nt == 0) {
      attn_store(u);
    }
  };

  // ---- prologue: K / V of units 0, 1 and 2, Q and W of the first groups in flight; unit 0's attention exposed -------
  dma_step(0);
  load_q(0);
  dma_step(1);
#pragma unroll
  for (int nt = 0; nt < kAoNT; ++nt) load_w1(0, nt);
  dma_step(2);
  // rows L .. 63 of every K / V region stay zero for the whole kernel (the DMA never touches them): a V tile reaches
  // them with P = 0 exactly, a K tile only produces scores the key mask discards
  for (int i = threadIdx.x; i < kAoStages * kAoStepHeads * 2 * (64 - L) * 8; i += kAoWaves * 64) {
    const int region = i / ((64 - L) * 8), off = i - region * ((64 - L) * 8);
    char* st = region / (kAoStepHeads * 2) == 0 ? kv0 : (region / (kAoStepHeads * 2) == 1 ? kv1 : kv2);
    *reinterpret_cast<uint4*>(st + (region % (kAoStepHeads * 2)) * kAoRegion + L * 128 + off * 16) =
        make_uint4(0u, 0u, 0u, 0u);
  }
  f32x4 acc[4][kAoNT];  // (first written by group 0's MFMAs: no registers held through the prologue)
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): this wave's pieces of units 0, 1 and 2 are in
  stamp(1);
  __syncthreads();
  stamp(2);
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    qk_slice(0, j);
    __builtin_amdgcn_sched_barrier(0);  // (slice by slice here too: all fragments at once do not fit the registers)
  }
  load_q(1);  // (unit 0's Q is in its scores)
#pragma unroll
  for (int j = 0; j < 4; ++j) softmax_slice(j);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    pv_slice(0, j);
    __builtin_amdgcn_sched_barrier(0);
  }
  attn_store(0);
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  stamp(3);
  __syncthreads();                     // O of unit 0 complete; stage 0 free
  stamp(4);

  // ---- steps: out_proj of unit s between the parts of unit s + 1's attention ---------------------------------------
  // Order of a step's vector-memory requests: vmcnt completes in order, so every wait for a W fragment (L2, a few
  // hundred cycles) also waits for whatever was requested BEFORE it — a K / V piece or Q rows from the fabric
  // (a thousand and more).  Each group therefore asks for its W fragments first and for one slow piece last: the
  // piece is then first waited for at the start of the group after the next, a whole group later.
  vec8 xres[4][3];  // the residual tile, requested during the last step (the attention registers are free by then)
  // (x addresses = the image's base, wave-uniform, + 32-bit lane offsets: see dma_piece)
  const char* xim = reinterpret_cast<const char*>(x + (size_t)img * L * C);
  const int swap_row = fr & 7;
  const unsigned swap_col = (64 * wid + (fr & 8) * 4 + 8 * g) * 2, hcol = (512 + 32 * wid + 8 * g) * 2;  // bytes
  auto xoff = [&](int row, unsigned colb) { return (unsigned)(row < L ? row : L - 1) * (unsigned)(C * 2) + colb; };
#pragma unroll
  for (int s = 0; s < kAoSteps; ++s) {
    const char* orow = obp(s) + lane * 16;
    const bool more = s + 1 < kAoSteps;
#pragma unroll
    for (int i = 0; i < kAoStepGroups; ++i) {
      const int G = kAoStepGroups * s + i;
      vec8 bf[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        bf[mt] = *reinterpret_cast<const vec8*>(orow + ((((i >> 1) * 4 + mt) * 2) + (i & 1)) * 1024);
      if (more) attn_slice(s + 1, i, -1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < kAoNT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][nt] = T16<T>::mfma(wf[nt], bf[mt], G == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt][nt]);
        if (G + 1 < kAoGroups) load_w1(G + 1, nt);  // into the registers just read: exactly one group of look-ahead
        if (more) attn_slice(s + 1, i, nt);         // unit s + 1's attention (its K / V arrived during earlier steps)
        __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the loads to just before their use: no look-ahead)
      }
      // the group's slow requests, after its W requests (see above)
      if (s + 3 < kAoSteps) dma_piece(s + 3, i);  // -> the stage unit s read (its attention ran in step s - 1)
      if (more && i == 0 && s + 2 < kAoSteps) load_q(s + 2);  // (unit s + 1's Q is in its scores)
      if (!more && i == 1) {  // row tiles 0 and 1 two groups before the end (2 and 3 at the start of the epilogue)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int ra = mt * 16 + swap_row;
          xres[mt][0] = *reinterpret_cast<const vec8*>(xim + xoff(ra, swap_col));
          xres[mt][1] = *reinterpret_cast<const vec8*>(xim + xoff(ra + 8, swap_col));
          xres[mt][2] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + fr, hcol));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(5 + 6 * s + i);
    }
    if (more) {
      // lgkmcnt(0): the O fragments are written.  vmcnt(7): everything but the step's last seven requests (six W
      // fragments, one K / V piece of unit s + 3) has arrived — in particular this wave's pieces of unit s + 2, whose
      // attention starts after the barrier
      __builtin_amdgcn_s_waitcnt(0x0077);
      stamp(5 + 6 * s + 4);
      __syncthreads();                     // O of unit s + 1 complete; every wave is done with unit s + 1's K / V stage
      stamp(5 + 6 * s + 5);
    }
  }

  // ---- epilogue: x[row, own columns] += acc + bias; (sum, sum^2) of every 64-column slice -> rowpart ---------------
  // lane (fr, g) owns, of row 16 mt + fr, the 8 columns ao_col0(wid, 2 t) + 8 g .. + 7 of pair t (tiles 2 t, 2 t + 1).
  // Pairs 0, 1 are the wave's own 128-byte line: memory is touched in full lines — piece A = (row 16 mt + (fr & 7),
  // 16-byte piece 4 (fr >> 3) + g), B = 8 rows below, lanes fr and fr ^ 8 swap one piece each (common.h).  Pair 2 is
  // half a line (the other half is the neighbour wave's): 16 rows x 64 bytes per instruction, plain stores.
#pragma unroll
  for (int mt = 2; mt < 4; ++mt) {
    const int ra = mt * 16 + swap_row;
    xres[mt][0] = *reinterpret_cast<const vec8*>(xim + xoff(ra, swap_col));
    xres[mt][1] = *reinterpret_cast<const vec8*>(xim + xoff(ra + 8, swap_col));
    xres[mt][2] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + fr, hcol));
  }
  float4 b0[3], b1[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int n = ao_col0(wid, 2 * t) + 8 * g;
    b0[t] = *reinterpret_cast<const float4*>(bias + n);
    b1[t] = *reinterpret_cast<const float4*>(bias + n + 4);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if (mt * 16 >= L) break;  // (uniform)
    const int m = mt * 16 + fr;
    const int ra = mt * 16 + swap_row, rb = ra + 8;
    vec8 xr[3];
    xr[2] = xres[mt][2];
    xr[0] = swap_piece(xres[mt][0], xres[mt][1], true);
    xr[1] = swap_piece(xres[mt][1], xres[mt][0], false);
    float ps1 = 0.f, ps2 = 0.f, ph1 = 0.f, ph2 = 0.f;
    u32x4_t qv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      f32x4 lo = acc[mt][2 * t], hi = acc[mt][2 * t + 1];
      lo[0] += b0[t].x; lo[1] += b0[t].y; lo[2] += b0[t].z; lo[3] += b0[t].w;
      hi[0] += b1[t].x; hi[1] += b1[t].y; hi[2] += b1[t].z; hi[3] += b1[t].w;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo[r] += to32<T>(xr[t][r]);
        hi[r] += to32<T>(xr[t][4 + r]);
        s1 += lo[r] + hi[r];
        s2 = fmaf(lo[r], lo[r], fmaf(hi[r], hi[r], s2));
      }
      if (t < 2) { ps1 += s1; ps2 += s2; } else { ph1 = s1; ph2 = s2; }
      const uint2 q0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]);
      const uint2 q1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
      qv[t] = u32x4_t{q0.x, q0.y, q1.x, q1.y};
    }
    const u32x4_t sa = swap_piece(qv[0], qv[1], true), sb = swap_piece(qv[1], qv[0], false);
    if (ra < L) store16_policy_s<1>(xim, xoff(ra, swap_col), sa);
    if (rb < L) store16_policy_s<1>(xim, xoff(rb, swap_col), sb);
    if (m < L) store16_policy_s<0>(xim, xoff(m, hcol), qv[2]);
    ps1 = rows16_sum(ps1);
    ps2 = rows16_sum(ps2);
    ph1 = rows16_sum(ph1);
    ph2 = rows16_sum(ph2);
    if (g == 0) {
      if (m < L) rowpart[((size_t)img * L + m) * 16 + wid] = make_float2(ps1, ps2);
      stat[m * kAoWaves + wid] = make_float2(ph1, ph2);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  stamp(41);
  __syncthreads();
  // slices 8 .. 11 = columns 512 + 64 j ..: the halves of waves 2 j and 2 j + 1, added in that order
  if (threadIdx.x < 256) {
    const int m = threadIdx.x & 63, j = threadIdx.x >> 6;
    if (m < L) {
      const float2 a = stat[m * kAoWaves + 2 * j], b = stat[m * kAoWaves + 2 * j + 1];
      rowpart[((size_t)img * L + m) * 16 + 8 + j] = make_float2(a.x + b.x, a.y + b.y);
    }
  }
  stamp(42);
}

template <typename T, bool TRACE>
hipError_t attn_out_launch_t(const void* qkv, const void* wperm, const float* bias, void* x, float* rowpart, int n,
                             int L, unsigned long long* trace, hipStream_t s) {
  OAKE_LAUNCH((attn_out_kernel<T, TRACE>), dim3(n), dim3(kAoWaves * 64), 0, s, reinterpret_cast<const T*>(qkv),
              reinterpret_cast<const T*>(wperm), bias, reinterpret_cast<T*>(x), reinterpret_cast<float2*>(rowpart), L,
              trace);
  return hipGetLastError();
}

}  // namespace

bool attn_out_supported(int L, int heads, int width) {
  return L >= 1 && L <= 64 && heads == kAoHeads && width == kAoC;
}

hipError_t launch_permute_out_w(int dtype16, const void* w, void* wp, hipStream_t s) {
  const int n = kAoC * kAoC;
  if (dtype16 == DT_BF16)
    hipLaunchKernelGGL(permute_out_w_kernel<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const bf16_t*>(w), reinterpret_cast<bf16_t*>(wp));
  else
    hipLaunchKernelGGL(permute_out_w_kernel<f16_t>, dim3((n + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const f16_t*>(w), reinterpret_cast<f16_t*>(wp));
  return hipGetLastError();
}

hipError_t launch_attn_out(int dtype16, const void* qkv, const void* wperm, const float* bias, void* x,
                           float* rowpart, int n, int L, hipStream_t s, unsigned long long* trace) {
  if (n <= 0) return hipSuccess;
  if (L < 1 || L > 64) return hipErrorInvalidValue;
  if (trace != nullptr)  // measurement: phase stamps of the first workgroups (f16 only)
    return attn_out_launch_t<f16_t, true>(qkv, wperm, bias, x, rowpart, n, L, trace, s);
  return dtype16 == DT_BF16 ? attn_out_launch_t<bf16_t, false>(qkv, wperm, bias, x, rowpart, n, L, nullptr, s)
                            : attn_out_launch_t<f16_t, false>(qkv, wperm, bias, x, rowpart, n, L, nullptr, s);
}

}  // namespace oake
