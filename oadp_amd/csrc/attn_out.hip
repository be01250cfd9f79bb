// attn_out.hip — multi-head self-attention + out_proj + residual of one transformer block in ONE kernel, for
// short sequences (L <= 64: encode_image at 224^2 / patch 32 and blocks mode, L = 50) on a 16-bit residual stream.
//
// Replaces two launches of the layer — `attention` (a pure memory round trip at L = 50: 59 MB of q/k/v in, 19.7 MB
// of attention output out) and `gemm_out_proj` (N = 768: 240 tiles for 256 CUs, one tile per CU, residual epilogue
// and end-of-kernel drain fully exposed) — and the `att` buffer between them:
//     x[n] += softmax(q k^T) v  W_out^T + b_out          [REF clip model.py ResidualAttentionBlock.attention, as called
//                                                         by encode_image: oadp/oake/globals.py:57, blocks.py:129]
// One workgroup per image (batch 256 = 256 CUs), 12 waves (three per SIMD, 168 registers each), in two ROLES:
//   * out_proj's K dimension is (head, d): its sum splits by head, so head h's contribution
//     O_h [L x 64] . W_out[:, 64 h .. 64 h + 63]^T is accumulated as soon as O_h exists.  A step = 2 heads.
//   * 4 attention waves (ids 8 .. 11: one per SIMD under round-robin placement): a pair of waves owns one head of
//     the step, a wave two of its four 16-row query tiles.  The pair fetches the head's Q / K / V rows by LDS-DMA
//     (source-side XOR swizzle as in gemm.hip; every other 8-row piece each) into a two-stage ring, a whole step
//     ahead; S^T = K Q^T as in attention_pair_kernel (softmax in registers, P as the B operand of the PV product
//     through a permuted key enumeration, V by transposing LDS reads), the wave's two query tiles interleaved.
//   * 8 out_proj waves (ids 0 .. 7: two per SIMD): wave p owns 96 output columns — one full 128-byte line of every
//     row (columns 64 p ..) and half of one of the last four lines (columns 512 + 32 p ..) — x all 64 (padded) rows:
//     4 x 6 accumulator tiles = 96 registers for the whole image.  While the attention waves work on step u + 1
//     they issue the MFMAs of step u (4 groups of 24, 32 k-columns each); ONE workgroup barrier per step publishes
//     O of step u + 1 and the rows of step u + 2.  Every SIMD issues 2 x 96 + 32 = 224 MFMAs per step.
//   * WHY roles: vmcnt completes in order.  The first forms of this kernel (every wave doing both jobs) waited, at
//     every W fragment (L2: a few hundred cycles), for whatever K / V / Q piece (fabric: ~3 k cycles) had been
//     requested before it — 38-41 us per launch against 36 for the two separate kernels, however the requests were
//     ordered (tools/attn_out_trace.py).  Now a wave has either only fast loads in flight or only slow ones.  (Two
//     attention waves for six out_proj waves, the second form, left the attention the critical path: 8 k cycles
//     per step — 21 LDS-DMA issues and four softmaxes per wave — for 4 k of out_proj.)
//   * the attention output never leaves the chip and is never transposed: a lane's PV accumulators are, as they
//     stand, two B-operand fragments of the out_proj MFMA for a permuted k enumeration (k = 16 (i >> 2) + 4 g +
//     (i & 3) within a 32-column block), written to LDS as 1-KiB fragments (ds_write_b128, linear) and read back by
//     the out_proj waves (ds_read_b128, linear: no swizzle, no conflicts).  W_out is stored ONCE at load time in
//     exactly that fragment order (permute_out_w_kernel): every A-operand fetch is one fully coalesced 1-KiB load
//     per wave straight into registers (SGPR base + lane offset), issued ahead of its use into the registers its
//     predecessor just left (sched_barrier keeps hipcc from sinking the loads back to their use).
//   * epilogue = EPI_RESID16 of gemm.hip: + bias + residual row (16-bit, in place; lane-swapped full 128-byte lines,
//     written through, for the wave's own line) and the (sum x, sum x^2) of every 64-column slice into rowpart — the
//     LayerNorm statistics the LN-folded c_fc GEMM that follows consumes; the four slices whose
//     halves belong to two waves are added up through LDS in a fixed order.
// Arithmetic per image: 224 x 6 steps x 16 cycles = 21.5 k cycles of MFMA issue per SIMD (rows padded 50 -> 64); W_out
// streams from the XCD's L2 at 1.18 MB per image = 18.4 k cycles of the CU's 64 B/clk vector-memory path; q/k/v
// 230 KB per image from the fabric.
#include "common.h"
#include "kernels.h"

#ifndef OAKE_LAB
#define OAKE_LAB 0  // 1: liboake_hip_lab.so (adds the s_memtime-stamped measurement build of the kernel)
#endif

namespace oake {

namespace {

constexpr int kAoWaves = 12;
constexpr int kAoPW = 8;                                  // out_proj waves (ids 0 .. 7); the other four do the attention
constexpr int kAoHeads = 12;
constexpr int kAoC = kAoHeads * 64;                       // 768
constexpr int kAoSteps = kAoHeads / 2;                    // 6 steps of 2 heads
constexpr int kAoGroups = 2 * kAoHeads;                   // 32-column k groups of out_proj per image: 24 (4 per step)
constexpr int kAoNT = 6;                                  // 16-column tiles per out_proj wave: 4 (its line) + 2 (its half line)
constexpr int kAoRegion = 64 * 128;                       // Q, K or V of one head: 64 rows x 128 B (rows >= L zero)
constexpr int kAoStage = 2 * 3 * kAoRegion;               // Q, K and V of a step's two heads: 48 KB
constexpr int kAoObuf = 2 * 4 * 2 * 1024;                 // a step's O fragments: [head][query tile][kk] x 1 KiB = 16 KB
constexpr float kLog2eAo = 1.4426950408889634f;
// measurement builds (wrong results): -DOAKE_AO_ABLATE=1 no W fetches after the first group, =2 no Q / K / V fetches
// after the first unit, =3 both — what the kernel costs without its two request streams
#ifndef OAKE_AO_ABLATE
#define OAKE_AO_ABLATE 0
#endif

// first output column of tile nt of out_proj wave p: its own line (tiles 0..3) or its half of line 8 + (p >> 1)
__host__ __device__ constexpr int ao_col0(int p, int nt) { return nt < 4 ? 64 * p + 32 * (nt >> 1) : 512 + 32 * p; }

// W_out [C, C] (row n = output feature, K contiguous) -> the A-operand fragments of attn_out_kernel, in fetch order:
// fragment (out_proj wave p, group G = 2 head + kk, column tile nt) is 64 lanes x 8 values = 1 KiB;
//   lane (r = l & 15, g = l >> 4), value i:  n = ao_col0(p, nt) + 8 (r >> 2) + 4 (nt & 1) + (r & 3)
//                                            k = 32 G + 16 (i >> 2) + 4 g + (i & 3)
// (n: the pair interleave of gemm.hip's 16-bit epilogues — a lane's accumulators of tiles 2t, 2t+1 are 8 consecutive
// columns; k: the enumeration in which a lane's PV accumulators are B fragments as they stand.)
template <typename T>
__global__ void permute_out_w_kernel(const T* __restrict__ w, T* __restrict__ wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one output value
  if (idx >= kAoC * kAoC) return;
  const int i = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
  const int nt = frag % kAoNT, G = (frag / kAoNT) % kAoGroups, p = frag / (kAoNT * kAoGroups);
  const int r = lane & 15, g = lane >> 4;
  const int n = ao_col0(p, nt) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
  const int k = 32 * G + 16 * (i >> 2) + 4 * g + (i & 3);
  wp[idx] = w[(size_t)n * kAoC + k];
}

// TRACE (measurement builds of the kernel only, oake_debug_attn_out_trace): s_memtime stamps of every wave of the
// first kAoTraceBlocks workgroups at the phase boundaries, trace[(block * 12 + wave) * 64 + point]
constexpr int kAoTraceBlocks = 4;

template <typename T, bool TRACE>
__global__ __launch_bounds__(kAoWaves * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void attn_out_kernel(const T* __restrict__ qkv, const T* __restrict__ wperm, const float* __restrict__ bias,
                     T* __restrict__ x, float2* __restrict__ rowpart, int L, unsigned long long* __restrict__ trace) {
  typedef typename T16<T>::vec8 vec8;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  // Separate LDS objects, not one dynamic array: hipcc guards every DS read that MAY alias an LDS-DMA in flight with
  // a vmcnt wait for it, and it tells accesses apart by the LDS variable they belong to (the alias scopes the
  // module-LDS lowering attaches).  With the ring's stages as distinct variables (and the step loop fully unrolled, so
  // that each access names its stage statically) a K / Q read of step u does not wait for the rows of step u + 1.
  __shared__ __attribute__((aligned(16))) char kv0[kAoStage], kv1[kAoStage];  // 2 x 48 KB
  __shared__ __attribute__((aligned(16))) char ob0[kAoObuf], ob1[kAoObuf];    // 2 x 16 KB
  __shared__ float2 stat[64 * kAoPW];                                         // 4 KB
  auto kvp = [&](int u) -> char* { return (u & 1) ? kv1 : kv0; };
  auto obp = [&](int u) -> char* { return (u & 1) ? ob1 : ob0; };
  constexpr int C = kAoC;
  constexpr unsigned ldb = 3 * C * 2;  // bytes per qkv row
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int img = blockIdx.x;
  auto stamp = [&](int point) {
    if constexpr (TRACE) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (blockIdx.x < kAoTraceBlocks && lane == 0) trace[(blockIdx.x * kAoWaves + wid) * 64 + point] = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  // workgroup barrier WITHOUT __syncthreads' fence: the fence would wait for every request in flight (the out_proj
  // waves' W fragments) where only LDS traffic has to be ordered: an attention wave waits for its own ds_writes and
  // LDS-DMA pieces before it, an out_proj wave's reads of the previous step were consumed by its MFMAs
#define OAKE_AO_BAR()                  \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

  if (wid >= kAoPW) {
    // ============== attention wave: head 2 u + h2 of every step u, query tiles 2 half and 2 half + 1 ==============
    const int h2 = (wid - kAoPW) >> 1, half = (wid - kAoPW) & 1;
    const int fsw = (fr >> 1) & 7;
    // (addresses = a wave-uniform base + a 32-bit lane offset: the requests take the SGPR-base form)
    const char* qkv_b = reinterpret_cast<const char*>(qkv + (size_t)img * L * 3 * C);
    const unsigned drow = lane >> 3;  // row of the lane within an 8-row piece
    // this wave moves the 8-row pieces of parity `half` of its head's three matrices; the 16-byte chunks of a row are
    // XOR-swizzled by (row >> 1) & 7 on the source side (gemm.hip): 4 (piece & 1) + (drow >> 1)
    const unsigned doff = drow * ldb + (((lane & 7) ^ ((4 * half + (drow >> 1)) & 7)) << 4);
    auto dma_unit = [&](int u) {
      char* dst = kvp(u) + h2 * 3 * kAoRegion;
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          const int jr = 2 * j2 + half;
          const char* src = qkv_b + (v * C + (2 * u + h2) * kHeadDim) * 2 + (size_t)jr * 8 * ldb;
          if (jr * 8 < L && (int)drow < L - jr * 8)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + doff), (lds_ptr_t)(dst + v * kAoRegion + jr * 1024), 16, 0,
                                             OAKE_STREAM_AUX);
        }
    };
    // rows L .. 63 of the head's Q / K / V regions stay zero for the whole kernel (the DMA never touches them; this
    // wave clears stage `half`, its partner the other): a V tile reaches them with P = 0 exactly, a K tile only
    // produces scores the key mask discards, a Q tile rows of O that are never stored
    for (int i = lane; i < 3 * (64 - L) * 8; i += 64) {
      const int region = i / ((64 - L) * 8), off = i - region * ((64 - L) * 8);
      char* st = half ? kv1 : kv0;
      *reinterpret_cast<uint4*>(st + (h2 * 3 + region) * kAoRegion + L * 128 + off * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    dma_unit(0);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): this wave's pieces of unit 0 and its zero rows are in
    stamp(1);
    OAKE_AO_BAR();  // ... and its partner's
#pragma unroll
    for (int u = 0; u < kAoSteps; ++u) {
      stamp(2 + 8 * u);
      const char* qs = kvp(u) + h2 * 3 * kAoRegion;
      const char* ks = qs + kAoRegion;
      const char* vs = ks + kAoRegion;
      // V fragments first (transposing reads; key enumeration of the score registers), BEFORE the next unit's rows are
      // requested: hipcc cannot tell which LDS object ds_read_tr16_b64 reads and guards it with a wait for every
      // LDS-DMA in flight (the plain ds_read_b128 of K and Q carry their variable's alias scope and are not guarded)
      vec8 vf[4][2];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          const int row0 = 32 * ks2 + 4 * g + (fr >> 2);  // and row0 + 16: same swizzle
          const int vsw = (row0 >> 1) & 7;
          const int c4 = (fr & 3) * 4;
          const char* p0 = vs + row0 * 128 + (((dt * 2 + (c4 >> 3)) ^ vsw) << 4) + (c4 & 4) * 2;
          typedef s16x4 __attribute__((address_space(3))) * lds4_t;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 16 * 128));
          s16x8 both;
          both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
          both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
          vf[dt][ks2] = __builtin_bit_cast(vec8, both);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (u + 1 < kAoSteps && !(OAKE_AO_ABLATE & 2)) dma_unit(u + 1);  // the next unit's rows, a whole step ahead
      __builtin_amdgcn_sched_barrier(0);
      // S^T[key][query] = K Q^T for the wave's two query tiles; the K fragments are read once and kept
      f32x4 sacc[2][4];
      {
        vec8 kf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            kf[kt][kk] = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * 128 + (((kk * 4 + g) ^ fsw) << 4));
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          vec8 qf[2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            qf[kk] = *reinterpret_cast<const vec8*>(qs + ((2 * half + q2) * 16 + fr) * 128 + (((kk * 4 + g) ^ fsw) << 4));
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            f32x4 c = T16<T>::mfma(kf[kt][0], qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            sacc[q2][kt] = T16<T>::mfma(kf[kt][1], qf[1], c);
          }
        }
      }
      stamp(3 + 8 * u);
      // softmax over the keys of a query: 16 in-register values + the four 16-lane rows; two independent chains
      float inv[2];
      vec8 pf[2][2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float sc = sacc[q2][kt][i];
            if (kt * 16 + 16 > L) {  // (uniform) a key tile with padded keys
              sc = kt * 16 + 4 * g + i < L ? sc : -1e30f;
              sacc[q2][kt][i] = sc;
            }
            mx = fmaxf(mx, sc);
          }
        mx = rows16_max(mx);
        const float nb = -mx * kLog2eAo;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sacc[q2][kt][i], kLog2eAo, nb));
            sacc[q2][kt][i] = e;
            sum += e;
          }
        inv[q2] = __builtin_amdgcn_rcpf(rows16_sum(sum));  // (1 ulp; the product is rounded to 16 bits next)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          vec8 p8;
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[q2][2 * ks2 + (j >> 2)][j & 3]);
          pf[q2][ks2] = p8;
        }
      }
      stamp(4 + 8 * u);
      // O^T[d][query] = V^T P^T; the lane's accumulators = O[query fr][d = 16 dt + 4 g + j]: B fragments as they stand
      char* ob = obp(u) + ((h2 * 4 + 2 * half) * 2) * 1024 + lane * 16;
      f32x4 oacc[2][4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          f32x4 c = T16<T>::mfma(vf[dt][0], pf[q2][0], f32x4{0.f, 0.f, 0.f, 0.f});
          oacc[q2][dt] = T16<T>::mfma(vf[dt][1], pf[q2][1], c);
        }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const f32x4 c = oacc[q2][2 * kk], d = oacc[q2][2 * kk + 1];
          const uint2 lo = pack4<T>(c[0] * inv[q2], c[1] * inv[q2], c[2] * inv[q2], c[3] * inv[q2]);
          const uint2 hi = pack4<T>(d[0] * inv[q2], d[1] * inv[q2], d[2] * inv[q2], d[3] * inv[q2]);
          *reinterpret_cast<uint4*>(ob + (q2 * 2 + kk) * 1024) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      stamp(5 + 8 * u);
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the fragments are written, this wave's pieces of unit u + 1 are in
      stamp(6 + 8 * u);
      OAKE_AO_BAR();  // O of unit u and the rows of unit u + 1 complete (the out_proj waves are done with unit u - 1)
    }
    return;
  }

  // ============= out_proj wave p: output columns 64 p .. 64 p + 63 and 512 + 32 p .. 512 + 32 p + 31 ================
  const int p = wid;
  const char* wbase = reinterpret_cast<const char*>(wperm) + (size_t)p * kAoGroups * kAoNT * 1024;
  const unsigned wlane = lane * 16u;
  vec8 wf[kAoNT];  // a group's A fragments; fragment nt of group G + 1 is requested as soon as group G's MFMAs on it are issued
  auto load_w1 = [&](int G, int nt) {
    // the base of each half group (3 fragments: the immediate offset reaches 4 KiB) is kept an opaque SGPR pair — left
    // to itself hipcc materialises a 64-bit VGPR address per 4 KiB of W for the whole unrolled kernel and spills them
    const char* b = wbase + (size_t)(G * kAoNT + (nt >= 3 ? 3 : 0)) * 1024;
    asm volatile("" : "+s"(b));
    typedef const __attribute__((address_space(1))) char* gchar_t;  // (the asm hides that this is global memory)
    typedef const __attribute__((address_space(1))) vec8* gvec8_t;
    wf[nt] = *(gvec8_t)((gchar_t)b + (nt % 3) * 1024 + wlane);
  };
#pragma unroll
  for (int nt = 0; nt < kAoNT; ++nt) load_w1(0, nt);
  // residual tile: lane (fr, g) owns, of row 16 mt + fr, the 8 columns ao_col0(p, 2 t) + 8 g .. + 7 of pair t (tiles
  // 2 t, 2 t + 1).  Pairs 0, 1 are the wave's own 128-byte line: memory is touched in full lines — piece A = (row
  // 16 mt + (fr & 7), 16-byte piece 4 (fr >> 3) + g), B = 8 rows below, lanes fr and fr ^ 8 swap one piece each
  // (common.h).  Pair 2 is half a line (the other half is the neighbour wave's): 16 rows x 64 bytes per instruction.
  f32x4 acc[4][kAoNT];  // (first written by group 0's MFMAs)
  stamp(1);
  OAKE_AO_BAR();  // (the attention waves' rows of unit 0)
  OAKE_AO_BAR();  // O of unit 0
  stamp(2);
#pragma unroll
  for (int s = 0; s < kAoSteps; ++s) {
    const char* orow = obp(s) + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int G = 4 * s + i;
      vec8 bf[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        bf[mt] = *reinterpret_cast<const vec8*>(orow + ((((i >> 1) * 4 + mt) * 2) + (i & 1)) * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < kAoNT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][nt] = T16<T>::mfma(wf[nt], bf[mt], G == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt][nt]);
        if (G + 1 < kAoGroups && !(OAKE_AO_ABLATE & 1)) load_w1(G + 1, nt);  // into the registers just read: one group of look-ahead (three waves per SIMD)
        __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the loads to just before their use: no look-ahead)
      }
    }
    stamp(3 + 2 * s);
    if (s + 1 < kAoSteps) {
      OAKE_AO_BAR();  // O of unit s + 1
      stamp(4 + 2 * s);
    }
  }

  // ---- epilogue: x[row, own columns] += acc + bias; (sum, sum^2) of every 64-column slice -> rowpart ---------------
  // Two passes over the row tiles — the wave's half line (pair 2) first: its 32 accumulators go, then its line (pairs
  // 0, 1) — each row tile between
  // sched_barriers with the next tile's residual pieces requested one tile ahead: everything at once does not fit 168
  // registers beside the 96 accumulators.
  // (everything lane-dependent of the epilogue is derived HERE from an opaque copy of the lane id: hipcc otherwise
  // computes these offsets in the prologue and keeps them in spilled registers through the whole kernel)
  int le = lane;
  asm volatile("" : "+v"(le));
  const int fre = le & 15, ge = le >> 4;
  const char* xim = reinterpret_cast<const char*>(x + (size_t)img * L * C);
  const int swap_row = fre & 7;
  const unsigned swap_col = (64 * p + (fre & 8) * 4 + 8 * ge) * 2, hcol = (512 + 32 * p + 8 * ge) * 2;  // bytes
  // (the row is made opaque at each use too)
  auto xoff = [&](int row, unsigned colb) {
    asm volatile("" : "+v"(row));
    return (unsigned)(row < L ? row : L - 1) * (unsigned)(C * 2) + colb;
  };
  vec8 xln[4][2], xhl[4];  // residual pieces: [row tile][A | B] of the wave's line, [row tile] of its half line
  auto load_xl = [&](int mt) {
    xln[mt][0] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + swap_row, swap_col));
    xln[mt][1] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + swap_row + 8, swap_col));
  };
  auto load_xh = [&](int mt) { xhl[mt] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + fre, hcol)); };
  load_xh(0);
  load_xl(0);
  load_xl(1);
  {
    const int n = 512 + 32 * p + 8 * ge;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt * 16 >= L) break;  // (uniform)
      if (mt + 1 < 4) load_xh(mt + 1);
      const int m = mt * 16 + fre;
      f32x4 lo = acc[mt][4], hi = acc[mt][5];
      lo[0] += b0.x; lo[1] += b0.y; lo[2] += b0.z; lo[3] += b0.w;
      hi[0] += b1.x; hi[1] += b1.y; hi[2] += b1.z; hi[3] += b1.w;
      float ph1 = 0.f, ph2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo[r] += to32<T>(xhl[mt][r]);
        hi[r] += to32<T>(xhl[mt][4 + r]);
        ph1 += lo[r] + hi[r];
        ph2 = fmaf(lo[r], lo[r], fmaf(hi[r], hi[r], ph2));
      }
      const uint2 q0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]);
      const uint2 q1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
      if (m < L) store16_policy_s<0>(xim, xoff(m, hcol), u32x4_t{q0.x, q0.y, q1.x, q1.y});
      ph1 = rows16_sum(ph1);
      ph2 = rows16_sum(ph2);
      if (ge == 0) stat[m * kAoPW + p] = make_float2(ph1, ph2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    float4 b0[2], b1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = 64 * p + 32 * t + 8 * ge;
      b0[t] = *reinterpret_cast<const float4*>(bias + n);
      b1[t] = *reinterpret_cast<const float4*>(bias + n + 4);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt * 16 >= L) break;  // (uniform)
      if (mt >= 1 && mt + 1 < 4) load_xl(mt + 1);
      const int m = mt * 16 + fre;
      const int ra = mt * 16 + swap_row, rb = ra + 8;
      vec8 xr[2];
      xr[0] = swap_piece(xln[mt][0], xln[mt][1], true);
      xr[1] = swap_piece(xln[mt][1], xln[mt][0], false);
      float ps1 = 0.f, ps2 = 0.f;
      u32x4_t qv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 lo = acc[mt][2 * t], hi = acc[mt][2 * t + 1];
        lo[0] += b0[t].x; lo[1] += b0[t].y; lo[2] += b0[t].z; lo[3] += b0[t].w;
        hi[0] += b1[t].x; hi[1] += b1[t].y; hi[2] += b1[t].z; hi[3] += b1[t].w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] += to32<T>(xr[t][r]);
          hi[r] += to32<T>(xr[t][4 + r]);
          ps1 += lo[r] + hi[r];
          ps2 = fmaf(lo[r], lo[r], fmaf(hi[r], hi[r], ps2));
        }
        const uint2 q0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]);
        const uint2 q1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
        qv[t] = u32x4_t{q0.x, q0.y, q1.x, q1.y};
      }
      const u32x4_t sa = swap_piece(qv[0], qv[1], true), sb = swap_piece(qv[1], qv[0], false);
      if (ra < L) store16_policy_s<1>(xim, xoff(ra, swap_col), sa);
      if (rb < L) store16_policy_s<1>(xim, xoff(rb, swap_col), sb);
      ps1 = rows16_sum(ps1);
      ps2 = rows16_sum(ps2);
      if (ge == 0 && m < L) rowpart[((size_t)img * L + m) * 16 + p] = make_float2(ps1, ps2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  stamp(40);
  OAKE_AO_BAR();  // (the attention waves have left: the eight out_proj waves)
  // slices 8 .. 11 = columns 512 + 64 j ..: the halves of waves 2 j and 2 j + 1, added in that order
  if (threadIdx.x < 256) {
    const int m = threadIdx.x & 63, j = threadIdx.x >> 6;
    if (m < L) {
      const float2 a = stat[m * kAoPW + 2 * j], b = stat[m * kAoPW + 2 * j + 1];
      rowpart[((size_t)img * L + m) * 16 + 8 + j] = make_float2(a.x + b.x, a.y + b.y);
    }
  }
  stamp(41);
}

#undef OAKE_AO_BAR

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
  if (trace != nullptr) {  // measurement: phase stamps of the first workgroups (f16 only; lab build)
#if OAKE_LAB
    return attn_out_launch_t<f16_t, true>(qkv, wperm, bias, x, rowpart, n, L, trace, s);
#else
    return hipErrorInvalidValue;
#endif
  }
  return dtype16 == DT_BF16 ? attn_out_launch_t<bf16_t, false>(qkv, wperm, bias, x, rowpart, n, L, nullptr, s)
                            : attn_out_launch_t<f16_t, false>(qkv, wperm, bias, x, rowpart, n, L, nullptr, s);
}

}  // namespace oake
