// attn_out.hip — multi-head self-attention + out_proj + residual of one transformer block in ONE kernel, for
// short sequences (L <= 64: encode_image at 224^2 / patch 32 and blocks mode, L = 50) on a 16-bit residual stream.
//
// Replaces two launches of the layer — `attention` (a pure memory round trip at L = 50: 59 MB of q/k/v in, 19.7 MB
// of attention output out) and `gemm_out_proj` (N = 768: 240 tiles for 256 CUs, one tile per CU, residual epilogue
// and end-of-kernel drain fully exposed) — and the `att` buffer between them:
//     x[n] += softmax(q k^T) v  W_out^T + b_out          [REF clip model.py ResidualAttentionBlock.attention, as called
//                                                         by encode_image: oadp/oake/globals.py:57, blocks.py:129]
// One workgroup per image (batch 256 = 256 CUs), 8 waves (two per SIMD, 256 registers each), in two ROLES:
//   * out_proj's K dimension is (head, d): its sum splits by head, so head h's contribution
//     O_h [L x 64] . W_out[:, 64 h .. 64 h + 63]^T is accumulated as soon as O_h exists.  A step = 2 heads.
//   * 2 attention waves (ids 2, 3 — with round-robin placement they share SIMDs 2, 3 with one out_proj wave each):
//     wave a owns head 2 u + a of step u, all four 16-row query tiles.  It fetches that head's K / V rows itself
//     (LDS-DMA into its own two-stage ring, source-side XOR swizzle as in gemm.hip) and its Q rows (registers), a
//     step ahead; S^T = K Q^T as in attention_pair_kernel (softmax in registers, P as the B operand of the PV
//     product through a permuted key enumeration, V by transposing LDS reads), the four query tiles interleaved so
//     that their dependent chains overlap; K and V fragments are read from LDS once per head, not once per tile.
//   * 6 out_proj waves: wave p owns output columns 128 p .. 128 p + 127 (two full 128-byte lines of every row) x all
//     64 (padded) rows: 4 x 8 accumulator tiles = 128 registers for the whole image.  While the attention waves work
//     on step u + 1 they issue the MFMAs of step u (4 groups of 32, 32 k-columns each); ONE workgroup barrier per
//     step publishes O of step u + 1.
//   * WHY roles: vmcnt completes in order.  The first forms of this kernel (every wave doing both jobs) waited, at
//     every W fragment (L2: a few hundred cycles), for whatever K / V / Q piece (fabric: ~3 k cycles) had been
//     requested before it — 38-41 us per launch against 36 for the two separate kernels, however the requests were
//     ordered (tools/attn_out_trace.py).  Now a wave has either only fast loads in flight or only slow ones.
//   * the attention output never leaves the chip and is never transposed: a lane's PV accumulators are, as they
//     stand, two B-operand fragments of the out_proj MFMA for a permuted k enumeration (k = 16 (i >> 2) + 4 g +
//     (i & 3) within a 32-column block), written to LDS as 1-KiB fragments (ds_write_b128, linear) and read back by
//     the out_proj waves (ds_read_b128, linear: no swizzle, no conflicts).  W_out is stored ONCE at load time in
//     exactly that fragment order (permute_out_w_kernel): every A-operand fetch is one fully coalesced 1-KiB load
//     per wave straight into registers (SGPR base + lane offset), issued two groups ahead of its use into the
//     registers its predecessor just left (sched_barrier keeps hipcc from sinking the loads back to their use).
//   * epilogue = EPI_RESID16 of gemm.hip: + bias + residual row (16-bit, in place; lane-swapped full 128-byte lines,
//     written through) and the (sum x, sum x^2) of the wave's two 64-column slices into rowpart — the LayerNorm
//     statistics the LN-folded c_fc GEMM that follows consumes.  The residual tile is requested
//     during the last two groups, when the W ring has drained.
// Arithmetic per image: out_proj 4 x 8 x 24 = 768 MFMAs per out_proj wave (rows padded 50 -> 64), attention 64 per
// attention wave and step; the two SIMDs with two out_proj waves issue 2 x 768 x 16 = 24.6 k cycles of MFMAs; W_out
// streams from the XCD's L2 at 1.18 MB per image = 18.4 k cycles of the CU's 64 B/clk vector-memory path; q/k/v
// 230 KB per image from the fabric.
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

constexpr int kAoWaves = 8;
constexpr int kAoHeads = 12;
constexpr int kAoC = kAoHeads * 64;                       // 768
constexpr int kAoSteps = kAoHeads / 2;                    // 6 steps of 2 heads
constexpr int kAoGroups = 2 * kAoHeads;                   // 32-column k groups of out_proj per image: 24 (4 per step)
constexpr int kAoNT = 8;                                  // 16-column tiles per out_proj wave (128 columns)
constexpr int kAoRegion = 64 * 128;                       // K or V of one head: 64 rows x 128 B (rows >= L zero)
constexpr int kAoStage = 2 * 3 * kAoRegion;               // both attention waves' Q, K and V of one step: 48 KB
constexpr int kAoObuf = 2 * 4 * 2 * 1024;                 // a step's O fragments: [head][query tile][kk] x 1 KiB = 16 KB
constexpr float kLog2eAo = 1.4426950408889634f;

// W_out [C, C] (row n = output feature, K contiguous) -> the A-operand fragments of attn_out_kernel, in fetch order:
// fragment (out_proj wave p, group G = 2 head + kk, column tile nt) is 64 lanes x 8 values = 1 KiB;
//   lane (r = l & 15, g = l >> 4), value i:  n = 128 p + 32 (nt >> 1) + 8 (r >> 2) + 4 (nt & 1) + (r & 3)
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
  const int n = 128 * p + 32 * (nt >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
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
  // module-LDS lowering attaches).  With the ring's stages as distinct variables (and the step loop fully unrolled, so
  // that each access names its stage statically) a K / V read of step u does not wait for the rows of step u + 1.
  __shared__ __attribute__((aligned(16))) char kv0[kAoStage], kv1[kAoStage];  // 2 x 48 KB
  __shared__ __attribute__((aligned(16))) char ob0[kAoObuf], ob1[kAoObuf];    // 2 x 16 KB
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
  // workgroup barrier WITHOUT __syncthreads' fence: the fence would wait for every request in flight — the attention
  // waves' rows of the NEXT unit, the out_proj waves' W fragments — where only LDS traffic has to be ordered: an
  // attention wave waits for its own ds_writes (lgkmcnt(0)) before it, an out_proj wave's reads of the previous
  // step were consumed by its MFMAs
#define OAKE_AO_BAR()                  \
  do {                                 \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

  if (wid == 2 || wid == 3) {
    // =================================== attention wave a: head 2 u + a of every step u ===========================
    const int a = wid - 2;
    const int fsw = (fr >> 1) & 7;
    // (addresses = a wave-uniform base + a 32-bit lane offset: the requests take the SGPR-base form)
    const char* qkv_b = reinterpret_cast<const char*>(qkv + (size_t)img * L * 3 * C);
    const unsigned drow = lane >> 3;  // row of the lane within an 8-row piece; the swizzle alternates with the piece's parity
    const unsigned doff[2] = {drow * ldb + (((lane & 7) ^ ((drow >> 1) & 7)) << 4),
                              drow * ldb + (((lane & 7) ^ ((4 + (drow >> 1)) & 7)) << 4)};
    auto dma_unit = [&](int u) {  // Q, K and V rows of head 2 u + a -> this wave's regions of stage u & 1
      char* dst = kvp(u) + a * 3 * kAoRegion;
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int jr = 0; jr < 8; ++jr) {
          const char* src = qkv_b + (v * C + (2 * u + a) * kHeadDim) * 2 + (size_t)jr * 8 * ldb;
          if (jr * 8 < L && (int)drow < L - jr * 8)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + doff[jr & 1]), (lds_ptr_t)(dst + v * kAoRegion + jr * 1024),
                                             16, 0, OAKE_STREAM_AUX);
        }
    };

    // rows L .. 63 of this wave's Q / K / V regions stay zero for the whole kernel (the DMA never touches them): a V
    // tile reaches them with P = 0 exactly, a K tile only produces scores the key mask discards, a Q tile rows of O
    // that are never stored
    for (int i = lane; i < 2 * 3 * (64 - L) * 8; i += 64) {
      const int region = i / ((64 - L) * 8), off = i - region * ((64 - L) * 8);  // (stage, Q | K | V)
      char* st = region >= 3 ? kv1 : kv0;
      *reinterpret_cast<uint4*>(st + (a * 3 + region % 3) * kAoRegion + L * 128 + off * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    dma_unit(0);

#pragma unroll
    for (int u = 0; u < kAoSteps; ++u) {
      // everything this wave has requested is in: Q / K / V of unit u (asked for a whole step ago)
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
      stamp(1 + 8 * u);
      const char* qs = kvp(u) + a * 3 * kAoRegion;
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
      if (u + 1 < kAoSteps) dma_unit(u + 1);  // the next unit's rows, a whole step ahead
      __builtin_amdgcn_sched_barrier(0);
      // S^T[key][query] = K Q^T for the four query tiles; the K fragments are read once and kept, Q streams through
      f32x4 sacc[4][4];
      {
        vec8 kf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            kf[kt][kk] = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * 128 + (((kk * 4 + g) ^ fsw) << 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vec8 qf[2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            qf[kk] = *reinterpret_cast<const vec8*>(qs + (q * 16 + fr) * 128 + (((kk * 4 + g) ^ fsw) << 4));
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            f32x4 c = T16<T>::mfma(kf[kt][0], qf[0], f32x4{0.f, 0.f, 0.f, 0.f});
            sacc[q][kt] = T16<T>::mfma(kf[kt][1], qf[1], c);
          }
        }
      }
      stamp(2 + 8 * u);
      // softmax over the keys of a query: 16 in-register values + the four 16-lane rows; four independent chains
      float inv[4];
      vec8 pf[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float sc = sacc[q][kt][i];
            if (kt * 16 + 16 > L) {  // (uniform) a key tile with padded keys
              sc = kt * 16 + 4 * g + i < L ? sc : -1e30f;
              sacc[q][kt][i] = sc;
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
            const float e = __builtin_amdgcn_exp2f(fmaf(sacc[q][kt][i], kLog2eAo, nb));
            sacc[q][kt][i] = e;
            sum += e;
          }
        inv[q] = __builtin_amdgcn_rcpf(rows16_sum(sum));  // (1 ulp; the product is rounded to 16 bits next)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          vec8 p8;
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[q][2 * ks2 + (j >> 2)][j & 3]);
          pf[q][ks2] = p8;
        }
      }
      stamp(3 + 8 * u);
      // O^T[d][query] = V^T P^T; the lane's accumulators = O[query fr][d = 16 dt + 4 g + j]: B fragments as they stand
      char* ob = obp(u) + (a * 4 * 2) * 1024 + lane * 16;
      f32x4 oacc[4][4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 c = T16<T>::mfma(vf[dt][0], pf[q][0], f32x4{0.f, 0.f, 0.f, 0.f});
          oacc[q][dt] = T16<T>::mfma(vf[dt][1], pf[q][1], c);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const f32x4 c = oacc[q][2 * kk], d = oacc[q][2 * kk + 1];
          const uint2 lo = pack4<T>(c[0] * inv[q], c[1] * inv[q], c[2] * inv[q], c[3] * inv[q]);
          const uint2 hi = pack4<T>(d[0] * inv[q], d[1] * inv[q], d[2] * inv[q], d[3] * inv[q]);
          *reinterpret_cast<uint4*>(ob + (q * 2 + kk) * 1024) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the fragments are written
      stamp(4 + 8 * u);
      OAKE_AO_BAR();                       // O of unit u complete (and the out_proj waves are done with unit u - 1)
      stamp(5 + 8 * u);
    }
    return;
  }

  // ===================================== out_proj wave p: output columns 128 p .. 128 p + 127 ========================
  const int p = wid < 2 ? wid : wid - 2;
  const char* wbase = reinterpret_cast<const char*>(wperm) + (size_t)p * kAoGroups * kAoNT * 1024;
  const unsigned wlane = lane * 16u;
  vec8 wf[2][kAoNT];  // ring of two groups; fragment nt of group G + 2 is requested as soon as group G's MFMAs on it are issued
  auto load_w1 = [&](int G, int nt) {
    // the base of each half group (4 fragments: the immediate offset reaches 4 KiB) is kept an opaque SGPR pair — left
    // to itself hipcc materialises a 64-bit VGPR address per 4 KiB of W for the whole unrolled kernel and spills them
    const char* b = wbase + (size_t)(G * kAoNT + (nt & ~3)) * 1024;
    asm volatile("" : "+s"(b));
    typedef const __attribute__((address_space(1))) char* gchar_t;  // (the asm hides that this is global memory)
    typedef const __attribute__((address_space(1))) vec8* gvec8_t;
    wf[G & 1][nt] = *(gvec8_t)((gchar_t)b + (nt & 3) * 1024 + wlane);
  };
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int nt = 0; nt < kAoNT; ++nt) load_w1(G, nt);
  // residual tile: lane (fr, g) owns, of row 16 mt + fr, the 8 columns 128 p + 32 t + 8 g .. + 7 of pair t = 0 .. 3
  // (tiles 2 t, 2 t + 1); memory is touched in full lines — piece A = (row 16 mt + (fr & 7), 16-byte piece 4 (fr >> 3)
  // + g of line t >> 1), B = 8 rows below — and lanes fr and fr ^ 8 swap one piece each (common.h)
  const char* xim = reinterpret_cast<const char*>(x + (size_t)img * L * C);
  const int swap_row = fr & 7;
  const unsigned swap_col = (128 * p + (fr & 8) * 4 + 8 * g) * 2;  // bytes; + 128 for the second line
  // (the row is made opaque at each use: hipcc otherwise computes all these offsets up front and spills them)
  auto xoff = [&](int row, unsigned colb) {
    asm volatile("" : "+v"(row));
    return (unsigned)(row < L ? row : L - 1) * (unsigned)(C * 2) + colb;
  };
  vec8 xres[4][4];  // [row tile][line * 2 + (A | B)]
  auto load_x = [&](int mt) {
#pragma unroll
    for (int ln = 0; ln < 2; ++ln) {
      xres[mt][ln * 2 + 0] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + swap_row, swap_col + ln * 128));
      xres[mt][ln * 2 + 1] = *reinterpret_cast<const vec8*>(xim + xoff(mt * 16 + swap_row + 8, swap_col + ln * 128));
    }
  };
  f32x4 acc[4][kAoNT];  // (first written by group 0's MFMAs)
  stamp(1);
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
          acc[mt][nt] = T16<T>::mfma(wf[G & 1][nt], bf[mt], G == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt][nt]);
        if (G + 2 < kAoGroups) load_w1(G + 2, nt);  // into the registers just read: two groups of look-ahead
        else if (G + 1 == kAoGroups && (nt & 3) == 3) load_x(nt >> 2);  // the ring has drained: residual row tiles 0, 1
        __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the loads to just before their use: no look-ahead)
      }
      if (i == 3) stamp(3 + 5 * s + i);
    }
    if (s + 1 < kAoSteps) {
      OAKE_AO_BAR();  // O of unit s + 1
      stamp(3 + 5 * s + 4);
    }
  }

  // ---- epilogue: x[row, own columns] += acc + bias; (sum, sum^2) of the two 64-column slices -> rowpart -------------
  // (line by line, row tile by row tile, sched_barrier between them: everything at once does not fit the registers)
  load_x(2);
  load_x(3);
  int ge = lane;  // (recomputed here, opaque: hipcc otherwise keeps 8 g from the prologue in a spilled register)
  asm volatile("" : "+v"(ge));
  ge >>= 4;
#pragma unroll
  for (int ln = 0; ln < 2; ++ln) {
    float4 b0[2], b1[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int n = 128 * p + 64 * ln + 32 * tt + 8 * ge;
      b0[tt] = *reinterpret_cast<const float4*>(bias + n);
      b1[tt] = *reinterpret_cast<const float4*>(bias + n + 4);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt * 16 >= L) break;  // (uniform)
      const int m = mt * 16 + fr;
      const int ra = mt * 16 + swap_row, rb = ra + 8;
      vec8 xr[2];
      xr[0] = swap_piece(xres[mt][ln * 2], xres[mt][ln * 2 + 1], true);
      xr[1] = swap_piece(xres[mt][ln * 2 + 1], xres[mt][ln * 2], false);
      float ps1 = 0.f, ps2 = 0.f;
      u32x4_t qv[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = ln * 2 + tt;
        f32x4 lo = acc[mt][2 * t], hi = acc[mt][2 * t + 1];
        lo[0] += b0[tt].x; lo[1] += b0[tt].y; lo[2] += b0[tt].z; lo[3] += b0[tt].w;
        hi[0] += b1[tt].x; hi[1] += b1[tt].y; hi[2] += b1[tt].z; hi[3] += b1[tt].w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] += to32<T>(xr[tt][r]);
          hi[r] += to32<T>(xr[tt][4 + r]);
          ps1 += lo[r] + hi[r];
          ps2 = fmaf(lo[r], lo[r], fmaf(hi[r], hi[r], ps2));
        }
        const uint2 q0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]);
        const uint2 q1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
        qv[tt] = u32x4_t{q0.x, q0.y, q1.x, q1.y};
      }
      const u32x4_t sa = swap_piece(qv[0], qv[1], true), sb = swap_piece(qv[1], qv[0], false);
      if (ra < L) store16_policy_s<1>(xim, xoff(ra, swap_col + ln * 128), sa);
      if (rb < L) store16_policy_s<1>(xim, xoff(rb, swap_col + ln * 128), sb);
      ps1 = rows16_sum(ps1);
      ps2 = rows16_sum(ps2);
      if (g == 0 && m < L) rowpart[((size_t)img * L + m) * 16 + 2 * p + ln] = make_float2(ps1, ps2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  stamp(40);
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
  if (trace != nullptr)  // measurement: phase stamps of the first workgroups (f16 only)
    return attn_out_launch_t<f16_t, true>(qkv, wperm, bias, x, rowpart, n, L, trace, s);
  return dtype16 == DT_BF16 ? attn_out_launch_t<bf16_t, false>(qkv, wperm, bias, x, rowpart, n, L, nullptr, s)
                            : attn_out_launch_t<f16_t, false>(qkv, wperm, bias, x, rowpart, n, L, nullptr, s);
}

}  // namespace oake
