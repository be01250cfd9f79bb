// attention.hip — multi-head self-attention for the ViT token sequences of OAKE
// (L = 50 for encode_image, L = 197 in objects mode), SURVEY.md §8 A15e and A11.
//
// attention_kernel: one wavefront per (crop, head, 64-query block).  Everything stays in
// registers except V:
//   * S^T = K Q^T on the matrix cores (K fragment as MFMA A operand, Q fragment as B operand), so a
//     lane's accumulator column is ONE query: the softmax row reductions are 16 in-register
//     values + two cross-lane steps (xor 16, 32) instead of a 64-lane butterfly per row.
//   * the exponentiated S^T accumulators are already in the MFMA B-operand layout for
//     O^T = V^T P^T if the contraction index (key) is enumerated as
//        key(ks, g, j) = 32 ks + 16 (j>>2) + 4 g + (j&3),  g = lane>>4, j = 0..7
//     so P never moves between lanes or through LDS.
//   * V is staged in wave-private LDS (coalesced 16-B rows in) and read back with the gfx950
//     transpose read ds_read_b64_tr_b16 in that same key enumeration (USE_TR), or with plain
//     16-bit gathers (fallback).
//   * keys are processed in chunks of 64 with an online softmax, so the same kernel serves L = 50
//     (one chunk) and L = 197 (four chunks).
// object_attention_kernel: the reference's object-token stream (oadp/oake/objects.py:232-247) has a
// single query per crop; it is a VALU/LDS kernel, one wavefront per (crop, head).
#include <cstdlib>
#include "common.h"
#include "kernels.h"

// -DOAKE_LAB=1 (liboake_hip_lab.so): the production kernels plus the forms that lost their A/B — V fragments by
// 16-bit LDS gathers, 64 queries per wave, the eight-wave blocks, the whole-K/V-in-LDS kernel — selectable by
// attention_variant.  The production library runs variant 31 only.
#ifndef OAKE_LAB
#define OAKE_LAB 0
#endif
#ifndef OAKE_ATTN_SETPRIO
#define OAKE_ATTN_SETPRIO 0
#endif

namespace oake {

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kVStride = 72;                       // halves per V row in LDS (64 + 8 pad = 144 B)
constexpr int kVBytesPerWave = 64 * kVStride * 2;  // 9216

template <typename T, bool USE_TR, int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MT == 2 ? 3 : 1, MT == 2 ? 3 : 2))) void attention_kernel(const T* __restrict__ qkv,
                                                        T* __restrict__ out, int L, int H, int QB,
                                                        int total_waves, int causal) {
  typedef typename T16<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[4 * kVBytesPerWave];

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wg = blockIdx.x * 4 + wid;
  if (wg >= total_waves) return;  // no block-level barriers below: wave-private LDS only
  const int qb = wg % QB;
  const int h = (wg / QB) % H;
  const int img = wg / (QB * H);
  const int C = H * kHeadDim;
  const size_t ld = (size_t)3 * C;
  const T* base = qkv + (size_t)img * L * ld + h * kHeadDim;
  const int fr = lane & 15;
  const int g = lane >> 4;
  const int q0 = qb * (16 * MT);
  T* vs = reinterpret_cast<T*>(smem + wid * kVBytesPerWave);

  // Q fragments (B operand): Q[q0 + 16 mt + fr][32 kk + 8 g .. +8), fetched as full 128-B lines and
  // un-swapped between lanes fr and fr ^ 8 (common.h, swap_piece)
  const int sw_row = fr & 7, sw_col = (fr & 8) * 4 + g * 8;  // piece A: row sw_row, B: row 8 + sw_row
  vec8 qf[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int ra = q0 + mt * 16 + sw_row, rb = ra + 8;
    ra = ra < L ? ra : L - 1;
    rb = rb < L ? rb : L - 1;
    const vec8 pa = *reinterpret_cast<const vec8*>(base + (size_t)ra * ld + sw_col);
    const vec8 pb = *reinterpret_cast<const vec8*>(base + (size_t)rb * ld + sw_col);
    qf[mt][0] = swap_piece(pa, pb, true);
    qf[mt][1] = swap_piece(pb, pa, false);
  }

  float m_run[MT], l_run[MT];
  f32x4 oacc[4][MT];  // [dt][mt] : O^T tile, rows d = 16 dt + 4 g + r, col query = 16 mt + fr
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    m_run[mt] = -1e30f;
    l_run[mt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nchunks = (L + 63) >> 6;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int k0 = kc * 64;

    // ---- issue ALL global loads of this chunk up front (V rows for the LDS stage, K fragments), so
    //      the wave pays one memory latency, not two ----
    uint4 vreg[8];
    {
      const int vr = lane >> 3, vc = lane & 7;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int r = k0 + vr + 8 * i;
        r = r < L ? r : L - 1;  // finite filler for padded keys (their P is exactly 0)
        vreg[i] = *reinterpret_cast<const uint4*>(base + (size_t)r * ld + 2 * C + vc * 8);
      }
    }
    vec8 kfr[MT == 2 ? 4 : 1][2];  // (raw pieces A, B here; un-swapped at the point of use)
    if (MT == 2) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        int ra = k0 + kt * 16 + sw_row, rb = ra + 8;
        ra = ra < L ? ra : L - 1;
        rb = rb < L ? rb : L - 1;
        kfr[kt][0] = *reinterpret_cast<const vec8*>(base + (size_t)ra * ld + C + sw_col);
        kfr[kt][1] = *reinterpret_cast<const vec8*>(base + (size_t)rb * ld + C + sw_col);
      }
    }
    {  // V -> LDS: lane -> (row = lane/8 + 8 i, 16-B chunk = lane%8)
      const int vr = lane >> 3, vc = lane & 7;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<uint4*>(vs + (vr + 8 * i) * kVStride + vc * 8) = vreg[i];
    }

    // ---- S^T[key][query] tiles ----
    f32x4 sacc[4][MT];  // [kt][mt]
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) sacc[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      vec8 pa, pb;
      if (MT == 2) {
        pa = kfr[kt][0];
        pb = kfr[kt][1];
      } else {
        int ra = k0 + kt * 16 + sw_row, rb = ra + 8;
        ra = ra < L ? ra : L - 1;
        rb = rb < L ? rb : L - 1;
        pa = *reinterpret_cast<const vec8*>(base + (size_t)ra * ld + C + sw_col);
        pb = *reinterpret_cast<const vec8*>(base + (size_t)rb * ld + C + sw_col);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const vec8 kf = kk == 0 ? swap_piece(pa, pb, true) : swap_piece(pb, pa, false);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sacc[kt][mt] = T16<T>::mfma(kf, qf[mt][kk], sacc[kt][mt]);
      }
    }

    // ---- online softmax over this chunk's keys; lane owns query 16 mt + fr ----
    vec8 pf[MT][2];  // [mt][ks] : B operand of the PV MFMA
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0 + kt * 16 + 4 * g + r;
          float s = sacc[kt][mt][r];
          // padded keys; causal (text tower, clip build_attention_mask): keys after the query
          s = (key < L && (!causal || key <= q0 + mt * 16 + fr)) ? s : -1e30f;
          sacc[kt][mt][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = rows16_max(mx);
      const float m_new = fmaxf(m_run[mt], mx);
      const float alpha = __expf(m_run[mt] - m_new);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(sacc[kt][mt][r] - m_new);
          sacc[kt][mt][r] = p;
          sum += p;
        }
      sum = rows16_sum(sum);
      l_run[mt] = l_run[mt] * alpha + sum;
      m_run[mt] = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        oacc[dt][mt][0] *= alpha;
        oacc[dt][mt][1] *= alpha;
        oacc[dt][mt][2] *= alpha;
        oacc[dt][mt][3] *= alpha;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        vec8 p8;
#pragma unroll
        for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[2 * ks + (j >> 2)][mt][j & 3]);
        pf[mt][ks] = p8;
      }
    }

    // LDS writes above and reads below are wave-private; the DS queue is in order per wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- O^T[d][query] += V^T[d][key] P^T[key][query] ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        vec8 vf;
        if (USE_TR) {
          // ds_read_b64_tr_b16: within a 16-lane group, lane i supplies the address of the 8-B
          // slice {row i/4, cols 4(i%4)..+3} of a 4x16 block and receives column i of it.
          const int sub = fr >> 2, c4 = (fr & 3) * 4;
          const T* p0 = vs + (32 * ks + 4 * g + sub) * kVStride + dt * 16 + c4;
          const T* p1 = p0 + 16 * kVStride;
          typedef s16x4 __attribute__((address_space(3))) * lds4_t;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p1));
          s16x8 both;
          both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
          both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
          vf = __builtin_bit_cast(vec8, both);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            vf[j] = vs[(32 * ks + 16 * (j >> 2) + 4 * g + (j & 3)) * kVStride + dt * 16 + fr];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) oacc[dt][mt] = T16<T>::mfma(vf, pf[mt][ks], oacc[dt][mt]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- normalise and store: lane holds O[query = 16 mt + fr][d = 16 dt + 4 g + 0..3], i.e. 8-B pieces
  //      of 16 different rows: stored like that, an instruction touches 16 quarter lines (measured:
  //      7.7 B/cycle/CU, a third of this kernel's time).  The tile goes through the wave's V staging
  //      area instead (free after the last PV) and leaves as 8 rows x 128 B per instruction. ----
  T* obase = out + (size_t)img * L * C + h * kHeadDim;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float inv = 1.0f / l_run[mt];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 o = oacc[dt][mt];
      *reinterpret_cast<uint2*>(vs + (mt * 16 + fr) * kVStride + dt * 16 + 4 * g) =
          pack4<T>(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 2 * MT; ++i) {
    const int row = (lane >> 3) + 8 * i;
    const uint4 v = *reinterpret_cast<const uint4*>(vs + row * kVStride + (lane & 7) * 8);
    if (q0 + row < L)
      *reinterpret_cast<uint4*>(obase + (size_t)(q0 + row) * C + (lane & 7) * 8) = v;
  }
}

// ---- attention for short sequences (L <= 64, no causal mask: encode_image and blocks mode) ----
// attention_kernel above is latency-bound at L = 50: a wave stages its own K and V through 64 of its
// 160-175 registers, K and V are fetched once per 32-query wave, and two or three such waves fit a SIMD.
// Here the two waves that share a (crop, head) item share its K and V in LDS:
//   * a block is 4 waves = 2 items.  Of an item's pair of waves one moves the K rows, the other the V
//     rows, with LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B per instruction, no registers); each
//     fetches its own 32 Q rows straight into fragment registers (full lines + swap_piece).  One block
//     barrier.
//   * LDS rows are 128 B with the 16-B chunks XOR-swizzled by (row >> 1) & 7 on the source side (as in
//     gemm.hip), so the K fragment reads (ds_read_b128) and the V transpose reads
//     (ds_read_b64_tr_b16) are conflict-free without padding.
//   * a region holds exactly L rows: [V0 V1 K0 K1 pad].  The MFMA tiles read 64 rows; rows past L of a
//     V region are the first rows of the next region — finite 16-bit data, multiplied by P = 0
//     exactly —, rows past L of a K region only produce scores that the key mask discards.  At L = 50
//     that is 27 KB per block and 72 registers: five blocks = 20 waves per CU.  (Measured at batch 256:
//     2 items per block 14.4 us; 3 items, i.e. the whole batch resident at once, 17.6 — every block
//     then loads, computes and stores in step and nothing overlaps; 1 item 17.3; attention_kernel 18.0.)
//   * O leaves through the wave's half of the K rows (after a second barrier): 8 rows x 128 B per store.
constexpr int kPairItems = 2;

__host__ __device__ constexpr int pair_lds_bytes(int L) { return (2 * kPairItems * L + (64 - L)) * 128; }

template <typename T>
__global__ __launch_bounds__(2 * kPairItems * 64) __attribute__((amdgpu_waves_per_eu(5, 5)))
void attention_pair_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int H, int items) {
  typedef typename T16<T>::vec8 vec8;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wid >> 1, qb = wid & 1, q0 = qb * 32;
  const int item_raw = blockIdx.x * kPairItems + pair;
  const bool valid = item_raw < items;  // (a ragged item count leaves the last pairs idle: they replay
  const int item = valid ? item_raw : items - 1;  // the last item so that every wave takes both barriers)
  const int img = item / H, h = item - img * H;
  const int C = H * kHeadDim;
  const size_t ld = (size_t)3 * C;
  const T* base = qkv + (size_t)img * L * ld + h * kHeadDim;
  const int region = L * 128;
  char* vs = smem + pair * region;
  char* ks = smem + (kPairItems + pair) * region;
  const int fr = lane & 15, g = lane >> 4;
  const int fsw = (fr >> 1) & 7;

  // wave 0 of the pair: the item's K rows -> LDS, wave 1: its V rows (lanes past row L - 1 stay off)
  {
    char* dst = qb == 0 ? ks : vs;
    const T* src0 = base + (qb == 0 ? C : 2 * C);
    for (int jr = 0; jr * 8 < L; ++jr) {
      const int row = jr * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      if (row < L)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src0 + (size_t)row * ld + chunk * 8),
                                         (lds_ptr_t)(dst + jr * 1024), 16, 0, OAKE_STREAM_AUX);
    }
  }
  // Q fragments (B operand): Q[q0 + 16 mt + fr][32 kk + 8 g .. +8), as full lines (common.h, swap_piece)
  vec8 qf[2][2];
  {
    const int sw_row = fr & 7, sw_col = (fr & 8) * 4 + g * 8;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      int ra = q0 + mt * 16 + sw_row, rb = ra + 8;
      ra = ra < L ? ra : L - 1;
      rb = rb < L ? rb : L - 1;
      const vec8 pa = stream_load16(reinterpret_cast<const vec8*>(base + (size_t)ra * ld + sw_col));
      const vec8 pb = stream_load16(reinterpret_cast<const vec8*>(base + (size_t)rb * ld + sw_col));
      qf[mt][0] = swap_piece(pa, pb, true);
      qf[mt][1] = swap_piece(pb, pa, false);
    }
  }
  // the pad rows behind the last region: a V tile of a short sequence (4 L < L + 64) reaches them, with
  // P = 0 exactly — whatever an earlier kernel left there must not be a NaN
  for (int i = threadIdx.x; i < (64 - L) * 8; i += 2 * kPairItems * 64)
    *reinterpret_cast<uint4*>(smem + 2 * kPairItems * region + i * 16) = make_uint4(0u, 0u, 0u, 0u);
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0), lgkmcnt(0): this wave's LDS-DMA pieces and zero rows are in
  __syncthreads();

  // S^T[key][query] = K Q^T: K fragment as A operand, Q fragment as B operand
  f32x4 sacc[4][2];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) sacc[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const vec8 kf = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * 128 + (((kk * 4 + g) ^ fsw) << 4));
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) sacc[kt][mt] = T16<T>::mfma(kf, qf[mt][kk], sacc[kt][mt]);
    }

  // softmax over the keys of a query: 16 in-register values + the four 16-lane rows
  vec8 pf[2][2];
  float inv[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = sacc[kt][mt][i];
        s = kt * 16 + 4 * g + i < L ? s : -1e30f;  // padded keys
        sacc[kt][mt][i] = s;
        mx = fmaxf(mx, s);
      }
    mx = rows16_max(mx);
    const float nb = -mx * kLog2e;  // exp(s - mx) = 2^(s log2(e) - mx log2(e)): one FMA + v_exp_f32
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][mt][i], kLog2e, nb));
        sacc[kt][mt][i] = p;
        sum += p;
      }
    inv[mt] = 1.0f / rows16_sum(sum);
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      vec8 p8;
#pragma unroll
      for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[2 * ks2 + (j >> 2)][mt][j & 3]);
      pf[mt][ks2] = p8;
    }
  }

  // O^T[d][query] = V^T P^T, V through the transpose read (key enumeration as in attention_kernel)
  f32x4 oacc[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) oacc[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2) {
    const int row0 = 32 * ks2 + 4 * g + (fr >> 2);  // and row0 + 16: same swizzle
    const int vsw = (row0 >> 1) & 7;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int c4 = (fr & 3) * 4;  // halves within the 16-column block dt
      const char* p0 = vs + row0 * 128 + (((dt * 2 + (c4 >> 3)) ^ vsw) << 4) + (c4 & 4) * 2;
      typedef s16x4 __attribute__((address_space(3))) * lds4_t;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0 + 16 * 128));
      s16x8 both;
      both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
      both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
      const vec8 vf = __builtin_bit_cast(vec8, both);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) oacc[dt][mt] = T16<T>::mfma(vf, pf[mt][ks2], oacc[dt][mt]);
    }
  }

  // O -> this wave's share of the item's K rows (every wave of the block is past S and PV) -> full
  // lines out.  Queries past L are not staged: their rows belong to the next region.
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 o = oacc[dt][mt];
      if (q0 + mt * 16 + fr < L)
        *reinterpret_cast<uint2*>(ks + (q0 + mt * 16 + fr) * 128 + (((dt * 2 + (g >> 1)) ^ fsw) << 4) + (g & 1) * 8) =
            pack4<T>(o[0] * inv[mt], o[1] * inv[mt], o[2] * inv[mt], o[3] * inv[mt]);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  T* obase = out + (size_t)img * L * C + h * kHeadDim;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = q0 + (lane >> 3) + 8 * i;
    const uint4 v = *reinterpret_cast<const uint4*>(ks + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    if (valid && row < L) aux_store16(obase + (size_t)row * C + (lane & 7) * 8, v);
  }
}

// ---- object-token attention -------------------------------------------------------------------
// One wavefront per (crop n, head h).  query = qkv_y[n].q_h ; keys/values: x rows 1..L-1 of crop n
// (bias -100*mask[n,p]) then the object token's own k/v from qkv_y[n] (bias 0).
constexpr int kObjMaxKeys = 1024;

template <typename T, typename TM>
__global__ __launch_bounds__(256) void object_attention_kernel(const T* __restrict__ qkv_x,
                                                               const T* __restrict__ qkv_y,
                                                               const TM* __restrict__ mask,
                                                               T* __restrict__ out, int L, int H,
                                                               int total_waves) {
  __shared__ float ps[4][kObjMaxKeys];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int wg = blockIdx.x * 4 + wid;
  if (wg >= total_waves) return;
  const int h = wg % H;
  const int n = wg / H;
  const int C = H * kHeadDim;
  const size_t ld = (size_t)3 * C;
  const int nk = L;  // L-1 patch keys + the object token
  const T* xb = qkv_x + (size_t)n * L * ld + h * kHeadDim;
  const T* yb = qkv_y + (size_t)n * ld + h * kHeadDim;
  const TM* mb = mask + (size_t)n * (L - 1);

  // q in registers (every lane holds the whole 64-vector; loads broadcast)
  float q[kHeadDim];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    typedef typename T16<T>::vec8 vec8;
    const vec8 v = *reinterpret_cast<const vec8*>(yb + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[i * 8 + j] = to32<T>(v[j]);
  }

  // scores: lane handles keys lane, lane + 64, ...
  float mx = -1e30f;
  for (int k = lane; k < nk; k += 64) {
    const T* kr = (k < L - 1) ? (xb + (size_t)(1 + k) * ld + C) : (yb + C);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      typedef typename T16<T>::vec8 vec8;
      const vec8 v = *reinterpret_cast<const vec8*>(kr + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += q[i * 8 + j] * to32<T>(v[j]);
    }
    if (k < L - 1) s += -100.0f * (float)mb[k];
    ps[wid][k] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < nk; k += 64) {
    const float p = __expf(ps[wid][k] - mx);
    ps[wid][k] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // out[d = lane] = sum_k p[k] V[k][d]
  // (eight V rows in flight per wave, four accumulators in key order k mod 4: the loop is load-latency-bound)
  float acc4[4] = {0.f, 0.f, 0.f, 0.f};
  const T* vb = xb + ld + 2 * C + lane;  // V row of patch key 0
  int k = 0;
  for (; k + 8 <= L - 1; k += 8) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = vb[(size_t)(k + j) * ld];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc4[j & 3] += ps[wid][k + j] * to32<T>(v[j]);
  }
  for (; k < L - 1; ++k) acc4[k & 3] += ps[wid][k] * to32<T>(vb[(size_t)k * ld]);
  float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  acc += ps[wid][L - 1] * to32<T>(yb[2 * C + lane]);
  out[(size_t)n * C + h * kHeadDim + lane] = to16<T>(acc * inv);
}

// attention_coop_kernel: for L > 64 (objects mode L = 197, text L = 77).  With one wave per 32-query
// block every wave streams ALL keys and values of its (crop, head): at L = 197 that is 7 waves
// re-reading the same K / V from L2 (2.2 GB per launch, 10 TB/s — the kernel's bound).  Here the four
// waves of a block own 4 x 32 consecutive queries of ONE (crop, head) and share each 64-key chunk of K
// and V through LDS: loaded once per block (2 blocks per head at L = 197), register-prefetched one
// chunk ahead so the global latency hides under the previous chunk's MFMAs.  Per-wave arithmetic is
// attention_kernel<T, true, 2>'s: S^T = K Q^T, in-register online softmax, P as the PV B operand, V
// through ds_read_b64_tr_b16.
// Objects mode: the reference Hooks' object token (one query per crop: keys = patch rows 1..L-1 with
// bias -100 * mask, plus the token's own k / v; oadp/oake/objects.py:232-247) rides on the block's
// first idle wave (L = 197: the last block has 69 queries on three waves) — the K / V chunks it needs
// are already in LDS.  It runs the same S / softmax / PV machinery with its query in every column;
// key 0 (the CLS row, not a key of the object token) is re-scored with the token's own key, and the
// V row it dragged in is swapped for the token's own v at the end.
struct ObjArgs {
  const void* qkv_y;   // [n, 3C] q/k/v of the object tokens (nullptr: no object token)
  const void* mask;    // [n, L-1]
  void* out_y;         // [n, C]
  int mask_f16;
};

// NW = 8 (L > 128; attention_variant bit 32, OFF by default): ONE block of eight waves per (crop, head)
// instead of two blocks of four — K and V are read once per head instead of twice (objects mode, L = 197:
// 1.41x -> 1.0x the algorithmic bytes).  Measured SLOWER, 16.3 vs 12.4 ms per objects step: at 136 registers
// one eight-wave block fits a CU (8 waves, all meeting at every chunk barrier) where three four-wave blocks
// do (12 waves, independent); at 128 registers it spills.  The kernel is latency-, not HBM-bound.
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void attention_coop_kernel(const T* __restrict__ qkv,
                                                             T* __restrict__ out, int L, int H, int QG,
                                                             int causal, ObjArgs obj, int n_items) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int MT = 2;
  static_assert(NW == 4 || NW == 8, "four or eight waves of 32 queries");
  __shared__ __attribute__((aligned(16))) T ks[64 * kVStride];
  __shared__ __attribute__((aligned(16))) T vs[64 * kVStride];
  __shared__ __attribute__((aligned(16))) T ostage[NW > 4 ? 128 * kVStride : 8];  // output staging of waves 4..7
  __shared__ float mbias[256];  // objects mode: -100 * mask of the crop's patch keys, indexed by key

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the object-token branches are scalar)
  // Block -> (crop, head, query group).  Blocks go to XCD (blockIdx.x % 8), each with its own L2: the QG
  // blocks of one (crop, head) — which read the same K / V — take consecutive slots of ONE XCD, so the second
  // read hits that L2 instead of HBM (with the plain order they sat on different XCDs: 1.41x the algorithmic
  // bytes at L = 197).  The host launches ceil(items / 8) * 8 * QG blocks.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int qg = slot % QG;
  const int item = (slot / QG) * 8 + xcd;
  if (item >= n_items) return;
  const int h = item % H;
  const int img = item / H;
  const int C = H * kHeadDim;
  const size_t ld = (size_t)3 * C;
  const T* base = qkv + (size_t)img * L * ld + h * kHeadDim;
  const int fr = lane & 15;
  const int g = lane >> 4;
  const int q0 = qg * (32 * NW) + wid * 32;
  // the object token takes the first wave of the last block that has no queries of its own
  const bool is_obj = obj.qkv_y != nullptr && qg == QG - 1 && q0 >= L && q0 - 32 < L;
  const bool active = q0 < L || is_obj;  // (inactive waves still load and synchronise)
  const T* yb = reinterpret_cast<const T*>(obj.qkv_y) + (size_t)img * ld + h * kHeadDim;

  vec8 qf[MT][2];
  float s_self = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int r = q0 + mt * 16 + fr;
    r = r < L ? r : L - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qf[mt][kk] = is_obj ? *reinterpret_cast<const vec8*>(yb + kk * 32 + g * 8)
                          : *reinterpret_cast<const vec8*>(base + (size_t)r * ld + kk * 32 + g * 8);
  }
  if (obj.qkv_y != nullptr && qg == QG - 1 && tid < 256) {
    // (visible after the first chunk's barrier; per-element global loads in the softmax loop made the
    // object token's wave the slowest of its block)
    float mval = 0.f;
    if (tid >= 1 && tid < L)
      mval = obj.mask_f16 ? (float)reinterpret_cast<const f16_t*>(obj.mask)[(size_t)img * (L - 1) + tid - 1]
                          : reinterpret_cast<const float*>(obj.mask)[(size_t)img * (L - 1) + tid - 1];
    mbias[tid] = -100.0f * mval;
  }
  if (is_obj) {  // q_y . k_y: the four lane groups hold d = 32 kk + 8 g + j between them
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const vec8 kv = *reinterpret_cast<const vec8*>(yb + C + kk * 32 + g * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s_self += to32<T>(qf[0][kk][j]) * to32<T>(kv[j]);
    }
    s_self = rows16_sum(s_self);
  }
  float m_run[MT], l_run[MT];
  f32x4 oacc[4][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    m_run[mt] = -1e30f;
    l_run[mt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // cooperative chunk load: thread -> rows (tid>>3) and (tid>>3)+32, 16-B chunk tid&7, of K and of V
  // (eight waves: one row per thread)
  const int lr = tid >> 3, lc = tid & 7;
  // (macros, not lambdas: register arrays captured by a lambda end up in scratch)
  uint4 kreg0, kreg1, vreg0, vreg1;
#define OAKE_FETCH(k0_)                                                                      \
  do {                                                                                       \
    int _r0 = (k0_) + lr, _r1 = (k0_) + lr + 32;                                             \
    _r0 = _r0 < L ? _r0 : L - 1; /* finite filler for padded keys (their P is exactly 0) */  \
    _r1 = _r1 < L ? _r1 : L - 1;                                                             \
    kreg0 = *reinterpret_cast<const uint4*>(base + (size_t)_r0 * ld + C + lc * 8);           \
    vreg0 = *reinterpret_cast<const uint4*>(base + (size_t)_r0 * ld + 2 * C + lc * 8);       \
    if (NW == 4) {                                                                           \
      kreg1 = *reinterpret_cast<const uint4*>(base + (size_t)_r1 * ld + C + lc * 8);         \
      vreg1 = *reinterpret_cast<const uint4*>(base + (size_t)_r1 * ld + 2 * C + lc * 8);     \
    }                                                                                        \
  } while (0)
#define OAKE_PUBLISH()                                                                       \
  do {                                                                                       \
    *reinterpret_cast<uint4*>(ks + lr * kVStride + lc * 8) = kreg0;                          \
    *reinterpret_cast<uint4*>(vs + lr * kVStride + lc * 8) = vreg0;                          \
    if (NW == 4) {                                                                           \
      *reinterpret_cast<uint4*>(ks + (lr + 32) * kVStride + lc * 8) = kreg1;                 \
      *reinterpret_cast<uint4*>(vs + (lr + 32) * kVStride + lc * 8) = vreg1;                 \
    }                                                                                        \
  } while (0)

  // Work that only pads is skipped in whole 16-row tiles (wave-uniform branches): a wave with n valid
  // queries runs ceil(n / 16) of its two query tiles (L = 197: the last block's third wave has 5 queries,
  // the object token's wave needs one tile), and a chunk with n valid keys ceil(n / 16) of its four key
  // tiles and ceil(n / 32) of its two PV halves (L = 197: the fourth chunk holds 5 keys; L = 77: 13).
  const int nmt = is_obj ? 1 : (q0 >= L ? 0 : (L - q0 > 16 ? 2 : 1));
  const int nchunks = (L + 63) >> 6;
  OAKE_FETCH(0);
  for (int kc = 0; kc < nchunks; ++kc) {
    const int k0 = kc * 64;
    OAKE_PUBLISH();
    __syncthreads();
    if (kc + 1 < nchunks) OAKE_FETCH(k0 + 64);  // in flight under this chunk's arithmetic
    // a chunk entirely after this wave's last query is masked out completely under the causal mask
    const bool skip = !active || (causal && !is_obj && k0 > q0 + 31);
    const int nkt = L - k0 >= 64 ? 4 : (L - k0 + 15) >> 4;  // key tiles of this chunk with a valid key
    const int nks = (nkt + 1) >> 1;
    // The common case — a full 64-key chunk, two full query tiles, no mask of any kind (objects mode: 18 of
    // the 24 wave-chunks of a head) — as straight-line code: no per-score validity tests, no guards around
    // the tiles, S accumulators started from the MFMA's zero operand.  The general path below measured ~25
    // VALU instructions per score, 3x the matrix work; this one ~5.  Same operations in the same order.
    const bool fast = active && !is_obj && !causal && nmt == 2 && L - k0 >= 64;  // (nkt == 4 also holds for 49..63 keys)
    if (fast) {
      const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 sacc[4][MT];
#if OAKE_ATTN_SETPRIO  // (measurement switch: issue priority for the MFMA clusters; objects 83.8 vs 83.6 images/s and
                       // 12 bytes of scratch at the kernel's 160 registers, profiles/r04/ab_session_setprio_objects.log: off)
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const vec8 kf0 = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * kVStride + g * 8);
        const vec8 kf1 = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * kVStride + 32 + g * 8);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          sacc[kt][mt] = T16<T>::mfma(kf0, qf[mt][0], zero4);
          sacc[kt][mt] = T16<T>::mfma(kf1, qf[mt][1], sacc[kt][mt]);
        }
      }
#if OAKE_ATTN_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      vec8 pf[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kt][mt][r]);
        mx = rows16_max(mx);
        const float m_new = fmaxf(m_run[mt], mx);
        const float alpha = __expf(m_run[mt] - m_new);
        const float nb = -m_new * kLog2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kt][mt][r], kLog2e, nb));
            sacc[kt][mt][r] = pv;
            sum += pv;
          }
        sum = rows16_sum(sum);
        l_run[mt] = l_run[mt] * alpha + sum;
        m_run[mt] = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          oacc[dt][mt][0] *= alpha;
          oacc[dt][mt][1] *= alpha;
          oacc[dt][mt][2] *= alpha;
          oacc[dt][mt][3] *= alpha;
        }
#pragma unroll
        for (int ksx = 0; ksx < 2; ++ksx) {
          vec8 p8;
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[2 * ksx + (j >> 2)][mt][j & 3]);
          pf[mt][ksx] = p8;
        }
      }
#if OAKE_ATTN_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int ksx = 0; ksx < 2; ++ksx) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int sub = fr >> 2, c4 = (fr & 3) * 4;
          const T* p0 = vs + (32 * ksx + 4 * g + sub) * kVStride + dt * 16 + c4;
          const T* p1 = p0 + 16 * kVStride;
          typedef s16x4 __attribute__((address_space(3))) * lds4_t;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p1));
          s16x8 both;
          both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
          both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
          const vec8 vf = __builtin_bit_cast(vec8, both);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) oacc[dt][mt] = T16<T>::mfma(vf, pf[mt][ksx], oacc[dt][mt]);
        }
      }
#if OAKE_ATTN_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    } else if (!skip) {
      f32x4 sacc[4][MT];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sacc[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const vec8 kf = *reinterpret_cast<const vec8*>(ks + (kt * 16 + fr) * kVStride + kk * 32 + g * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              if (mt < nmt) sacc[kt][mt] = T16<T>::mfma(kf, qf[mt][kk], sacc[kt][mt]);
          }
        }
      }
      vec8 pf[MT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (mt >= nmt) continue;
        float mx = -1e30f;
        if (is_obj) {  // (one scalar branch around the whole tile, not one per score)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nkt) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = k0 + kt * 16 + 4 * g + r;
              float sv = sacc[kt][mt][r];
              // the CLS row (key 0) is not a key of the object token: its own key instead
              sv = key == 0 ? s_self : (key < L ? sv + mbias[key & 255] : -1e30f);
              sacc[kt][mt][r] = sv;
              mx = fmaxf(mx, sv);
            }
          }
        } else {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nkt) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = k0 + kt * 16 + 4 * g + r;
              float sv = sacc[kt][mt][r];
              // (skipping the masking for entirely valid 16-key tiles behind a uniform branch measured slower)
              sv = (key < L && (!causal || key <= q0 + mt * 16 + fr)) ? sv : -1e30f;
              sacc[kt][mt][r] = sv;
              mx = fmaxf(mx, sv);
            }
          }
        }
        mx = rows16_max(mx);
        const float m_new = fmaxf(m_run[mt], mx);
        const float alpha = __expf(m_run[mt] - m_new);
        const float nb = -m_new * kLog2e;  // exp(s - m) = 2^(s log2(e) - m log2(e)): one FMA + v_exp_f32
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][mt][r], kLog2e, nb));
              sacc[kt][mt][r] = p;
              sum += p;
            }
          } else {
            sacc[kt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};  // P of a key tile without valid keys is exactly 0
          }
        }
        sum = rows16_sum(sum);
        l_run[mt] = l_run[mt] * alpha + sum;
        m_run[mt] = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          oacc[dt][mt][0] *= alpha;
          oacc[dt][mt][1] *= alpha;
          oacc[dt][mt][2] *= alpha;
          oacc[dt][mt][3] *= alpha;
        }
#pragma unroll
        for (int ksx = 0; ksx < 2; ++ksx) {
          vec8 p8;
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = to16<T>(sacc[2 * ksx + (j >> 2)][mt][j & 3]);
          pf[mt][ksx] = p8;
        }
      }
#pragma unroll
      for (int ksx = 0; ksx < 2; ++ksx) {
        if (ksx >= nks) continue;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int sub = fr >> 2, c4 = (fr & 3) * 4;
          const T* p0 = vs + (32 * ksx + 4 * g + sub) * kVStride + dt * 16 + c4;
          const T* p1 = p0 + 16 * kVStride;
          typedef s16x4 __attribute__((address_space(3))) * lds4_t;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(p1));
          s16x8 both;
          both[0] = lo[0]; both[1] = lo[1]; both[2] = lo[2]; both[3] = lo[3];
          both[4] = hi[0]; both[5] = hi[1]; both[6] = hi[2]; both[7] = hi[3];
          const vec8 vf = __builtin_bit_cast(vec8, both);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (mt < nmt) oacc[dt][mt] = T16<T>::mfma(vf, pf[mt][ksx], oacc[dt][mt]);
        }
      }
    }
    __syncthreads();  // every wave is done with this chunk's K / V before the next publish
  }

#undef OAKE_FETCH
#undef OAKE_PUBLISH
  if (!active) return;
  if (is_obj) {
    // every query column holds the object token; lanes fr == 0 write it.  The PV product used V row 0
    // (the CLS row) with the weight of the token's own key: swap in the token's own v.
    const float inv = 1.0f / l_run[0];
    const float p0 = __expf(s_self - m_run[0]) * inv;
    T* oy = reinterpret_cast<T*>(obj.out_y) + (size_t)img * C + h * kHeadDim;
    if (fr == 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + 4 * g;
        typedef typename T16<T>::vec4 vec4;
        const vec4 vy = __builtin_bit_cast(vec4, *reinterpret_cast<const uint2*>(yb + 2 * C + d));
        const vec4 vc = __builtin_bit_cast(vec4, *reinterpret_cast<const uint2*>(base + 2 * C + d));
        const f32x4 o = oacc[dt][0];
        *reinterpret_cast<uint2*>(oy + d) =
            pack4<T>(o[0] * inv + p0 * (to32<T>(vy[0]) - to32<T>(vc[0])),
                     o[1] * inv + p0 * (to32<T>(vy[1]) - to32<T>(vc[1])),
                     o[2] * inv + p0 * (to32<T>(vy[2]) - to32<T>(vc[2])),
                     o[3] * inv + p0 * (to32<T>(vy[3]) - to32<T>(vc[3])));
      }
    }
    return;
  }
  // O through LDS (the K / V chunk buffers are free after the loop's last barrier; wave w takes 32 rows
  // of ks or vs) so that it leaves as 8 rows x 128 B per store instead of 16 quarter lines
  T* obase = out + (size_t)img * L * C + h * kHeadDim;
  T* stage = wid >= 4 ? ostage + (wid - 4) * 32 * kVStride : ((wid & 2) ? vs : ks) + (wid & 1) * 32 * kVStride;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (mt >= nmt) continue;
    const float inv = 1.0f / l_run[mt];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 o = oacc[dt][mt];
      *reinterpret_cast<uint2*>(stage + (mt * 16 + fr) * kVStride + dt * 16 + 4 * g) =
          pack4<T>(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 2 * MT; ++i) {
    const int row = (lane >> 3) + 8 * i;
    const uint4 v = *reinterpret_cast<const uint4*>(stage + row * kVStride + (lane & 7) * 8);
    if (q0 + row < L) aux_store16(obase + (size_t)(q0 + row) * C + (lane & 7) * 8, v);
  }
}

#if OAKE_LAB
#include "attention_lab_full.inc"
#endif
#include "attention_head.inc"


__global__ void tr_read_probe_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  typedef s16x4 __attribute__((address_space(3))) * lds4_t;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

}  // namespace

// LaunchOpts::attention_variant bits (default 31: bits 32 and 64 are measured experiments, off):
//   1  ds_read_b64_tr_b16 V fragments (else 16-bit LDS gathers)
//   2  32 queries per wave (else 64: twice the registers, half the waves)
//   4  sequences longer than one key chunk share K / V through LDS (attention_coop_kernel)
//   8  objects mode: the object token's attention rides on an idle wave of that kernel
//  16  L <= 64 without a causal mask: two waves share an item's K / V in LDS (attention_pair_kernel)
//  32  L > 128: one block of eight waves per (crop, head) in attention_coop_kernel (else two of four) — slower, off
//  64  64 < L <= 208: the whole K / V of a head in LDS by LDS-DMA, no barrier in the key loop (attention_full_kernel)
//      — the same speed as the cooperative kernel, off
// 128  192 < L <= 208 without a causal mask (objects mode, L = 197): one block per (crop, head), the whole S^T of a
//      32-query unit in registers, one-pass softmax (attention_head_kernel, attention_head.inc)
namespace {
struct AttnBits {
  bool use_tr, q32, coop, fuse_obj, pair, coop8, full, head;
  explicit AttnBits(const LaunchOpts* o) {
    // (production: 159, or 31 = the cooperative kernel at L = 197 for A/B runs; oake_set_option rejects the others)
    const int v = o ? o->attention_variant : kAttentionVariantDefault;
    use_tr = v & 1; q32 = v & 2; coop = v & 4; fuse_obj = v & 8; pair = v & 16; coop8 = v & 32; full = v & 64;
    head = v & 128;
  }
};
}  // namespace

template <typename T, int MT>
static void attn_launch_t(const void* qkv, void* out, int n, int L, int heads, int causal,
                          hipStream_t s, bool use_tr) {
  const int QB = (L + 16 * MT - 1) / (16 * MT);
  const int tw = n * heads * QB;
  const dim3 g((tw + 3) / 4), b(256);
  const T* in = reinterpret_cast<const T*>(qkv);
  T* o = reinterpret_cast<T*>(out);
#if OAKE_LAB
  if (!use_tr)
    OAKE_LAUNCH((attention_kernel<T, false, MT>), g, b, 0, s, in, o, L, heads, QB, tw, causal);
  else
#endif
    OAKE_LAUNCH((attention_kernel<T, true, MT>), g, b, 0, s, in, o, L, heads, QB, tw, causal);
}

template <typename T>
static hipError_t attn_pair_launch_t(const void* qkv, void* out, int n, int L, int heads, hipStream_t s) {
  static DynLdsAttr attr;
  auto kern = attention_pair_kernel<T>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), pair_lds_bytes(64)); e != hipSuccess) return e;
  const int items = n * heads;
  OAKE_LAUNCH(kern, dim3((items + kPairItems - 1) / kPairItems), dim3(2 * kPairItems * 64),
                     pair_lds_bytes(L), s, reinterpret_cast<const T*>(qkv), reinterpret_cast<T*>(out), L,
                     heads, items);
  return hipGetLastError();
}

// attention_head_kernel; obj_only: the object token alone (the last layer of objects mode)
static hipError_t launch_attention_head(int dtype16, const void* qkv, void* out, int n, int L, int heads, hipStream_t s,
                                        const void* qkv_y, const void* mask, int mask_dtype, void* out_y,
                                        int obj_only) {
  const int n_items = n * heads;
  const ObjArgs obj{qkv_y, mask, out_y, mask_dtype == DT_F16 ? 1 : 0};
  if (dtype16 == DT_F16)
    OAKE_LAUNCH((attention_head_kernel<f16_t, kHeadNKT>), dim3(n_items), dim3(256), 0, s,
                reinterpret_cast<const f16_t*>(qkv), reinterpret_cast<f16_t*>(out), L, heads, obj, n_items, obj_only);
  else if (dtype16 == DT_BF16)
    OAKE_LAUNCH((attention_head_kernel<bf16_t, kHeadNKT>), dim3(n_items), dim3(256), 0, s,
                reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(out), L, heads, obj, n_items, obj_only);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

bool attention_variant_supported(int v) {
#if OAKE_LAB
  return v >= 0 && v <= 255;
#else
  return v == 31 || v == kAttentionVariantDefault;
#endif
}

bool attention_fuses_object_token(int L, const LaunchOpts* opts) {
  // needs the cooperative kernel and a wave without queries in the last block of each head
  const AttnBits b(opts);
  if (b.head && b.use_tr && b.fuse_obj && head_supported(L, kHeadNKT, true)) return true;  // a query tile of its own
#if OAKE_LAB
  const bool full = b.full && L <= kFullMaxL;                  // attention_full_kernel: four waves per block
#else
  const bool full = false;
#endif
  const int per = !full && b.coop8 && L > 128 ? 256 : 128;     // queries per block
  const int tail = L - per * ((L + per - 1) / per - 1);
  return b.coop && b.use_tr && b.fuse_obj && L > 64 && L <= 256 && tail <= per - 32;
}

hipError_t launch_attention(int dtype16, const void* qkv, void* out, int n, int L, int heads,
                            int causal, hipStream_t s, const void* qkv_y, const void* mask,
                            int mask_dtype, void* out_y, const LaunchOpts* opts) {
  if (n <= 0) return hipSuccess;
  const AttnBits bits(opts);
  if (qkv_y != nullptr && !attention_fuses_object_token(L, opts)) return hipErrorInvalidValue;
  if (qkv_y != nullptr && mask_dtype != DT_F32 && mask_dtype != DT_F16) return hipErrorInvalidValue;
  if (L <= 0 || heads <= 0) return hipErrorInvalidValue;
  if ((long)n * heads * ((L + 31) / 32) > 0x7fffffffL) return hipErrorInvalidValue;
  if (bits.head && bits.use_tr && !causal && head_supported(L, kHeadNKT, qkv_y != nullptr))
    return launch_attention_head(dtype16, qkv, out, n, L, heads, s, qkv_y, mask, mask_dtype, out_y, 0);
#if OAKE_LAB
  if (bits.coop && bits.use_tr && bits.full && L > 64 && L <= kFullMaxL) {
    const int QG = (L + 127) / 128;
    const int n_items = n * heads;
    const dim3 grid(((n_items + 7) / 8) * 8 * QG), blk(256);
    const ObjArgs obj{qkv_y, mask, out_y, mask_dtype == DT_F16 ? 1 : 0};
    const int lds = full_lds_bytes(L, qkv_y != nullptr);
    static DynLdsAttr attr16, attrbf;
    if (dtype16 == DT_F16) {
      auto kern = attention_full_kernel<f16_t>;
      if (hipError_t e = attr16.ensure(reinterpret_cast<const void*>(kern), full_lds_bytes(kFullMaxL, true) + 1024);
          e != hipSuccess)
        return e;
      OAKE_LAUNCH(kern, grid, blk, lds, s, reinterpret_cast<const f16_t*>(qkv), reinterpret_cast<f16_t*>(out), L,
                  heads, QG, causal, obj, n_items);
    } else if (dtype16 == DT_BF16) {
      auto kern = attention_full_kernel<bf16_t>;
      if (hipError_t e = attrbf.ensure(reinterpret_cast<const void*>(kern), full_lds_bytes(kFullMaxL, true) + 1024);
          e != hipSuccess)
        return e;
      OAKE_LAUNCH(kern, grid, blk, lds, s, reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(out), L,
                  heads, QG, causal, obj, n_items);
    } else {
      return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
#endif
  if (bits.coop && bits.use_tr && L > 64) {
    const bool eight = bits.coop8 && L > 128;
    const int per = eight ? 256 : 128;
    const int QG = (L + per - 1) / per;
    const int n_items = n * heads;
    const dim3 grid(((n_items + 7) / 8) * 8 * QG), blk(eight ? 512 : 256);
    const ObjArgs obj{qkv_y, mask, out_y, mask_dtype == DT_F16 ? 1 : 0};
    if (dtype16 == DT_F16) {
#if OAKE_LAB
      if (eight)
        OAKE_LAUNCH((attention_coop_kernel<f16_t, 8>), grid, blk, 0, s, reinterpret_cast<const f16_t*>(qkv),
                    reinterpret_cast<f16_t*>(out), L, heads, QG, causal, obj, n_items);
      else
#endif
        OAKE_LAUNCH((attention_coop_kernel<f16_t, 4>), grid, blk, 0, s, reinterpret_cast<const f16_t*>(qkv),
                    reinterpret_cast<f16_t*>(out), L, heads, QG, causal, obj, n_items);
    } else if (dtype16 == DT_BF16) {
#if OAKE_LAB
      if (eight)
        OAKE_LAUNCH((attention_coop_kernel<bf16_t, 8>), grid, blk, 0, s, reinterpret_cast<const bf16_t*>(qkv),
                    reinterpret_cast<bf16_t*>(out), L, heads, QG, causal, obj, n_items);
      else
#endif
        OAKE_LAUNCH((attention_coop_kernel<bf16_t, 4>), grid, blk, 0, s, reinterpret_cast<const bf16_t*>(qkv),
                    reinterpret_cast<bf16_t*>(out), L, heads, QG, causal, obj, n_items);
    } else {
      return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if (bits.pair && bits.use_tr && L <= 64 && !causal && qkv_y == nullptr) {
    if (dtype16 == DT_F16) return attn_pair_launch_t<f16_t>(qkv, out, n, L, heads, s);
    if (dtype16 == DT_BF16) return attn_pair_launch_t<bf16_t>(qkv, out, n, L, heads, s);
    return hipErrorInvalidValue;
  }
  if (dtype16 == DT_F16) {
#if OAKE_LAB
    if (!bits.q32) attn_launch_t<f16_t, 4>(qkv, out, n, L, heads, causal, s, bits.use_tr);
    else
#endif
      attn_launch_t<f16_t, 2>(qkv, out, n, L, heads, causal, s, bits.use_tr);
  } else if (dtype16 == DT_BF16) {
#if OAKE_LAB
    if (!bits.q32) attn_launch_t<bf16_t, 4>(qkv, out, n, L, heads, causal, s, bits.use_tr);
    else
#endif
      attn_launch_t<bf16_t, 2>(qkv, out, n, L, heads, causal, s, bits.use_tr);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T>
static hipError_t obj_attn_t(const void* qkv_x, const void* qkv_y, const void* mask, int mask_dtype,
                             void* out, int n, int L, int heads, hipStream_t s) {
  const int total = n * heads;
  const dim3 g((total + 3) / 4), b(256);
  if (mask_dtype == DT_F32)
    OAKE_LAUNCH((object_attention_kernel<T, float>), g, b, 0, s,
                       reinterpret_cast<const T*>(qkv_x), reinterpret_cast<const T*>(qkv_y),
                       reinterpret_cast<const float*>(mask), reinterpret_cast<T*>(out), L, heads,
                       total);
  else if (mask_dtype == DT_F16)
    OAKE_LAUNCH((object_attention_kernel<T, f16_t>), g, b, 0, s,
                       reinterpret_cast<const T*>(qkv_x), reinterpret_cast<const T*>(qkv_y),
                       reinterpret_cast<const f16_t*>(mask), reinterpret_cast<T*>(out), L, heads,
                       total);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_object_attention(int dtype16, const void* qkv_x, const void* qkv_y,
                                   const void* mask, int mask_dtype, void* out, int n, int L,
                                   int heads, hipStream_t s, const LaunchOpts* opts) {
  if (n <= 0) return hipSuccess;
  if (L < 2 || L > kObjMaxKeys) return hipErrorInvalidValue;
  if (mask_dtype != DT_F32 && mask_dtype != DT_F16) return hipErrorInvalidValue;
  // K / V of the head by LDS-DMA and the token as one MFMA query tile where attention_head_kernel covers L (objects
  // mode's last layer: 40 -> 12 us per 128 crops against the one-wave VALU kernel below)
  if (const AttnBits bits(opts); bits.head && bits.use_tr && head_supported(L, kHeadNKT, true))
    return launch_attention_head(dtype16, qkv_x, nullptr, n, L, heads, s, qkv_y, mask, mask_dtype, out, 1);
  if (dtype16 == DT_F16) return obj_attn_t<f16_t>(qkv_x, qkv_y, mask, mask_dtype, out, n, L, heads, s);
  if (dtype16 == DT_BF16) return obj_attn_t<bf16_t>(qkv_x, qkv_y, mask, mask_dtype, out, n, L, heads, s);
  return hipErrorInvalidValue;
}

__global__ __launch_bounds__(64) void cu_census_kernel(unsigned* __restrict__ out, int hold_ticks) {
  extern __shared__ char census_lds[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    census_lds[0] = 1;  // (the LDS allocation is what keeps blocks one per CU)
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

hipError_t launch_cu_census(unsigned* out, int nblocks, int hold_us, hipStream_t s) {
  static DynLdsAttr attr;
  constexpr int lds = 96 * 1024;  // more than half a CU's LDS: one block per CU
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(cu_census_kernel), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(cu_census_kernel, dim3(nblocks), dim3(64), lds, s, out, hold_us * 100);  // wall clock: 100 MHz
  return hipGetLastError();
}

hipError_t launch_tr_read_probe(const uint16_t* in, uint16_t* out, hipStream_t s) {
  OAKE_LAUNCH(tr_read_probe_kernel, dim3(1), dim3(64), 0, s, in, out);
  return hipGetLastError();
}

}  // namespace oake
