// resample.hip — Pillow-exact antialiased bicubic crop + resize + ToTensor + Normalize on the GPU.
//
// The reference's DataLoader workers run `preprocess(image.crop(box))` with PIL on the CPU
// (oadp/oake/objects.py:116-127, blocks.py:54-81, globals.py:26-33): Image.crop (zero fill outside
// the image), torchvision Resize(224, BICUBIC) = PIL.Image.resize, CenterCrop, ToTensor, Normalize.
// This file restates Pillow's 8-bit resampler (src/libImaging/Resample.c: precompute_coeffs,
// normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc) so that the device result is
// BIT-EXACT with PIL's uint8 pixels (tests/test_resample_gpu.py compares against PIL itself):
//   * coefficients in double precision with Pillow's exact operation order (no FMA contraction),
//     filter support scaled by the downscale factor (antialiasing), normalised, then converted to
//     22-bit fixed point with Pillow's rounding;
//   * two passes, horizontal then vertical, with the intermediate image rounded and clipped to
//     uint8 exactly like Pillow's temporary image;
//   * the final uint8 pixel goes through ToTensor (/255) and Normalize ((x - mean) / std) in fp32.
// A uint8 HWC source image is uploaded once; every job names its own source image, so all crops of a
// whole flush of images are produced by three kernel launches (coefficients, horizontal,
// vertical+normalise).  crop_normalize_jobs_kernel is the no-resampling special case (the 224x224 blocks
// of a pyramid level): HBM-bound byte work, 8 pixels per thread, 16-byte stores per colour plane.
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

#pragma clang fp contract(off)

constexpr int kPrecisionBits = 32 - 8 - 2;  // Pillow: PRECISION_BITS

__device__ __forceinline__ double bicubic_filter(double x) {
  // Pillow bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// One thread = one output index of one axis of one job: Pillow precompute_coeffs +
// normalize_coeffs_8bpc for that index.  coef layout: [out index][ksize] int32, bounds [out index][2].
__global__ void resample_coeffs_kernel(const ResampleJob* __restrict__ jobs, int njobs,
                                       int32_t* __restrict__ coef, int32_t* __restrict__ bounds) {
  const int job = blockIdx.y;
  const int axis = blockIdx.z;  // 0 = horizontal, 1 = vertical
  const ResampleJob jb = jobs[job];
  const int in_size = axis == 0 ? jb.cw : jb.ch;
  const int out_size = axis == 0 ? jb.rw : jb.rh;
  const int ksize = axis == 0 ? jb.kh : jb.kv;
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  int32_t* k_out = coef + (axis == 0 ? jb.coefh_off : jb.coefv_off) + (long)xx * ksize;
  int32_t* b_out = bounds + (axis == 0 ? jb.boundh_off : jb.boundv_off) + 2L * xx;

  double filterscale, scale;
  filterscale = scale = (double)((float)in_size - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;  // bicubic support 2.0
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      w = bicubic_filter((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
    }
    int32_t kq;
    if (w < 0)
      kq = (int)(-0.5 + w * (1 << kPrecisionBits));
    else
      kq = (int)(0.5 + w * (1 << kPrecisionBits));
    // tap() multiplies 24-bit factors: |k| < 2^23 always holds for the bicubic window (|w| < 2, see tap) — a coefficient
    // outside it could only come from a degenerate window sum and must not be truncated silently
    constexpr int kMaxCoef = (1 << 23) - 1;
    kq = kq > kMaxCoef ? kMaxCoef : (kq < -kMaxCoef ? -kMaxCoef : kq);
    k_out[x] = kq;
  }
  b_out[0] = xmin;
  b_out[1] = xmax;
}

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= kPrecisionBits;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// One filter tap: pixel (0 .. 255) x fixed-point coefficient.  Pillow's 8-bit coefficients are round(w * 2^22)
// (normalize_coeffs_8bpc, PRECISION_BITS = 22) with |w| < 2, i.e. |k| < 2^23: a normalised bicubic weight EXCEEDS 1 where
// the window is cut at an image border (~1.125 when the negative lobe of one side is clipped away), and the bound that
// matters is the 24-bit one — resample_coeffs_kernel clamps to it (never reached by the bicubic filter; a degenerate
// window sum must not wrap silently), tests/test_resample_gpu.py checks max |k| over border windows.  So both factors fit
// 24 signed bits and the product is v_mul_i32_i24 /
// v_mad_i32_i24 — full-rate instructions — where a 32-bit `*` compiles to v_mul_lo_u32 / v_mad_u64_u32 at a quarter of the
// rate (68 + 44 of them per thread of the horizontal pass: the pass was multiply-bound at 0.12 of the HBM peak).  Same
// integers: the low 32 bits of the exact product.
__device__ __forceinline__ int tap(int px, int k) { return __mul24(px, k); }

// 16 bytes from a 4-byte-aligned address (global_load_dwordx4 needs no more)
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// Horizontal pass: temp[job][y][x][c] for y in [0,ch), x in [0,rw).  Source = crop window of the
// HWC image with PIL's zero fill outside.  (Identity when cw == rw: Pillow skips the pass.)
// A thread owns output column x of kHRows consecutive rows: the column's filter window (bounds, coefficients, source
// column range) and the index arithmetic are per column, so they are paid once per four pixels, and where all four
// rows take the fast path their taps share the coefficient loads.  Same integer products per channel, summed in
// 32-bit wrap-around arithmetic (order-free): bit-identical to Pillow as before.  `resample` 1.45 -> 1.08 ms per objects
// step (profiles/r04/ab_session_resample_h_four_rows.log).
constexpr int kHRows = 4;
// the last (short) tap group of a window through the vector path (1) or byte loads (0: round 5; A/B builds)
#ifndef OAKE_RESAMPLE_TAIL_VEC
#define OAKE_RESAMPLE_TAIL_VEC 1
#endif

// one output pixel the general way (rows that leave the image, transposed jobs, windows that hang over the source,
// loads that would end past the image)
__device__ __forceinline__ void resample_h_pixel(const ResampleJob& jb, const uint8_t* __restrict__ img,
                                                 const int32_t* __restrict__ k, int xmin, int cnt, int sy,
                                                 uint8_t* __restrict__ o) {
  const int height = jb.height, width = jb.width;
  const int sy_lim = jb.tr ? width : height, sx_lim = jb.tr ? height : width;
  const long sy_step = jb.tr ? 3 : (long)width * 3, sx_step = jb.tr ? (long)width * 3 : 3;
  const bool row_ok = sy >= 0 && sy < sy_lim;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  // Fast path (the filter window lies inside the source row, job not transposed): the window's pixels are 3 * cnt
  // contiguous bytes — four pixels per 16-byte load from the enclosing 4-byte-aligned address + v_alignbyte, instead
  // of three byte loads and two bounds tests per tap.
  // The load covers up to 3 bytes more than the 12 it uses, so it is taken only where it ends inside the image.
  const int sx_first = jb.sx0 + xmin;
  if (row_ok && !jb.tr && sx_first >= 0 && sx_first + cnt <= sx_lim) {
    const uint8_t* q = img + (long)sy * sy_step + (long)sx_first * 3;
    const uint8_t* img_end = img + (long)height * width * 3;
    int t = 0;
    for (; t < cnt; t += 4) {  // (the last group may be short: see resample_h_kernel)
      const uintptr_t a = reinterpret_cast<uintptr_t>(q + 3 * t);
      const uint8_t* al = reinterpret_cast<const uint8_t*>(a & ~(uintptr_t)3);
      if (al + 16 > img_end) break;
      const unsigned sh = (unsigned)(a & 3);
      const u32x4_a4 d = *reinterpret_cast<const u32x4_a4*>(al);
      const unsigned w0 = __builtin_amdgcn_alignbyte(d[1], d[0], sh), w1 = __builtin_amdgcn_alignbyte(d[2], d[1], sh),
                     w2 = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
      const int k0 = k[t], k1 = t + 1 < cnt ? k[t + 1] : 0, k2 = t + 2 < cnt ? k[t + 2] : 0, k3 = t + 3 < cnt ? k[t + 3] : 0;
      s0 += tap((int)(w0 & 0xffu), k0); s1 += tap((int)((w0 >> 8) & 0xffu), k0); s2 += tap((int)((w0 >> 16) & 0xffu), k0);
      s0 += tap((int)(w0 >> 24), k1); s1 += tap((int)(w1 & 0xffu), k1); s2 += tap((int)((w1 >> 8) & 0xffu), k1);
      s0 += tap((int)((w1 >> 16) & 0xffu), k2); s1 += tap((int)(w1 >> 24), k2); s2 += tap((int)(w2 & 0xffu), k2);
      s0 += tap((int)((w2 >> 8) & 0xffu), k3); s1 += tap((int)((w2 >> 16) & 0xffu), k3); s2 += tap((int)(w2 >> 24), k3);
    }
    for (; t < cnt; ++t) {
      const uint8_t* p = q + 3 * t;
      const int kv = k[t];
      s0 += tap(p[0], kv);
      s1 += tap(p[1], kv);
      s2 += tap(p[2], kv);
    }
  } else if (row_ok) {
    const uint8_t* rowp = img + sy * sy_step;
    for (int t = 0; t < cnt; ++t) {
      const int sx = jb.sx0 + xmin + t;
      if (sx >= 0 && sx < sx_lim) {
        const uint8_t* p = rowp + sx * sx_step;
        const int kv = k[t];
        s0 += tap(p[0], kv);
        s1 += tap(p[1], kv);
        s2 += tap(p[2], kv);
      }
    }
  }
  o[0] = clip8(s0);
  o[1] = clip8(s1);
  o[2] = clip8(s2);
}

// 4 x 4 transpose of one dword per (lane, row) inside every quad of lanes: lane q of a quad ends up with its quad's four
// values of row q.  Two butterfly stages (lane ^ 1, lane ^ 2), each two quad_perm DPP moves + selects; no LDS.
__device__ __forceinline__ void quad_transpose4(unsigned (&m)[4], int lane_in_quad) {
  const bool odd = lane_in_quad & 1, hi = lane_in_quad & 2;
  unsigned s0 = odd ? m[0] : m[1], s1 = odd ? m[2] : m[3];
  unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
  unsigned n0 = odd ? r0 : m[0], n1 = odd ? m[1] : r0, n2 = odd ? r1 : m[2], n3 = odd ? m[3] : r1;
  s0 = hi ? n0 : n2;
  s1 = hi ? n1 : n3;
  r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0x4E, 0xF, 0xF, true);
  m[0] = hi ? r0 : n0;
  m[1] = hi ? r1 : n1;
  m[2] = hi ? n2 : r0;
  m[3] = hi ? n3 : r1;
}

typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));

// Threads: row group yq (kHRows rows) x column x of the row padded to a multiple of four columns (rw4), so that a quad of
// lanes is four consecutive columns of ONE row group.  A lane computes its column for the four rows (packed RGB, one
// dword per row); the quad then transposes and every lane writes the quad's four pixels of ONE row as a single 12-byte
// store.  (Rounds 1-5 wrote twelve single bytes per lane — 64 lanes x 1 byte at a 3-byte stride per instruction; the pass
// ran at 0.14 of the HBM peak with neither its multiplies, nor its loads' latency, nor their number the bound:
// profiles/r05/ab_resample_mul24_objects.log, profiles/r06/resample_kernel_stats_ab.txt.)  The intermediate image's rows
// are jb.tstride = 12 ceil(rw / 4) bytes apart: dword-aligned stores, and the last quad of a row may write its padding.
__global__ __launch_bounds__(256) void resample_h_kernel(const ResampleJob* __restrict__ jobs,
                                                         const int32_t* __restrict__ coef,
                                                         const int32_t* __restrict__ bounds,
                                                         uint8_t* __restrict__ temp) {
  const ResampleJob jb = jobs[blockIdx.y];
  const uint8_t* __restrict__ img = jb.img;
  const int height = jb.height, width = jb.width;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;  // (column x of row group yq; < 2^31: run_resample)
  const unsigned nyq = ((unsigned)jb.ch + kHRows - 1) / kHRows;
  const unsigned rw4 = ((unsigned)jb.rw + 3u) & ~3u;
  if (idx >= nyq * rw4) return;  // (whole quads: rw4 is a multiple of four)
  const int yq = idx / rw4, x = idx - (unsigned)yq * rw4;
  const int y0 = yq * kHRows;
  const int nrows = jb.ch - y0 < kHRows ? jb.ch - y0 : kHRows;
  // job space (sy, sx) -> image (row, column); a transposed job walks image columns
  const int sy_lim = jb.tr ? width : height, sx_lim = jb.tr ? height : width;
  const long sy_step = jb.tr ? 3 : (long)width * 3, sx_step = jb.tr ? (long)width * 3 : 3;
  unsigned px[kHRows] = {0u, 0u, 0u, 0u};  // this column's pixel of each row: r | g << 8 | b << 16
  if (x < jb.rw) {
    if (jb.cw == jb.rw) {
      const int sx = jb.sx0 + x;
      for (int r = 0; r < nrows; ++r) {
        const int sy = jb.sy0 + y0 + r;
        const bool ok = sy >= 0 && sy < sy_lim && sx >= 0 && sx < sx_lim;
        const uint8_t* p = img + sy * sy_step + sx * sx_step;
        const unsigned v = ok ? (unsigned)p[0] | (unsigned)p[1] << 8 | (unsigned)p[2] << 16 : 0u;
#pragma unroll
        for (int rr = 0; rr < kHRows; ++rr)
          if (rr == r) px[rr] = v;
      }
    } else {
      const int32_t* k = coef + jb.coefh_off + (long)x * jb.kh;
      const int xmin = bounds[jb.boundh_off + 2L * x], cnt = bounds[jb.boundh_off + 2L * x + 1];
      const int sx_first = jb.sx0 + xmin, sy_first = jb.sy0 + y0;
      // All four rows inside the image, window inside the row, job not transposed: the rows' taps share the coefficient
      // loads; a tap group whose 16-byte load would end past the image (the last image row only) takes byte loads.
      if (nrows == kHRows && !jb.tr && sy_first >= 0 && sy_first + kHRows <= sy_lim && sx_first >= 0 && sx_first + cnt <= sx_lim) {
        const uint8_t* q = img + (long)sy_first * sy_step + (long)sx_first * 3;
        const uint8_t* img_end = img + (long)height * width * 3;
        int acc[kHRows][3];
#pragma unroll
        for (int r = 0; r < kHRows; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (kPrecisionBits - 1);
        int t = 0;
        // Taps in groups of four, the LAST group too when the window's length is not a multiple of four (5, 7, 9 taps are
        // the common lengths): its 16-byte loads cover up to three pixels past the window — inside the image buffer, or
        // the group is left to the byte loop below — whose coefficients are taken as 0, so the same integers are summed.
        // The coefficient reads of a short last group may run into the next column's row of the table (the table carries
        // 16 bytes of slack behind the last one, run_resample): discarded by the select.
        for (; OAKE_RESAMPLE_TAIL_VEC ? t < cnt : t + 4 <= cnt; t += 4) {
          const uintptr_t a_last = reinterpret_cast<uintptr_t>(q + (kHRows - 1) * sy_step + 3 * t);
          if (reinterpret_cast<const uint8_t*>(a_last & ~(uintptr_t)3) + 16 > img_end) break;
          const int k0 = k[t], k1 = t + 1 < cnt ? k[t + 1] : 0, k2 = t + 2 < cnt ? k[t + 2] : 0, k3 = t + 3 < cnt ? k[t + 3] : 0;
#pragma unroll
          for (int r = 0; r < kHRows; ++r) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(q + r * sy_step + 3 * t);
            const unsigned sh = (unsigned)(a & 3);
            const u32x4_a4 d = *reinterpret_cast<const u32x4_a4*>(a & ~(uintptr_t)3);
            const unsigned w0 = __builtin_amdgcn_alignbyte(d[1], d[0], sh), w1 = __builtin_amdgcn_alignbyte(d[2], d[1], sh),
                           w2 = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
            acc[r][0] += tap((int)(w0 & 0xffu), k0); acc[r][1] += tap((int)((w0 >> 8) & 0xffu), k0); acc[r][2] += tap((int)((w0 >> 16) & 0xffu), k0);
            acc[r][0] += tap((int)(w0 >> 24), k1); acc[r][1] += tap((int)(w1 & 0xffu), k1); acc[r][2] += tap((int)((w1 >> 8) & 0xffu), k1);
            acc[r][0] += tap((int)((w1 >> 16) & 0xffu), k2); acc[r][1] += tap((int)(w1 >> 24), k2); acc[r][2] += tap((int)(w2 & 0xffu), k2);
            acc[r][0] += tap((int)((w2 >> 8) & 0xffu), k3); acc[r][1] += tap((int)((w2 >> 16) & 0xffu), k3); acc[r][2] += tap((int)(w2 >> 24), k3);
          }
        }
        for (; t < cnt; ++t) {
          const int kv = k[t];
#pragma unroll
          for (int r = 0; r < kHRows; ++r) {
            const uint8_t* p = q + r * sy_step + 3 * t;
            acc[r][0] += tap(p[0], kv);
            acc[r][1] += tap(p[1], kv);
            acc[r][2] += tap(p[2], kv);
          }
        }
#pragma unroll
        for (int r = 0; r < kHRows; ++r)
          px[r] = (unsigned)clip8(acc[r][0]) | (unsigned)clip8(acc[r][1]) << 8 | (unsigned)clip8(acc[r][2]) << 16;
      } else {
        for (int r = 0; r < nrows; ++r) {
          uint8_t o3[3];
          resample_h_pixel(jb, img, k, xmin, cnt, sy_first + r, o3);
          const unsigned v = (unsigned)o3[0] | (unsigned)o3[1] << 8 | (unsigned)o3[2] << 16;
#pragma unroll
          for (int rr = 0; rr < kHRows; ++rr)
            if (rr == r) px[rr] = v;
        }
      }
    }
  }
  // quad transpose: lane q of the quad now holds columns x0 .. x0 + 3 of row y0 + q; one 12-byte store
  const int ql = threadIdx.x & 3;
  quad_transpose4(px, ql);
  if (ql < nrows) {
    const int x0 = x & ~3;
    u32x3_a4 w;
    w[0] = px[0] | px[1] << 24;
    w[1] = px[1] >> 8 | px[2] << 16;
    w[2] = px[2] >> 16 | px[3] << 8;
    *reinterpret_cast<u32x3_a4*>(temp + jb.temp_off + (long)(y0 + ql) * jb.tstride + (long)x0 * 3) = w;
  }
}

// Vertical pass + CenterCrop + ToTensor + Normalize: out[job][c][oy][ox].
template <typename TOUT>
__global__ __launch_bounds__(256) void resample_v_kernel(const ResampleJob* __restrict__ jobs,
                                                         const int32_t* __restrict__ coef,
                                                         const int32_t* __restrict__ bounds,
                                                         const uint8_t* __restrict__ temp, int out_size,
                                                         float m0, float m1, float m2, float d0,
                                                         float d1, float d2, TOUT* __restrict__ out) {
  const ResampleJob jb = jobs[blockIdx.y];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out_size * out_size) return;
  const int oy = p / out_size, ox = p - oy * out_size;
  // position in the resized image, in job space (a transposed job: output (oy, ox) is its (ox, oy))
  const int ry = (jb.tr ? ox : oy) + jb.cy, rx = (jb.tr ? oy : ox) + jb.cx;
  float r = 0.f, g = 0.f, b = 0.f;
  if (ry >= 0 && ry < jb.rh && rx >= 0 && rx < jb.rw) {
    const uint8_t* tcol = temp + jb.temp_off + (long)rx * 3;
    int v0, v1, v2;
    if (jb.ch == jb.rh) {
      const uint8_t* q = tcol + (long)ry * jb.tstride;
      v0 = q[0]; v1 = q[1]; v2 = q[2];
    } else {
      const int32_t* k = coef + jb.coefv_off + (long)ry * jb.kv;
      const int ymin = bounds[jb.boundv_off + 2L * ry], cnt = bounds[jb.boundv_off + 2L * ry + 1];
      int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
      for (int t = 0; t < cnt; ++t) {
        const uint8_t* q = tcol + (long)(ymin + t) * jb.tstride;
        const int kv = k[t];
        s0 += tap(q[0], kv);
        s1 += tap(q[1], kv);
        s2 += tap(q[2], kv);
      }
      v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    }
    r = (float)v0; g = (float)v1; b = (float)v2;
  }
  const size_t plane = (size_t)out_size * out_size;
  TOUT* o = out + (size_t)jb.out_row * 3 * plane + p;
  o[0] = (TOUT)((r / 255.0f - m0) / d0);
  o[plane] = (TOUT)((g / 255.0f - m1) / d1);
  o[2 * plane] = (TOUT)((b / 255.0f - m2) / d2);
}

// The same pass, four consecutive output pixels of one row per thread (out_size % 4 == 0).  The 4 x RGB source
// bytes of a filter tap are 12 contiguous bytes of the intermediate image: one 16-byte load from the enclosing
// 4-byte-aligned address + v_alignbyte instead of twelve byte loads, and one 8-byte (f16) / 16-byte (f32) store per
// colour plane instead of four scalar ones.  Same integer arithmetic in the same order: bit-identical.  Threads
// whose four pixels are not all inside the resized image (crop windows hanging over it) and transposed jobs take
// the per-pixel path.  (The 16-byte load starts at the 4-byte-aligned address at or below the window: it may touch up to 4
// bytes past the 12 it needs; run_resample asks for 16 bytes more than the temp images take, api.hip.)
template <typename TOUT>
__global__ __launch_bounds__(256) void resample_v4_kernel(const ResampleJob* __restrict__ jobs,
                                                          const int32_t* __restrict__ coef,
                                                          const int32_t* __restrict__ bounds,
                                                          const uint8_t* __restrict__ temp, int out_size,
                                                          float m0, float m1, float m2, float d0,
                                                          float d1, float d2, TOUT* __restrict__ out) {
  const ResampleJob jb = jobs[blockIdx.y];
  const int q4 = out_size >> 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= out_size * q4) return;
  const int oy = t / q4, ox0 = (t - oy * q4) << 2;
  float v[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[c][i] = 0.f;
  const int ry0 = oy + jb.cy, rx0 = ox0 + jb.cx;
  if (!jb.tr && ry0 >= 0 && ry0 < jb.rh && rx0 >= 0 && rx0 + 4 <= jb.rw) {
    const uint8_t* tcol = temp + jb.temp_off + (long)rx0 * 3;
    const long rstep = jb.tstride;
    int acc[12];
    auto load12 = [](const uint8_t* q, unsigned (&w)[3]) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(q);
      const unsigned sh = (unsigned)(a & 3);
      const u32x4_a4 d = *reinterpret_cast<const u32x4_a4*>(a & ~(uintptr_t)3);
      w[0] = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
      w[1] = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
      w[2] = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
    };
    if (jb.ch == jb.rh) {
      unsigned w[3];
      load12(tcol + (long)ry0 * rstep, w);
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = (int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
    } else {
      const int32_t* k = coef + jb.coefv_off + (long)ry0 * jb.kv;
      const int ymin = bounds[jb.boundv_off + 2L * ry0], cnt = bounds[jb.boundv_off + 2L * ry0 + 1];
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = 1 << (kPrecisionBits - 1);
      for (int tt = 0; tt < cnt; ++tt) {
        unsigned w[3];
        load12(tcol + (long)(ymin + tt) * rstep, w);
        const int kv = k[tt];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] += tap((int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu), kv);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = clip8(acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c][i] = (float)acc[3 * i + c];
  } else {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int ox = ox0 + i;
      const int ry = (jb.tr ? ox : oy) + jb.cy, rx = (jb.tr ? oy : ox) + jb.cx;
      if (ry >= 0 && ry < jb.rh && rx >= 0 && rx < jb.rw) {
        const uint8_t* tcol = temp + jb.temp_off + (long)rx * 3;
        int v0, v1, v2;
        if (jb.ch == jb.rh) {
          const uint8_t* q = tcol + (long)ry * jb.tstride;
          v0 = q[0]; v1 = q[1]; v2 = q[2];
        } else {
          const int32_t* k = coef + jb.coefv_off + (long)ry * jb.kv;
          const int ymin = bounds[jb.boundv_off + 2L * ry], cnt = bounds[jb.boundv_off + 2L * ry + 1];
          int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
          for (int tt = 0; tt < cnt; ++tt) {
            const uint8_t* q = tcol + (long)(ymin + tt) * jb.tstride;
            const int kv = k[tt];
            s0 += tap(q[0], kv);
            s1 += tap(q[1], kv);
            s2 += tap(q[2], kv);
          }
          v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
        }
        // (runtime index into v: four selects per channel; this is the rare path)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j == i) { v[0][j] = (float)v0; v[1][j] = (float)v1; v[2][j] = (float)v2; }
      }
    }
  }
  const size_t plane = (size_t)out_size * out_size;
  TOUT* o = out + (size_t)jb.out_row * 3 * plane + (size_t)oy * out_size + ox0;
  const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    alignas(16) TOUT r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (TOUT)((v[c][i] / 255.0f - mean[c]) / sd[c]);
    if constexpr (sizeof(TOUT) == 2)
      *reinterpret_cast<uint2*>(o + c * plane) = *reinterpret_cast<const uint2*>(r);
    else
      *reinterpret_cast<uint4*>(o + c * plane) = *reinterpret_cast<const uint4*>(r);
  }
}

// resample_v4_kernel writing straight into the ZERO-PADDED 16-bit batch conv1 gathers its patches from when the
// convolution pads (objects mode: stride 16, padding 15 -> [n,3,254,256], csrc/gemm.hip patch gather; SURVEY §8 A10/A15a,
// [REF oadp/oake/objects.py:116-127,298-301]): out[job][c][opad + oy][opad + ox], rows `ows` pixels apart, planes `ohp`
// rows.  The separate pad pass (pad_nchw_kernel: 0.53 ms per objects step, reads the dense crops and writes this buffer) is
// then not run.  A thread owns four consecutive PADDED columns 4j .. 4j + 3 of one image row — an 8-byte-aligned store per
// colour plane, where image column groups (4k + opad) would start 2 bytes past a dword — i.e. image columns 4j - opad + i;
// columns outside the image are written as literal zeros: they are the convolution's padding.  A row is written WHOLE
// (ows / 4 groups = one wave per 256-pixel row): with only the 57 groups that touch the image, the first and last 128-byte
// line of every 512-byte row were partial writes — a read-modify-write per line at the memory side: the pass took 0.49 ms
// more than the dense one per objects step, all that dropping pad_nchw had saved (profiles/r06/ab_padded_crops_objects.log).
// The border ROWS keep the zeros the buffer was created with.
template <typename TOUT>
__global__ __launch_bounds__(256) void resample_v4p_kernel(const ResampleJob* __restrict__ jobs,
                                                           const int32_t* __restrict__ coef,
                                                           const int32_t* __restrict__ bounds,
                                                           const uint8_t* __restrict__ temp, int out_size,
                                                           float m0, float m1, float m2, float d0,
                                                           float d1, float d2, TOUT* __restrict__ out, int opad,
                                                           int ohp, int ows) {
  const ResampleJob jb = jobs[blockIdx.y];
  const int ngroups = ows >> 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= out_size * ngroups) return;
  const int oy = t / ngroups, jg = t - oy * ngroups;
  const int ox0 = 4 * jg - opad;  // image column of the group's first pixel (may be negative / reach past the image)
  float v[3][4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ok[i] = ox0 + i >= 0 && ox0 + i < out_size;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c][i] = 0.f;
  }
  const int ry0 = oy + jb.cy, rx0 = ox0 + jb.cx;
  // The groups that straddle the image's left / right edge (one of each per row = two lanes of every wave) take the vector
  // path too: their 12-byte window starts up to 3 pixels before / ends up to 3 pixels past the valid columns — bytes of the
  // neighbouring row of the intermediate image (or its 16 bytes of slack), loaded, multiplied and then discarded by ok[].
  // (Sent down the per-pixel path, those two lanes made every wave wait for four serial tap loops: the pass took 1.09 ms
  // per objects step against 0.50 for the dense one.)  Only a window that would start before the buffer does — the first
  // row of the first job — is left to the per-pixel path.
  const int lo = ox0 < 0 ? 0 : ox0, hi = ox0 + 4 > out_size ? out_size : ox0 + 4;  // valid image columns [lo, hi)
  const long rstep = jb.tstride;
  const long first_off = jb.temp_off + (long)(jb.ch == jb.rh ? (ry0 < 0 ? 0 : ry0) : 0) * rstep + (long)rx0 * 3;
  if (!jb.tr && lo < hi && ry0 >= 0 && ry0 < jb.rh && lo + jb.cx >= 0 && hi + jb.cx <= jb.rw && first_off >= 0) {
    const uint8_t* tcol = temp + jb.temp_off + (long)rx0 * 3;
    int acc[12];
    auto load12 = [](const uint8_t* q, unsigned (&w)[3]) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(q);
      const unsigned sh = (unsigned)(a & 3);
      const u32x4_a4 d = *reinterpret_cast<const u32x4_a4*>(a & ~(uintptr_t)3);
      w[0] = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
      w[1] = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
      w[2] = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
    };
    if (jb.ch == jb.rh) {
      unsigned w[3];
      load12(tcol + (long)ry0 * rstep, w);
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = (int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
    } else {
      const int32_t* k = coef + jb.coefv_off + (long)ry0 * jb.kv;
      const int ymin = bounds[jb.boundv_off + 2L * ry0], cnt = bounds[jb.boundv_off + 2L * ry0 + 1];
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = 1 << (kPrecisionBits - 1);
      for (int tt = 0; tt < cnt; ++tt) {
        unsigned w[3];
        load12(tcol + (long)(ymin + tt) * rstep, w);
        const int kv = k[tt];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] += tap((int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu), kv);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = clip8(acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c][i] = (float)acc[3 * i + c];
  } else {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int ox = ox0 + i;
      if (ox < 0 || ox >= out_size) continue;
      const int ry = (jb.tr ? ox : oy) + jb.cy, rx = (jb.tr ? oy : ox) + jb.cx;
      if (ry >= 0 && ry < jb.rh && rx >= 0 && rx < jb.rw) {
        const uint8_t* tcol = temp + jb.temp_off + (long)rx * 3;
        int v0, v1, v2;
        if (jb.ch == jb.rh) {
          const uint8_t* q = tcol + (long)ry * jb.tstride;
          v0 = q[0]; v1 = q[1]; v2 = q[2];
        } else {
          const int32_t* k = coef + jb.coefv_off + (long)ry * jb.kv;
          const int ymin = bounds[jb.boundv_off + 2L * ry], cnt = bounds[jb.boundv_off + 2L * ry + 1];
          int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
          for (int tt = 0; tt < cnt; ++tt) {
            const uint8_t* q = tcol + (long)(ymin + tt) * jb.tstride;
            const int kv = k[tt];
            s0 += tap(q[0], kv);
            s1 += tap(q[1], kv);
            s2 += tap(q[2], kv);
          }
          v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (jj == i) { v[0][jj] = (float)v0; v[1][jj] = (float)v1; v[2][jj] = (float)v2; }
      }
    }
  }
  const size_t plane = (size_t)ohp * ows;
  TOUT* o = out + (size_t)jb.out_row * 3 * plane + (size_t)(oy + opad) * ows + 4 * jg;
  const float mean[3] = {m0, m1, m2}, sd[3] = {d0, d1, d2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    alignas(16) TOUT r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ok[i] ? (TOUT)((v[c][i] / 255.0f - mean[c]) / sd[c]) : (TOUT)0.f;
    if constexpr (sizeof(TOUT) == 2)
      *reinterpret_cast<uint2*>(o + c * plane) = *reinterpret_cast<const uint2*>(r);
    else
      *reinterpret_cast<uint4*>(o + c * plane) = *reinterpret_cast<const uint4*>(r);
  }
}

// Vertical pass to a uint8 HWC image (whole-image resize for the blocks pyramid), four consecutive pixels of one output
// row per thread (jobs that are not transposed): a tap's 4 x RGB
// source bytes are 12 contiguous bytes of the intermediate image (resample_v4_kernel's load), and the four result pixels
// leave as ONE 12-byte store at the pixel's own byte alignment (the pyramid level's rows are 3 * width bytes apart:
// dword-unaligned in general, which gfx950 serves) instead of twelve single-byte stores.  The last, partial group of a
// row and transposed jobs go pixel by pixel.  Same integers as Pillow's: bit-identical.
__global__ __launch_bounds__(256) void resample_v4_u8_kernel(const ResampleJob* __restrict__ jobs,
                                                             const int32_t* __restrict__ coef,
                                                             const int32_t* __restrict__ bounds,
                                                             const uint8_t* __restrict__ temp) {
  const ResampleJob jb = jobs[blockIdx.y];
  const int ow = jb.tr ? jb.rh : jb.rw, oh = jb.tr ? jb.rw : jb.rh;  // the OUTPUT image: oh rows of ow pixels
  const int q4 = (ow + 3) >> 2;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)oh * q4) return;
  const int orow = (int)(t / q4), oc0 = (int)(t - (long)orow * q4) << 2;
  uint8_t* o = jb.u8_out + ((long)orow * ow + oc0) * 3;
  if (!jb.tr && oc0 + 4 <= ow) {
    const int ry = orow, rx0 = oc0;
    const uint8_t* tcol = temp + jb.temp_off + (long)rx0 * 3;
    const long rstep = jb.tstride;
    unsigned w[3];
    auto load12 = [](const uint8_t* q, unsigned (&ww)[3]) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(q);
      const unsigned sh = (unsigned)(a & 3);
      const u32x4_a4 d = *reinterpret_cast<const u32x4_a4*>(a & ~(uintptr_t)3);
      ww[0] = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
      ww[1] = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
      ww[2] = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
    };
    typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
    if (jb.ch == jb.rh) {
      load12(tcol + (long)ry * rstep, w);
    } else {
      const int32_t* k = coef + jb.coefv_off + (long)ry * jb.kv;
      const int ymin = bounds[jb.boundv_off + 2L * ry], cnt = bounds[jb.boundv_off + 2L * ry + 1];
      int acc[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = 1 << (kPrecisionBits - 1);
      for (int tt = 0; tt < cnt; ++tt) {
        unsigned x[3];
        load12(tcol + (long)(ymin + tt) * rstep, x);
        const int kv = k[tt];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] += tap((int)((x[i >> 2] >> (8 * (i & 3))) & 0xffu), kv);
      }
      // (the clipped bytes go through an opaque asm before they are packed: hipcc / ROCm 7.2 matches
      // "two saturated (x >> 22) side by side" to v_ashr_pk_u8_i32 and then ORs bytes 2 and 3 into a register whose upper
      // half that instruction does not clear — wrong bytes 2 / 3 of every dword, found by tests/test_resample_gpu.py)
      unsigned b8[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        b8[i] = clip8(acc[i]);
        asm volatile("" : "+v"(b8[i]));
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) w[j] = b8[4 * j] | b8[4 * j + 1] << 8 | b8[4 * j + 2] << 16 | b8[4 * j + 3] << 24;
    }
    u32x3_a1 out3;
    out3[0] = w[0]; out3[1] = w[1]; out3[2] = w[2];
    *reinterpret_cast<u32x3_a1*>(o) = out3;
    return;
  }
  for (int i = 0; i < 4 && oc0 + i < ow; ++i) {
    const int ocol = oc0 + i;
    const int ry = jb.tr ? ocol : orow, rx = jb.tr ? orow : ocol;
    const uint8_t* tcol = temp + jb.temp_off + (long)rx * 3;
    uint8_t* op = o + 3 * i;
    if (jb.ch == jb.rh) {
      const uint8_t* q = tcol + (long)ry * jb.tstride;
      op[0] = q[0]; op[1] = q[1]; op[2] = q[2];
      continue;
    }
    const int32_t* k = coef + jb.coefv_off + (long)ry * jb.kv;
    const int ymin = bounds[jb.boundv_off + 2L * ry], cnt = bounds[jb.boundv_off + 2L * ry + 1];
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int tt = 0; tt < cnt; ++tt) {
      const uint8_t* q = tcol + (long)(ymin + tt) * jb.tstride;
      const int kv = k[tt];
      s0 += tap(q[0], kv);
      s1 += tap(q[1], kv);
      s2 += tap(q[2], kv);
    }
    op[0] = clip8(s0);
    op[1] = clip8(s1);
    op[2] = clip8(s2);
  }
}

// Exact-size crop + ToTensor + Normalize, many images per launch.  One thread = 8 consecutive pixels of
// one output row, all three channels: 24 source bytes (contiguous in HWC) in, one 16-byte (f16) or two
// 16-byte (f32) stores per colour plane out.  out_size must be a multiple of 8.
template <typename TOUT>
__global__ __launch_bounds__(256) void crop_normalize_jobs_kernel(const CropJob* __restrict__ jobs,
                                                                  int out_size, float m0, float m1, float m2,
                                                                  float s0, float s1, float s2,
                                                                  TOUT* __restrict__ out) {
  const CropJob jb = jobs[blockIdx.y];
  const int per_row = out_size >> 3;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= out_size * per_row) return;
  const int oy = t / per_row, ox = (t - oy * per_row) << 3;
  const int sy = jb.y1 + oy, sx = jb.x1 + ox;
  float v[3][8];
  const bool row_ok = sy >= 0 && sy < jb.height;
  const uint8_t* rowp = jb.img + (size_t)(row_ok ? sy : 0) * jb.width * 3;
  if (row_ok && sx >= 0 && sx + 8 <= jb.width) {
    // the eight pixels are 24 contiguous bytes inside the row: one 16-byte and one 8-byte load at whatever alignment the
    // crop origin has (gfx950 serves unaligned global accesses) instead of 24 single-byte loads — round 5 measured the
    // kernel at 0.33 of the HBM peak, 5 % of a blocks step
    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u32x2_a1 __attribute__((ext_vector_type(2), aligned(1)));
    const uint8_t* q = rowp + (size_t)sx * 3;
    const u32x4_a1 d0 = *reinterpret_cast<const u32x4_a1*>(q);
    const u32x2_a1 d1 = *reinterpret_cast<const u32x2_a1*>(q + 16);
    const unsigned w[6] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1]};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int b = 3 * i + c;
        v[c][i] = (float)((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
      }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int x = sx + i;
      const bool ok = row_ok && x >= 0 && x < jb.width;  // PIL crop pads with zeros outside the image
      const uint8_t* px = rowp + (size_t)(ok ? x : 0) * 3;
      v[0][i] = ok ? (float)px[0] : 0.f;
      v[1][i] = ok ? (float)px[1] : 0.f;
      v[2][i] = ok ? (float)px[2] : 0.f;
    }
  }
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  const size_t plane = (size_t)out_size * out_size;
  TOUT* o = out + (size_t)jb.out_row * 3 * plane + (size_t)oy * out_size + ox;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // ToTensor: /255 in fp32 ; Normalize: (x - mean) / std in fp32 — torchvision's operation order
    alignas(16) TOUT r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (TOUT)((v[c][i] / 255.0f - mean[c]) / sd[c]);
    if constexpr (sizeof(TOUT) == 2) {
      *reinterpret_cast<uint4*>(o + c * plane) = *reinterpret_cast<const uint4*>(r);
    } else {
      *reinterpret_cast<uint4*>(o + c * plane) = *reinterpret_cast<const uint4*>(r);
      *reinterpret_cast<uint4*>(o + c * plane + 4) = *reinterpret_cast<const uint4*>(r + 4);
    }
  }
}

}  // namespace

// gridDim.y carries the job index and is limited to 65535: a flush's job list (all proposals / all block crops
// of a whole batch of images) is launched in slices of at most kMaxJobsPerLaunch jobs.  Jobs are
// self-describing (source image, scratch offsets, output row), so a slice is just a pointer offset.
constexpr int kMaxJobsPerLaunch = 65535;

hipError_t launch_crop_normalize_jobs(const CropJob* d_jobs, int njobs, int out_size, const float* mean3,
                                      const float* std3, void* out, int out_dtype, hipStream_t s) {
  if (njobs <= 0) return hipSuccess;
  if (out_size <= 0 || out_size % 8 != 0) return hipErrorInvalidValue;
  if (out_dtype != DT_F32 && out_dtype != DT_F16) return hipErrorInvalidValue;
  for (int j0 = 0; j0 < njobs; j0 += kMaxJobsPerLaunch) {
    const int nj = njobs - j0 < kMaxJobsPerLaunch ? njobs - j0 : kMaxJobsPerLaunch;
    const dim3 g((out_size * (out_size / 8) + 255) / 256, nj), b(256);
    if (out_dtype == DT_F32)
      hipLaunchKernelGGL(crop_normalize_jobs_kernel<float>, g, b, 0, s, d_jobs + j0, out_size, mean3[0], mean3[1],
                         mean3[2], std3[0], std3[1], std3[2], reinterpret_cast<float*>(out));
    else
      hipLaunchKernelGGL(crop_normalize_jobs_kernel<f16_t>, g, b, 0, s, d_jobs + j0, out_size, mean3[0], mean3[1],
                         mean3[2], std3[0], std3[1], std3[2], reinterpret_cast<f16_t*>(out));
  }
  return hipGetLastError();
}

hipError_t launch_resample(const ResampleJob* d_jobs, int njobs, int max_out, long max_chq_rw, long max_rh_rw,
                           int32_t* d_coef, int32_t* d_bounds, uint8_t* d_temp, int out_size,
                           const float* mean3, const float* std3, void* out, int out_dtype, hipStream_t s,
                           int out_pad, int out_hp, int out_ws) {
  if (njobs <= 0) return hipSuccess;
  if (out_dtype != DT_U8 && out_dtype != DT_F32 && out_dtype != DT_F16) return hipErrorInvalidValue;
  const bool padded = out_ws > 0;  // rows of the zero-padded batch conv1 gathers from (resample_v4p_kernel)
  if (padded && (out_dtype != DT_F16 || out_size % 4 != 0 || out_pad < 0 || out_ws % 4 != 0 ||
                 out_ws < ((out_pad + out_size + 3) & ~3) || out_hp < out_size + out_pad ||
                 reinterpret_cast<uintptr_t>(out) % 16 != 0))
    return hipErrorInvalidValue;
  for (int j0 = 0; j0 < njobs; j0 += kMaxJobsPerLaunch) {
    const int nj = njobs - j0 < kMaxJobsPerLaunch ? njobs - j0 : kMaxJobsPerLaunch;
    const ResampleJob* jobs = d_jobs + j0;
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3((max_out + 63) / 64, nj, 2), dim3(64), 0, s,
                       jobs, nj, d_coef, d_bounds);
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((max_chq_rw + 255) / 256), nj), dim3(256), 0, s,
                       jobs, d_coef, d_bounds, d_temp);
    if (out_dtype == DT_U8) {  // whole-image resizes: every job writes its own image
      // (four pixels per thread: at most max_rh_rw / 4 + one partial group per output row <= max_out rows of a job)
      hipLaunchKernelGGL(resample_v4_u8_kernel, dim3((unsigned)((max_rh_rw / 4 + max_out + 255) / 256), nj), dim3(256), 0,
                         s, jobs, d_coef, d_bounds, d_temp);
      continue;
    }
    if (padded) {
      const int ngroups = out_ws >> 2;  // (whole rows: full 128-byte lines)
      hipLaunchKernelGGL(resample_v4p_kernel<f16_t>, dim3((out_size * ngroups + 255) / 256, nj), dim3(256), 0, s, jobs,
                         d_coef, d_bounds, d_temp, out_size, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                         reinterpret_cast<f16_t*>(out), out_pad, out_hp, out_ws);
      continue;
    }
    const bool v4 = out_size % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
    const dim3 g(((v4 ? out_size * (out_size / 4) : out_size * out_size) + 255) / 256, nj), b(256);
    if (out_dtype == DT_F32) {
      if (v4)
        hipLaunchKernelGGL(resample_v4_kernel<float>, g, b, 0, s, jobs, d_coef, d_bounds, d_temp,
                           out_size, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                           reinterpret_cast<float*>(out));
      else
        hipLaunchKernelGGL(resample_v_kernel<float>, g, b, 0, s, jobs, d_coef, d_bounds, d_temp,
                           out_size, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                           reinterpret_cast<float*>(out));
    } else {
      if (v4)
        hipLaunchKernelGGL(resample_v4_kernel<f16_t>, g, b, 0, s, jobs, d_coef, d_bounds, d_temp,
                           out_size, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                           reinterpret_cast<f16_t*>(out));
      else
        hipLaunchKernelGGL(resample_v_kernel<f16_t>, g, b, 0, s, jobs, d_coef, d_bounds, d_temp,
                           out_size, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                           reinterpret_cast<f16_t*>(out));
    }
  }
  return hipGetLastError();
}

}  // namespace oake
