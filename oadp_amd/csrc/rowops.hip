// rowops.hip — the HBM/L2-bound glue of the encoder (SURVEY.md §8 A15b,c,h): LayerNorm rows with
// wavefront-shuffle reductions, CLS/pos-emb embedding + ln_pre, im2col for conv1, the
// ln_post -> proj -> L2-normalise head, weight casts, and the uint8 crop+normalise gather.
// Every kernel moves 8-16 B per lane per access and keeps the row statistics in registers.
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

constexpr float kLnEps = 1e-5f;
constexpr int kMaxVec = 4;  // float4 per lane -> rows up to 4*4*64 = 1024 wide

// Loads one fp32 row (c = 4*nv floats) into v[], returns mean/rstd via wave shuffles.
__device__ __forceinline__ void ln_stats(const float4 (&v)[kMaxVec], int lane, int nv, int c,
                                         float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  s = wave_sum(s);
  mean = s / (float)c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  q = wave_sum(q);
  rstd = rsqrtf(q / (float)c + kLnEps);
}

// 4 consecutive elements of a row of the residual stream (fp32 or 16-bit) <-> float4
template <typename TX>
__device__ __forceinline__ float4 load4(const TX* row, int i4) {
  if constexpr (sizeof(TX) == 4) {
    return reinterpret_cast<const float4*>(row)[i4];
  } else {
    typedef TX v4 __attribute__((ext_vector_type(4)));
    const v4 v = __builtin_bit_cast(v4, reinterpret_cast<const uint2*>(row)[i4]);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
}
template <typename TX>
__device__ __forceinline__ void store4(TX* row, int i4, float a, float b, float c, float d) {
  if constexpr (sizeof(TX) == 4) {
    reinterpret_cast<float4*>(row)[i4] = make_float4(a, b, c, d);
  } else {
    reinterpret_cast<uint2*>(row)[i4] = pack4<TX>(a, b, c, d);
  }
}

template <typename T, typename TX>
__global__ __launch_bounds__(256) void layernorm_kernel(const TX* __restrict__ x, long stride,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        T* __restrict__ y, int rows, int c) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = c >> 2;
  const TX* xr = x + (size_t)row * stride;
  float4 v[kMaxVec];
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) v[i] = load4<TX>(xr, lane + 64 * i);
  float mean, rstd;
  ln_stats(v, lane, nv, c, mean, rstd);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  uint2* yr = reinterpret_cast<uint2*>(y + (size_t)row * c);
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      const float4 g = g4[lane + 64 * i], b = b4[lane + 64 * i];
      yr[lane + 64 * i] = pack4<T>((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                   (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
    }
}

// ---- LayerNorm folded into the consuming GEMM (16-bit residual stream) -------------------------
// LN(x) W^T + b  =  rstd * (x (g o W)^T)  -  rstd * mean * colsum  +  (b + W beta)      with
// colsum[n] = sum_k (g o W)[n,k]: the GEMM runs on the RAW residual rows with gamma-scaled weights
// and its epilogue applies the two per-row scalars below, so the normalised activations are never
// written to / re-read from HBM (2 x 19.7 MB per LayerNorm at bs 256) — only these 8 B per row.
// stat[row] = (rstd, -mean * rstd); statistics of the row exactly as the GEMM will read it.
// (SUMS: stat[row * 16] = (sum x, sum x^2) instead — slot 0 of the GEMM-to-GEMM row-statistics
// buffer, see kernels.h GemmArgs::rowpart_in)
template <typename TX, bool SUMS>
__global__ __launch_bounds__(256) void rowstat_kernel(const TX* __restrict__ x, long stride,
                                                      float2* __restrict__ stat, int rows, int c) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = c >> 2;
  const TX* xr = x + (size_t)row * stride;
  float4 v[kMaxVec];
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) v[i] = load4<TX>(xr, lane + 64 * i);
  if constexpr (SUMS) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i)
      if (lane + 64 * i < nv) {
        s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) stat[(size_t)row * 16] = make_float2(s1, s2);
  } else {
    float mean, rstd;
    ln_stats(v, lane, nv, c, mean, rstd);
    if (lane == 0) stat[row] = make_float2(rstd, -mean * rstd);
  }
}

// One wave per output feature n:  wf[n,:] = T(w32[n,:] * gamma),  colsum[n] = sum_k wf[n,k] (the
// ROUNDED weights, i.e. what the MFMA multiplies the row mean's contribution by),
// bf[n] = bias[n] + sum_k w32[n,k] * beta[k].
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ w32,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ bias,
                                                      T* __restrict__ wf, float* __restrict__ colsum,
                                                      float* __restrict__ bf, int n_out, int k) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_out) return;
  const float* wr = w32 + (size_t)n * k;
  T* fr = wf + (size_t)n * k;
  float cs = 0.f, bb = 0.f;
  for (int kk = lane; kk < k; kk += 64) {
    const float w = wr[kk];
    const T f = to16<T>(w * gamma[kk]);
    fr[kk] = f;
    cs += to32<T>(f);
    bb += w * beta[kk];
  }
  cs = wave_sum(cs);
  bb = wave_sum(bb);
  if (lane == 0) {
    colsum[n] = cs;
    bf[n] = bias[n] + bb;
  }
}

// x[n*L + t] in place: t == 0 takes cls + pos[0] (patch rows already carry conv + pos from the
// EPI_PATCH GEMM epilogue), then ln_pre.
template <typename TX>
__global__ __launch_bounds__(256) void embed_ln_pre_kernel(TX* __restrict__ x,
                                                           const float* __restrict__ cls,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int rows,
                                                           int L, int c, float2* __restrict__ rowpart) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = c >> 2;
  TX* xr = x + (size_t)row * c;
  const bool is_cls = (row % L) == 0;
  const float4* c4 = reinterpret_cast<const float4*>(cls);
  const float4* p4 = reinterpret_cast<const float4*>(pos);
  float4 v[kMaxVec];
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      if (is_cls) {
        const float4 a = c4[lane + 64 * i], b = p4[lane + 64 * i];
        v[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
      } else {
        v[i] = load4<TX>(xr, lane + 64 * i);
      }
    }
  float mean, rstd;
  ln_stats(v, lane, nv, c, mean, rstd);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      const float4 g = g4[lane + 64 * i], b = b4[lane + 64 * i];
      const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
      const float o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
      store4<TX>(xr, lane + 64 * i, o0, o1, o2, o3);
      s1 += (o0 + o1) + (o2 + o3);
      s2 += (o0 * o0 + o1 * o1) + (o2 * o2 + o3 * o3);
    }
  // layer 0's ln_1 statistics ride along (slot 0 of the GEMM-to-GEMM row-statistics buffer)
  if (rowpart != nullptr) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) rowpart[(size_t)row * 16] = make_float2(s1, s2);
  }
}

// ---- im2col -----------------------------------------------------------------------------
// (measurement switch -DOAKE_IM2COL_NT=1: the input batch is read once — non-temporal loads)
#ifndef OAKE_IM2COL_NT
#define OAKE_IM2COL_NT 0
#endif
template <typename TIN>
__device__ __forceinline__ float load_px(const TIN* p) {
#if OAKE_IM2COL_NT
  return (float)__builtin_nontemporal_load(p);
#else
  return (float)(*p);
#endif
}

template <typename T, typename TIN>
__global__ __launch_bounds__(256) void im2col_kernel(const TIN* __restrict__ img,
                                                     T* __restrict__ out, long total8, int image,
                                                     int patch, int stride, int pad, int grid) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total8) return;
  const int p8 = patch >> 3;               // 8-element chunks per patch row
  const int k8_per_row = 3 * patch * p8;   // chunks per A row
  const long m = idx / k8_per_row;
  const int k8 = idx - m * k8_per_row;
  const int kx0 = (k8 % p8) << 3;
  const int ky = (k8 / p8) % patch;
  const int ch = k8 / (p8 * patch);
  const int g2 = grid * grid;
  const long n = m / g2;
  const int gi = m - n * g2;
  const int gy = gi / grid, gx = gi - gy * grid;
  const int iy = gy * stride - pad + ky;
  const int ix0 = gx * stride - pad + kx0;
  float v[8];
  if (iy < 0 || iy >= image) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  } else {
    const TIN* row = img + (((size_t)n * 3 + ch) * image + iy) * image;
    if (ix0 >= 0 && ix0 + 8 <= image) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = load_px(row + ix0 + j);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ix = ix0 + j;
        v[j] = (ix >= 0 && ix < image) ? load_px(row + ix) : 0.f;
      }
    }
  }
  uint4 o;
  const uint2 lo = pack4<T>(v[0], v[1], v[2], v[3]);
  const uint2 hi = pack4<T>(v[4], v[5], v[6], v[7]);
  o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
  aux_store16(reinterpret_cast<uint4*>(out) + idx, o);
}

// Zero-padded 16-bit copy of an NCHW batch for the conv1 GEMM's patch gather when the convolution pads or its
// stride cuts patches (objects mode: patch 32, stride 16, padding 15): out[n][c][hp][ws] with the image at
// (pad, pad), zeros around it, ws = row stride (a multiple of 8 pixels = 16 bytes).  One thread = 8 output pixels
// = one 16-byte store; the cast of fp32 / the other 16-bit type happens here too.  A third of the bytes of the
// im2col matrix it replaces (201 vs 617 MB per 512 crops at 224^2), and the GEMM re-reads overlapping patches
// from the L2 instead of streaming them from HBM.
template <typename T, typename TIN>
__global__ __launch_bounds__(256) void pad_nchw_kernel(const TIN* __restrict__ img, T* __restrict__ out, long total8,
                                                       int image, int pad, int hp, int ws) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total8) return;
  const int w8 = ws >> 3;
  const long row = idx / w8;            // (n * 3 + c) * hp + y
  const int x0 = (int)(idx - row * w8) << 3;
  const long plane = row / hp;
  const int y = (int)(row - plane * hp) - pad;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (y >= 0 && y < image) {
    const TIN* src = img + ((size_t)plane * image + y) * image;
    const int xs = x0 - pad;
    if (xs >= 0 && xs + 8 <= image) {  // (interior: no per-pixel tests)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = load_px(src + xs + j);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int x = xs + j;
        if (x >= 0 && x < image) v[j] = load_px(src + x);
      }
    }
  }
  uint4 o;
  const uint2 lo = pack4<T>(v[0], v[1], v[2], v[3]);
  const uint2 hi = pack4<T>(v[4], v[5], v[6], v[7]);
  o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
  reinterpret_cast<uint4*>(out)[idx] = o;
}

// the inverse, 16-bit to 16-bit: the dense [n,3,image,image] batch out of the zero-padded one (a pass of a padded-layout
// call that is too small for the persistent conv1 GEMM and takes the im2col route instead; a handful of crops)
__global__ __launch_bounds__(256) void unpad_nchw_kernel(const uint16_t* __restrict__ padded, uint16_t* __restrict__ out,
                                                         long total, int image, int pad, int hp, int ws) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / image;  // (n * 3 + c) * image + y
  const int x = (int)(idx - row * image);
  const long plane = row / image;
  const int y = (int)(row - plane * image);
  out[idx] = padded[((size_t)plane * hp + y + pad) * ws + x + pad];
}

// ---- text tower glue --------------------------------------------------------------------------
// token + positional embedding (clip model.py encode_text: token_embedding(text) + positional_embedding)
template <typename TX>
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ tokens,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos,
                                                         TX* __restrict__ x, int rows, int L, int c,
                                                         int vocab, float2* __restrict__ rowpart) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = c >> 2;
  int tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  const float4* e4 = reinterpret_cast<const float4*>(tok_emb + (size_t)tok * c);
  const float4* p4 = reinterpret_cast<const float4*>(pos + (size_t)(row % L) * c);
  TX* xr = x + (size_t)row * c;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      const float4 a = e4[lane + 64 * i], b = p4[lane + 64 * i];
      const float o0 = a.x + b.x, o1 = a.y + b.y, o2 = a.z + b.z, o3 = a.w + b.w;
      store4<TX>(xr, lane + 64 * i, o0, o1, o2, o3);
      s1 += (o0 + o1) + (o2 + o3);
      s2 += (o0 * o0 + o1 * o1) + (o2 * o2 + o3 * o3);
    }
  if (rowpart != nullptr) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) rowpart[(size_t)row * 16] = make_float2(s1, s2);
  }
}

// one wave per sequence: argmax over the token ids (first maximum, like torch.argmax), then copy that row
template <typename TX>
__global__ __launch_bounds__(256) void gather_eot_kernel(const int32_t* __restrict__ tokens,
                                                         const TX* __restrict__ x, float* __restrict__ y,
                                                         int n, int L, int c) {
  const int lane = threadIdx.x & 63;
  const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seq >= n) return;
  int best = -2147483647 - 1, at = 0x7fffffff;
  for (int t = lane; t < L; t += 64) {
    const int v = tokens[(size_t)seq * L + t];
    if (v > best) {
      best = v;
      at = t;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o, 64), oa = __shfl_xor(at, o, 64);
    if (ob > best || (ob == best && oa < at)) {
      best = ob;
      at = oa;
    }
  }
  const TX* xr = x + ((size_t)seq * L + at) * c;
  float4* yr = reinterpret_cast<float4*>(y + (size_t)seq * c);
  for (int i = lane; i < (c >> 2); i += 64) yr[i] = load4<TX>(xr, i);
}

// ---- head tail: rows of [n, e] fp32 -> optional L2 normalise -> fp16 / fp32 ------------------
// (ln_post is a layernorm_kernel launch over the CLS rows and `@ proj` a small GEMM: the first
// version — one block streaming all of proj per 4 crops — took 85 us cold for 0.2 GFLOP.)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ in,
                                                          void* __restrict__ out, int out_f16,
                                                          int normalize, int n, int e) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int nv = e >> 2;  // e % 4 == 0, e <= 1024
  const float4* ir = reinterpret_cast<const float4*>(in + (size_t)row * e);
  float4 v[kMaxVec];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      v[i] = ir[lane + 64 * i];
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  float scale = 1.0f;
  if (normalize) scale = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);  // F.normalize eps
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i)
    if (lane + 64 * i < nv) {
      const float a = v[i].x * scale, b = v[i].y * scale, c = v[i].z * scale, d = v[i].w * scale;
      if (out_f16)
        reinterpret_cast<uint2*>(reinterpret_cast<f16_t*>(out) + (size_t)row * e)[lane + 64 * i] =
            pack4<f16_t>(a, b, c, d);
      else
        reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * e)[lane + 64 * i] =
            make_float4(a, b, c, d);
    }
}

// ---- casts ----------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, size_t numel,
                            float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < numel) out[i] = to16<T>(in[i] * scale);
}

__global__ void scale_kernel(float* __restrict__ x, size_t numel, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < numel) x[i] *= scale;
}

// ---- crop + ToTensor + Normalize (no resampling: box is out x out) ---------------------------
template <typename TOUT>
__global__ __launch_bounds__(256) void crop_normalize_kernel(const uint8_t* __restrict__ img,
                                                             int height, int width,
                                                             const int32_t* __restrict__ boxes,
                                                             int out_size, float m0, float m1,
                                                             float m2, float s0, float s1, float s2,
                                                             TOUT* __restrict__ out) {
  // grid: (ceil(out*out/256), k)
  const int k = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out_size * out_size) return;
  const int oy = p / out_size, ox = p - oy * out_size;
  const int x1 = boxes[4 * k + 0], y1 = boxes[4 * k + 1];
  const int sx = x1 + ox, sy = y1 + oy;
  float r = 0.f, g = 0.f, b = 0.f;  // PIL crop pads with zeros outside the image
  if (sx >= 0 && sx < width && sy >= 0 && sy < height) {
    const uint8_t* px = img + ((size_t)sy * width + sx) * 3;
    r = (float)px[0];
    g = (float)px[1];
    b = (float)px[2];
  }
  // ToTensor: /255 in fp32 ; Normalize: (x - mean) / std in fp32 — same operation order
  const size_t plane = (size_t)out_size * out_size;
  TOUT* o = out + (size_t)k * 3 * plane + p;
  o[0] = (TOUT)((r / 255.0f - m0) / s0);
  o[plane] = (TOUT)((g / 255.0f - m1) / s1);
  o[2 * plane] = (TOUT)((b / 255.0f - m2) / s2);
}

}  // namespace

hipError_t launch_layernorm(int dtype16, const void* x, int x_dtype, long x_row_stride,
                            const float* gamma, const float* beta, void* y, int rows, int c,
                            hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (c % 4 != 0 || c > kMaxVec * 256) return hipErrorInvalidValue;
  if (x_dtype != DT_F32 && x_dtype != dtype16) return hipErrorInvalidValue;
  const dim3 g((rows + 3) / 4), b(256);
  const bool x32 = x_dtype == DT_F32;
  if (dtype16 == DT_F16) {
    f16_t* yy = reinterpret_cast<f16_t*>(y);
    if (x32)
      hipLaunchKernelGGL((layernorm_kernel<f16_t, float>), g, b, 0, s,
                         reinterpret_cast<const float*>(x), x_row_stride, gamma, beta, yy, rows, c);
    else
      hipLaunchKernelGGL((layernorm_kernel<f16_t, f16_t>), g, b, 0, s,
                         reinterpret_cast<const f16_t*>(x), x_row_stride, gamma, beta, yy, rows, c);
  } else if (dtype16 == DT_BF16) {
    bf16_t* yy = reinterpret_cast<bf16_t*>(y);
    if (x32)
      hipLaunchKernelGGL((layernorm_kernel<bf16_t, float>), g, b, 0, s,
                         reinterpret_cast<const float*>(x), x_row_stride, gamma, beta, yy, rows, c);
    else
      hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), g, b, 0, s,
                         reinterpret_cast<const bf16_t*>(x), x_row_stride, gamma, beta, yy, rows, c);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_rowstat(const void* x, int x_dtype, long x_row_stride, float* stat, int rows, int c,
                          hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (c % 4 != 0 || c > kMaxVec * 256) return hipErrorInvalidValue;
  const dim3 g((rows + 3) / 4), b(256);
  float2* st = reinterpret_cast<float2*>(stat);
  if (x_dtype == DT_F32)
    hipLaunchKernelGGL((rowstat_kernel<float, false>), g, b, 0, s, reinterpret_cast<const float*>(x),
                       x_row_stride, st, rows, c);
  else if (x_dtype == DT_F16)
    hipLaunchKernelGGL((rowstat_kernel<f16_t, false>), g, b, 0, s, reinterpret_cast<const f16_t*>(x),
                       x_row_stride, st, rows, c);
  else if (x_dtype == DT_BF16)
    hipLaunchKernelGGL((rowstat_kernel<bf16_t, false>), g, b, 0, s, reinterpret_cast<const bf16_t*>(x),
                       x_row_stride, st, rows, c);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_rowsums(const void* x, int x_dtype, long x_row_stride, float* rowpart, int rows,
                          int c, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (c % 4 != 0 || c > kMaxVec * 256) return hipErrorInvalidValue;
  const dim3 g((rows + 3) / 4), b(256);
  float2* st = reinterpret_cast<float2*>(rowpart);
  if (x_dtype == DT_F32)
    hipLaunchKernelGGL((rowstat_kernel<float, true>), g, b, 0, s, reinterpret_cast<const float*>(x),
                       x_row_stride, st, rows, c);
  else if (x_dtype == DT_F16)
    hipLaunchKernelGGL((rowstat_kernel<f16_t, true>), g, b, 0, s, reinterpret_cast<const f16_t*>(x),
                       x_row_stride, st, rows, c);
  else if (x_dtype == DT_BF16)
    hipLaunchKernelGGL((rowstat_kernel<bf16_t, true>), g, b, 0, s,
                       reinterpret_cast<const bf16_t*>(x), x_row_stride, st, rows, c);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_fold_ln(int dtype16, const float* w32, const float* gamma, const float* beta,
                          const float* bias, void* wf, float* colsum, float* bf, int n_out, int k,
                          hipStream_t s) {
  if (n_out <= 0) return hipSuccess;
  const dim3 g((n_out + 3) / 4), b(256);
  if (dtype16 == DT_F16)
    hipLaunchKernelGGL((fold_ln_kernel<f16_t>), g, b, 0, s, w32, gamma, beta, bias,
                       reinterpret_cast<f16_t*>(wf), colsum, bf, n_out, k);
  else if (dtype16 == DT_BF16)
    hipLaunchKernelGGL((fold_ln_kernel<bf16_t>), g, b, 0, s, w32, gamma, beta, bias,
                       reinterpret_cast<bf16_t*>(wf), colsum, bf, n_out, k);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_text_embed(const int32_t* tokens, const float* tok_emb, const float* pos, void* x,
                             int x_dtype, int n, int L, int c, int vocab, float* rowpart, hipStream_t s) {
  const int rows = n * L;
  if (rows <= 0) return hipSuccess;
  if (c % 4 != 0 || c > kMaxVec * 256) return hipErrorInvalidValue;
  const dim3 g((rows + 3) / 4), b(256);
  float2* rp = reinterpret_cast<float2*>(rowpart);
  if (x_dtype == DT_F32)
    hipLaunchKernelGGL(text_embed_kernel<float>, g, b, 0, s, tokens, tok_emb, pos,
                       reinterpret_cast<float*>(x), rows, L, c, vocab, rp);
  else if (x_dtype == DT_F16)
    hipLaunchKernelGGL(text_embed_kernel<f16_t>, g, b, 0, s, tokens, tok_emb, pos,
                       reinterpret_cast<f16_t*>(x), rows, L, c, vocab, rp);
  else if (x_dtype == DT_BF16)
    hipLaunchKernelGGL(text_embed_kernel<bf16_t>, g, b, 0, s, tokens, tok_emb, pos,
                       reinterpret_cast<bf16_t*>(x), rows, L, c, vocab, rp);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_gather_eot(const int32_t* tokens, const void* x, int x_dtype, float* y, int n, int L,
                             int c, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (c % 4 != 0) return hipErrorInvalidValue;
  const dim3 g((n + 3) / 4), b(256);
  if (x_dtype == DT_F32)
    hipLaunchKernelGGL(gather_eot_kernel<float>, g, b, 0, s, tokens, reinterpret_cast<const float*>(x), y, n, L, c);
  else if (x_dtype == DT_F16)
    hipLaunchKernelGGL(gather_eot_kernel<f16_t>, g, b, 0, s, tokens, reinterpret_cast<const f16_t*>(x), y, n, L, c);
  else if (x_dtype == DT_BF16)
    hipLaunchKernelGGL(gather_eot_kernel<bf16_t>, g, b, 0, s, tokens, reinterpret_cast<const bf16_t*>(x), y, n, L, c);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_embed_ln_pre(void* x, int x_dtype, const float* cls, const float* pos,
                               const float* gamma, const float* beta, int n, int L, int c,
                               float* rowpart, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (c % 4 != 0 || c > kMaxVec * 256) return hipErrorInvalidValue;
  const int rows = n * L;
  const dim3 g((rows + 3) / 4), b(256);
  if (x_dtype == DT_F32)
    hipLaunchKernelGGL(embed_ln_pre_kernel<float>, g, b, 0, s, reinterpret_cast<float*>(x), cls, pos,
                       gamma, beta, rows, L, c, reinterpret_cast<float2*>(rowpart));
  else if (x_dtype == DT_F16)
    hipLaunchKernelGGL(embed_ln_pre_kernel<f16_t>, g, b, 0, s, reinterpret_cast<f16_t*>(x), cls, pos,
                       gamma, beta, rows, L, c, reinterpret_cast<float2*>(rowpart));
  else if (x_dtype == DT_BF16)
    hipLaunchKernelGGL(embed_ln_pre_kernel<bf16_t>, g, b, 0, s, reinterpret_cast<bf16_t*>(x), cls,
                       pos, gamma, beta, rows, L, c, reinterpret_cast<float2*>(rowpart));
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

template <typename T>
static hipError_t im2col_in(const void* img, int in_dtype, void* out, long total8, int image,
                            int patch, int stride, int pad, int grid, hipStream_t s) {
  const dim3 g((total8 + 255) / 256), b(256);
  T* o = reinterpret_cast<T*>(out);
  switch (in_dtype) {
    case DT_F32:
      hipLaunchKernelGGL((im2col_kernel<T, float>), g, b, 0, s,
                         reinterpret_cast<const float*>(img), o, total8, image, patch, stride, pad,
                         grid);
      break;
    case DT_F16:
      hipLaunchKernelGGL((im2col_kernel<T, f16_t>), g, b, 0, s,
                         reinterpret_cast<const f16_t*>(img), o, total8, image, patch, stride, pad,
                         grid);
      break;
    case DT_BF16:
      hipLaunchKernelGGL((im2col_kernel<T, bf16_t>), g, b, 0, s,
                         reinterpret_cast<const bf16_t*>(img), o, total8, image, patch, stride, pad,
                         grid);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_im2col(int dtype16, const void* img, int in_dtype, void* out, int n, int image,
                         int patch, int stride, int pad, int grid, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (patch % 8 != 0) return hipErrorInvalidValue;
  const long total8 = (long)n * grid * grid * 3 * patch * (patch / 8);
  if (dtype16 == DT_F16)
    return im2col_in<f16_t>(img, in_dtype, out, total8, image, patch, stride, pad, grid, s);
  if (dtype16 == DT_BF16)
    return im2col_in<bf16_t>(img, in_dtype, out, total8, image, patch, stride, pad, grid, s);
  return hipErrorInvalidValue;
}

template <typename T>
static hipError_t pad_in(const void* img, int in_dtype, void* out, long total8, int image, int pad, int hp, int ws,
                         hipStream_t s) {
  const dim3 g((total8 + 255) / 256), b(256);
  T* o = reinterpret_cast<T*>(out);
  switch (in_dtype) {
    case DT_F32:
      hipLaunchKernelGGL((pad_nchw_kernel<T, float>), g, b, 0, s, reinterpret_cast<const float*>(img), o, total8,
                         image, pad, hp, ws);
      break;
    case DT_F16:
      hipLaunchKernelGGL((pad_nchw_kernel<T, f16_t>), g, b, 0, s, reinterpret_cast<const f16_t*>(img), o, total8,
                         image, pad, hp, ws);
      break;
    case DT_BF16:
      hipLaunchKernelGGL((pad_nchw_kernel<T, bf16_t>), g, b, 0, s, reinterpret_cast<const bf16_t*>(img), o, total8,
                         image, pad, hp, ws);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_pad_nchw(int dtype16, const void* img, int in_dtype, void* out, int n, int image, int pad, int hp,
                           int ws, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (ws % 8 != 0 || hp < image + pad || ws < image + pad) return hipErrorInvalidValue;
  const long total8 = (long)n * 3 * hp * (ws / 8);
  if (dtype16 == DT_F16) return pad_in<f16_t>(img, in_dtype, out, total8, image, pad, hp, ws, s);
  if (dtype16 == DT_BF16) return pad_in<bf16_t>(img, in_dtype, out, total8, image, pad, hp, ws, s);
  return hipErrorInvalidValue;
}

hipError_t launch_unpad_nchw(const void* padded, void* out, int n, int image, int pad, int hp, int ws, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (hp < image + pad || ws < image + pad) return hipErrorInvalidValue;
  const long total = (long)n * 3 * image * image;
  hipLaunchKernelGGL(unpad_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const uint16_t*>(padded), reinterpret_cast<uint16_t*>(out), total, image, pad, hp, ws);
  return hipGetLastError();
}

hipError_t launch_l2norm_rows(const float* in, void* out, int out_dtype, int normalize, int n, int e,
                              hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (e % 4 != 0 || e > kMaxVec * 256) return hipErrorInvalidValue;
  if (out_dtype != DT_F32 && out_dtype != DT_F16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, in, out,
                     out_dtype == DT_F16 ? 1 : 0, normalize, n, e);
  return hipGetLastError();
}

hipError_t launch_cast_f32_to_16(int dtype16, const float* in, void* out, size_t numel, float scale,
                                 hipStream_t s) {
  if (numel == 0) return hipSuccess;
  const dim3 g((numel + 255) / 256), b(256);
  if (dtype16 == DT_F16)
    hipLaunchKernelGGL(cast_kernel<f16_t>, g, b, 0, s, in, reinterpret_cast<f16_t*>(out), numel,
                       scale);
  else if (dtype16 == DT_BF16)
    hipLaunchKernelGGL(cast_kernel<bf16_t>, g, b, 0, s, in, reinterpret_cast<bf16_t*>(out), numel,
                       scale);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_scale_f32(float* x, size_t numel, float scale, hipStream_t s) {
  if (numel == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_kernel, dim3((numel + 255) / 256), dim3(256), 0, s, x, numel, scale);
  return hipGetLastError();
}

hipError_t launch_crop_normalize(const uint8_t* img, int height, int width, const int32_t* boxes,
                                 int k, int out_size, const float* mean3, const float* std3,
                                 void* out, int out_dtype, hipStream_t s) {
  if (k <= 0) return hipSuccess;
  const dim3 g((out_size * out_size + 255) / 256, k), b(256);
  if (out_dtype == DT_F32)
    hipLaunchKernelGGL(crop_normalize_kernel<float>, g, b, 0, s, img, height, width, boxes, out_size,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                       reinterpret_cast<float*>(out));
  else if (out_dtype == DT_F16)
    hipLaunchKernelGGL(crop_normalize_kernel<f16_t>, g, b, 0, s, img, height, width, boxes, out_size,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                       reinterpret_cast<f16_t*>(out));
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace oake
