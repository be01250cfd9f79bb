// gemm.hip — C[M,N] = A[M,K] * W[N,K]^T on the gfx950 matrix cores (v_mfma_f32_16x16x32_{f16,bf16}).
//
// This one kernel carries 98.9 % of encode_image's FLOPs (SURVEY.md §8 A15a,d,f,g): conv1 as an
// im2col GEMM, QKV in-proj, attention out-proj, MLP c_fc (+QuickGELU) and c_proj (+residual).
// Both operands are K-contiguous ("B^T input"), which is exactly PyTorch's nn.Linear / Conv2d
// weight layout, so no weight transposition is needed at load time.
//
// Structure (wave64, 4 waves = 2x2, each wave a 64x64 sub-tile = 4x4 MFMA tiles of 16x16):
//   * 128x128x64 block tile, register-staged global->LDS double buffer, one barrier per K-tile
//     (loads of tile t+1 are issued before the MFMAs of tile t, written to the other buffer after).
//   * LDS image: [rows][64] 16-bit = 128 B per row; 16-B chunk index XOR-swizzled with
//     (row>>1)&7, which makes both the ds_write_b128 staging writes and the ds_read_b128
//     fragment reads bank-conflict free (a 256-B bank row holds two tile rows).
//   * MFMA operands are swapped (W fragment as the A operand, activation fragment as the B operand)
//     so each lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row:
//     the epilogue stores 8 B (16-bit out) or 16 B (fp32 out) per lane instead of 2-4 B.
//   * blockIdx -> tile mapping is XCD-aware (common.h: xcd_remap).
#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 64;
constexpr int kRowBytes = BK * 2;                    // 128
constexpr int kATileBytes = BM * kRowBytes;          // 16384
constexpr int kStageBytes = (BM + BN) * kRowBytes;   // 32768
constexpr int kGemmLds = 2 * kStageBytes;            // 65536

struct EpiParams {
  const float* bias;
  void* out;
  int ldo;
  const float* pos;
  int P2;
  int L;
};

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const T* __restrict__ A,
                                                      const T* __restrict__ W, int M, int N, int K,
                                                      EpiParams ep, int tiles_n, int nwg) {
  typedef typename T16<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int logical = xcd_remap(blockIdx.x, nwg);
  const int tm = logical / tiles_n, tn = logical % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: thread t moves 16-B chunk (row = t/8 + 32 i, chunk = t%8), i = 0..3 ----
  const int lrow = tid >> 3;
  const int lch = tid & 7;
  const T* ap[4];
  const T* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + lrow + 32 * i;
    r = r < M ? r : M - 1;
    ap[i] = A + (size_t)r * K + lch * 8;
    int c = n0 + lrow + 32 * i;
    c = c < N ? c : N - 1;
    wp[i] = W + (size_t)c * K + lch * 8;
  }
  const int wr_off = lrow * kRowBytes + ((lch ^ ((lrow >> 1) & 7)) << 4);

  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const uint4*>(ap[i] + (size_t)kt * BK);
      rb[i] = *reinterpret_cast<const uint4*>(wp[i] + (size_t)kt * BK);
    }
  };
  auto lwrite = [&](int stage) {
    char* base = smem + stage * kStageBytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(base + wr_off + i * 32 * kRowBytes) = ra[i];
      *reinterpret_cast<uint4*>(base + kATileBytes + wr_off + i * 32 * kRowBytes) = rb[i];
    }
  };

  // ---- fragment reads: lane -> (row = lane&15, k-chunk = lane>>4) ----
  const int frow = lane & 15;
  const int fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int a_base = (wm * 64 + frow) * kRowBytes;
  const int b_base = kATileBytes + (wn * 64 + frow) * kRowBytes;
  const int koff0 = ((0 * 4 + fg) ^ fsw) << 4;
  const int koff1 = ((1 * 4 + fg) ^ fsw) << 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  gload(0);
  lwrite(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* st = smem + cur * kStageBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int koff = kk == 0 ? koff0 : koff1;
      vec8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff);
        bf[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = T16<T>::mfma(bf[ni], af[mi], acc[mi][ni]);
    }
    if (kt + 1 < nk) lwrite(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+lane&15][n = ..+4*(lane>>4) + 0..3] ----
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + frow;
    if (m >= M) continue;
    size_t orow;
    const float* posrow = nullptr;
    if (EPI == EPI_PATCH) {
      const int img = m / ep.P2;
      const int p = m - img * ep.P2;
      orow = (size_t)img * ep.L + 1 + p;
      posrow = ep.pos + (size_t)(1 + p) * N;
    } else {
      orow = (size_t)m;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + 4 * fg;
      if (n >= N) continue;
      f32x4 v = acc[mi][ni];
      if (EPI == EPI_PATCH) {
        const float4 pv = *reinterpret_cast<const float4*>(posrow + n);
        v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
      } else if (ep.bias != nullptr) {
        const float4 bv = *reinterpret_cast<const float4*>(ep.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (EPI == EPI_F32_BIAS || EPI == EPI_PATCH) {
        float* o = reinterpret_cast<float*>(ep.out) + orow * ep.ldo + n;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_RESID) {
        float* o = reinterpret_cast<float*>(ep.out) + orow * ep.ldo + n;
        float4 r = *reinterpret_cast<const float4*>(o);
        r.x += v[0]; r.y += v[1]; r.z += v[2]; r.w += v[3];
        *reinterpret_cast<float4*>(o) = r;
      } else if (EPI == EPI_T16_BIAS) {
        T* o = reinterpret_cast<T*>(ep.out) + orow * ep.ldo + n;
        *reinterpret_cast<uint2*>(o) = pack4<T>(v[0], v[1], v[2], v[3]);
      } else {  // EPI_T16_GELU
        T* o = reinterpret_cast<T*>(ep.out) + orow * ep.ldo + n;
        *reinterpret_cast<uint2*>(o) =
            pack4<T>(quick_gelu(v[0]), quick_gelu(v[1]), quick_gelu(v[2]), quick_gelu(v[3]));
      }
    }
  }
}

template <typename T, int EPI>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gemm_kernel<T, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles_m = (a.M + BM - 1) / BM;
  const int tiles_n = (a.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L};
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), kGemmLds, s, reinterpret_cast<const T*>(a.A),
                     reinterpret_cast<const T*>(a.W), a.M, a.N, a.K, ep, tiles_n, nwg);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_epi(int epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_F32_BIAS: return launch_t<T, EPI_F32_BIAS>(a, s);
    case EPI_T16_BIAS: return launch_t<T, EPI_T16_BIAS>(a, s);
    case EPI_T16_GELU: return launch_t<T, EPI_T16_GELU>(a, s);
    case EPI_RESID: return launch_t<T, EPI_RESID>(a, s);
    case EPI_PATCH: return launch_t<T, EPI_PATCH>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_gemm(int dtype16, int epi, const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
  if (a.K % BK != 0 || a.N % 4 != 0 || a.ldo % 4 != 0) return hipErrorInvalidValue;
  if (dtype16 == DT_F16) return launch_epi<f16_t>(epi, a, s);
  if (dtype16 == DT_BF16) return launch_epi<bf16_t>(epi, a, s);
  return hipErrorInvalidValue;
}

}  // namespace oake
