// Micro-benchmark for DESIGN §9 1(a): ONE compute wave per SIMD running a 160x64 wave tile (10 x 4
// accumulator tiles, 160 registers) straight through a K-tile — 80 v_mfma_f32_16x16x32_f16 with the
// fragment reads (ds_read_b128 from a swizzled 128-B-row LDS tile, as gemm.hip) software-pipelined under
// them — against the production layout's measured 1586 cycles per K-tile (two waves per SIMD in
// alternating load / MFMA phases).  No DMA, no barriers: this is the upper bound of such a K-loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_stream tools/ubench/mfma_stream.hip && /tmp/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 160, BN = 256, ROW = 128;  // bytes per K-tile row (64 halves)

template <int PIPE>
__global__ __launch_bounds__(256) void stream_kernel(long long* cycles, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // A tile then B tile, twice (2 stages)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * (BM + BN) * ROW / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i % 7;
  __syncthreads();
  const int fr = lane & 15, g = lane >> 4, fsw = (fr >> 1) & 7;
  f32x4 acc[10][4];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* st = smem + (it & 1) * (BM + BN) * ROW;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int koff = ((kk * 4 + g) ^ fsw) << 4;
      const char* a0 = st + fr * ROW + koff;
      const char* b0 = st + BM * ROW + (wid * 64 + fr) * ROW + koff;
      f16x8 bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(b0 + j * 16 * ROW);
      if (PIPE) {
        f16x8 af = *reinterpret_cast<const f16x8*>(a0);
#pragma unroll
        for (int mi = 0; mi < 10; ++mi) {
          const f16x8 cur = af;
          if (mi + 1 < 10) af = *reinterpret_cast<const f16x8*>(a0 + (mi + 1) * 16 * ROW);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], cur, acc[mi][j], 0, 0, 0);
        }
      } else {
        f16x8 af[10];
#pragma unroll
        for (int mi = 0; mi < 10; ++mi) af[mi] = *reinterpret_cast<const f16x8*>(a0 + mi * 16 * ROW);
#pragma unroll
        for (int mi = 0; mi < 10; ++mi)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[mi], acc[mi][j], 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.f) sink[0] = s;
  if (lane == 0) cycles[blockIdx.x * 4 + wid] = t1 - t0;
}

template <int PIPE>
void run(const char* name, long long* cyc, float* sink, int blocks) {
  const int iters = 2000, lds = 2 * (BM + BN) * ROW;
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL(stream_kernel<PIPE>, dim3(blocks), dim3(256), lds, 0, cyc, sink, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-44s %4d blocks: %.0f cycles per K-tile (80 MFMAs per SIMD): %.1f cycles per MFMA\n", name, blocks,
         sum / h.size() / iters, sum / h.size() / iters / 80);
}

int main() {
  long long* cyc; float* sink;
  hipMalloc(&cyc, 256 * 4 * sizeof(long long)); hipMalloc(&sink, 4);
  for (int blocks : {1, 256}) {
    run<0>("one wave/SIMD, all 14 fragment reads then MFMAs", cyc, sink, blocks);
    run<1>("one wave/SIMD, A fragments prefetched one ahead", cyc, sink, blocks);
  }
  return 0;
}
