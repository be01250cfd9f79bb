// Micro-benchmark for docs/history/round1_notes.md 1(a): ONE compute wave per SIMD running a 160x64 wave tile (10 x 4
// accumulator tiles, 160 registers) straight through a K-tile — 80 v_mfma_f32_16x16x32_f16 with the
// fragment reads (ds_read_b128 from a swizzled 128-B-row LDS tile, as gemm.hip) software-pipelined under
// them — against the production layout's measured 1586 cycles per K-tile (two waves per SIMD in
// alternating load / MFMA phases).  No DMA, no barriers.  RESULTS: by default hipcc keeps the 160
// accumulator registers in AGPRs and moves them around every iteration (26-31 cycles per MFMA of
// compiler-made traffic); built with -mllvm -amdgpu-mfma-vgpr-form=1: MFMAs only 16.3 cycles per MFMA,
// unpipelined reads 23.2, reads pipelined over the kk halves 17.7 (1416 cycles per K-tile).
// bare_kernel (5 x 4 tiles, in place either way): 17.2.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o /tmp/mfma_stream tools/ubench/mfma_stream.hip
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
  if (PIPE == 3) {  // no fragment reads at all: the bare MFMA issue rate of one wave
    f16x8 bf[4], af[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(smem + (fr + 16 * j) * ROW + g * 16);
    af[0] = *reinterpret_cast<const f16x8*>(smem + fr * ROW + 64 + g * 16);
    af[1] = *reinterpret_cast<const f16x8*>(smem + (fr + 16) * ROW + 64 + g * 16);
    for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
      for (int mi = 0; mi < 10; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[mi & 1], acc[mi][j], 0, 0, 0);
    }
  } else if (PIPE == 2) {
    // software pipeline over "halves" (kk of a K-tile): while the 40 MFMAs of half h run, the A
    // fragments of h are fetched one row block ahead and the 4 B fragments + first A fragment of
    // half h + 1 are fetched; sched_group_barrier pins {DS reads, 4 MFMAs} groups in that order
    auto a_ptr = [&](int h) { return smem + ((h >> 1) & 1) * (BM + BN) * ROW + fr * ROW + ((((h & 1) * 4 + g) ^ fsw) << 4); };
    auto b_ptr = [&](int h) { return smem + ((h >> 1) & 1) * (BM + BN) * ROW + BM * ROW + (wid * 64 + fr) * ROW + ((((h & 1) * 4 + g) ^ fsw) << 4); };
    f16x8 bf[2][4];
    f16x8 af[5];  // row blocks mi, mi + 1, mi + 2 (+ slack: 10 row blocks per half = 2 x 5 slots)
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(b_ptr(0) + j * 16 * ROW);
    af[0] = *reinterpret_cast<const f16x8*>(a_ptr(0));
    af[1] = *reinterpret_cast<const f16x8*>(a_ptr(0) + 16 * ROW);
    for (int h2 = 0; h2 < 2 * iters; h2 += 2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int h = h2 + hh;
        const char* a0 = a_ptr(h);
        const char* an = a_ptr(h + 1);
        const char* bn = b_ptr(h + 1);
#pragma unroll
        for (int mi = 0; mi < 10; ++mi) {
          const int cur = mi % 5, nxt = (mi + 2) % 5;
          // A fragment two row blocks ahead: of this half, or row block 0 / 1 of the next half
          af[nxt] = *reinterpret_cast<const f16x8*>(mi + 2 < 10 ? a0 + (mi + 2) * 16 * ROW : an + (mi + 2 - 10) * 16 * ROW);
          if (mi >= 2 && mi < 6) bf[hh ^ 1][mi - 2] = *reinterpret_cast<const f16x8*>(bn + (mi - 2) * 16 * ROW);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[hh][j], af[cur], acc[mi][j], 0, 0, 0);
          if (mi >= 2 && mi < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
      }
    }
  } else
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
      if (PIPE == 2) {
        // (handled by the fully pipelined loop below)
      } else if (PIPE) {
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

// The bare issue rate of ONE wave per SIMD on the production wave tile (80x64: 5 x 4 accumulator
// tiles, compiled like gemm_pp_kernel for 3 waves per SIMD so the accumulators stay in place in VGPRs —
// with the 160-register tile above hipcc parks accumulators in AGPRs and shuffles them every iteration,
// which is what the 26 / 31 cycles per MFMA of those variants measure).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void bare_kernel(long long* cycles, float* sink,
                                                                                               int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i % 7;
  __syncthreads();
  const int fr = lane & 15, g = lane >> 4;
  f16x8 bf[4], af[5];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8*>(smem + (fr + 16 * j) * ROW + g * 16);
#pragma unroll
  for (int i = 0; i < 5; ++i) af[i] = *reinterpret_cast<const f16x8*>(smem + (fr + 16 * i) * ROW + 64 + g * 16);
  f32x4 acc[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 4 * iters; ++it) {
#pragma unroll
    for (int mi = 0; mi < 5; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[mi], acc[mi][j], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
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
    run<2>("one wave/SIMD, pipelined over halves + sched groups", cyc, sink, blocks);
    run<3>("one wave/SIMD, MFMAs only (no fragment reads)", cyc, sink, blocks);
  }
  {
    const int iters = 2000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bare_kernel, dim3(blocks), dim3(256), 16384, 0, cyc, sink, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    printf("bare MFMA stream, one wave/SIMD, 5x4 accumulator tiles in VGPRs: %.1f cycles per MFMA\n",
           sum / h.size() / (4.0 * iters) / 20);
  }
  return 0;
}
