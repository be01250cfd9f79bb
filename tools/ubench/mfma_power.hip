// Micro-benchmark: the MFMA rate the board sustains under its power cap as a function of the operand DATA.
// Every SIMD of every CU runs two waves of back-to-back v_mfma_f32_16x16x32_f16 on a 5 x 4 block of
// accumulator tiles (the production wave tile) from registers only — no LDS, no memory — with operand
// fragments that are (0) all zero, (1) all 1.0, (2) N(0, 0.5^2) random halves (what the encoder multiplies).
// Same instruction stream in all three; only the bits differ.  Runs ~2 s per mode so clocks settle.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o /tmp/mfma_power tools/ubench/mfma_power.hip
//   /tmp/mfma_power [seconds per mode = 2]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ORDER 0: consecutive MFMAs change the FIRST operand (the W fragment in gemm.hip) and keep the second for four
// instructions (the production loop order); ORDER 1: they keep the first operand for five instructions and
// change the second.  Same instructions, same data: does the operand that stays matter for power?
template <int ORDER>
__global__ __launch_bounds__(512) void mfma_kernel(const f16x8* __restrict__ frags, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 af[5], bf[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) af[i] = frags[i * 64 + lane];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = frags[(5 + j) * 64 + lane];
  f32x4 acc[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) sink[0] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same FLOPs per iteration from v_mfma_f32_32x32x16_f16: 10 instructions on a 160 x 64 strip... here a
// (5 x 32) x (2 x 32) block of accumulator tiles = 160 accumulator registers, 5 + 2 operand fragments
__global__ __launch_bounds__(512) void mfma32_kernel(const f16x8* __restrict__ frags, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 af[5], bf[2];
#pragma unroll
  for (int i = 0; i < 5; ++i) af[i] = frags[i * 64 + lane];
#pragma unroll
  for (int j = 0; j < 2; ++j) bf[j] = frags[(5 + j) * 64 + lane];
  f32x16 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  f16x8* d_frags;
  float* d_sink;
  hipMalloc(&d_frags, 9 * 64 * sizeof(f16x8));
  hipMalloc(&d_sink, 4);
  const char* names[3] = {"zeros", "ones", "random N(0,0.25)"};
  for (int rnd = 0; rnd < 2; ++rnd)
  for (int shape = 0; shape < 3; ++shape)
  for (int mode = 0; mode < 3; ++mode) {
    if (shape >= 1 && mode == 1) continue;
    std::vector<_Float16> h(9 * 64 * 8);
    srand(7);
    for (auto& v : h) {
      if (mode == 0) v = (_Float16)0.f;
      else if (mode == 1) v = (_Float16)1.f;
      else {
        const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
        v = (_Float16)(0.5 * sqrt(-2.0 * log(u1)) * cos(6.283185307 * u2));
      }
    }
    hipMemcpy(d_frags, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 20000;  // 20 MFMAs x 16384 flop x iters per wave
    const double flop_per_launch = (double)cus * 8 * 20 * 16384.0 * iters;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
      if (shape == 0) mfma_kernel<0><<<cus, 512>>>(d_frags, d_sink, iters);
      else if (shape == 2) mfma_kernel<1><<<cus, 512>>>(d_frags, d_sink, iters);
      else mfma32_kernel<<<cus, 512>>>(d_frags, d_sink, iters);
    };
    launch();  // warm
    hipDeviceSynchronize();
    // launches back to back until `seconds` have passed; report the rate of the LAST half
    int n = 0;
    double total_ms = 0, last_ms = 0;
    int last_n = 0;
    while (total_ms < seconds * 1e3) {
      hipEventRecord(e0);
      for (int k = 0; k < 10; ++k) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      total_ms += ms; n += 10;
      if (total_ms > seconds * 500) { last_ms += ms; last_n += 10; }
    }
    printf("%s %-18s %7.0f TFLOP/s over the last %.2f s (%d launches of %.2f ms)\n", shape == 1 ? "32x32x16" : (shape == 2 ? "16x16x32 first operand kept" : "16x16x32"), names[mode],
           flop_per_launch * last_n / (last_ms * 1e-3) / 1e12, last_ms * 1e-3, last_n, last_ms / last_n);
    fflush(stdout);
  }
  return 0;
}
