// Micro-benchmark: how fast can a CU issue the tile-end burst of 16-B stores of a 160x256 16-bit tile,
// as a function of how a wave's 64 lanes are laid over the rows?  (GPU box only.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bench tools/ubench/store_bench.hip && /tmp/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// pattern 0: MFMA-accumulator layout — lane (r = l & 15, g = l >> 4) writes 16 B at row r, byte 16 g
//            (+ 64 B per second piece): an instruction covers 16 rows x 64 B
// pattern 1: row swap layout — 8 rows x 128 B per instruction
// pattern 2: 2 rows x 512 B per instruction (what an LDS-transposed tile could store)
// pattern 3: as 0 but every wave writes dwordx2 (twice the instructions)
template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(char* out, int ld_bytes, int waves_per_tile_row,
                                                    long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // block b owns tile rows [160 b', +160) x 512 B at column block (b % 3) of a 12800 x 768 f16 matrix
  const int tm = blockIdx.x / 3, tn = blockIdx.x % 3;
  char* tile = out + (size_t)tm * 160 * ld_bytes + tn * 512;
  const int wm = wave / 4, wn = wave % 4;  // 2 x 4 waves of 80 x 64 columns (128 B)
  char* wt = tile + (size_t)wm * 80 * ld_bytes + wn * 128;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (PATTERN == 0) {
#pragma unroll
    for (int mi = 0; mi < 5; ++mi)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<uint4*>(wt + (size_t)(mi * 16 + (lane & 15)) * ld_bytes + t * 64 + (lane >> 4) * 16) = v;
  } else if (PATTERN == 1) {
#pragma unroll
    for (int mi = 0; mi < 5; ++mi)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<uint4*>(wt + (size_t)(mi * 16 + (lane >> 3) * 2 + t) * ld_bytes + (lane & 7) * 16) = v;
  } else if (PATTERN == 2) {
    // the block's 160 rows x 512 B as 80 KiB: wave w stores rows [20 w, +20), 2 rows per instruction
#pragma unroll
    for (int i = 0; i < 10; ++i)
      *reinterpret_cast<uint4*>(tile + (size_t)(wave * 20 + i * 2 + (lane >> 5)) * ld_bytes + (lane & 31) * 16) = v;
  } else {
#pragma unroll
    for (int mi = 0; mi < 5; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<uint2*>(wt + (size_t)(mi * 16 + (lane & 15)) * ld_bytes + t * 32 + (lane >> 4) * 8) =
            make_uint2(v.x, v.y);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): until the last store is acknowledged
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int P>
void run(const char* name, char* out, long long* cyc, int blocks) {
  std::vector<long long> h(240 * 8);
  float best_ms = 1e9f;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(store_kernel<P>, dim3(blocks), dim3(512), 0, 0, out, 768 * 2, 4, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best_ms = ms < best_ms ? ms : best_ms;
  }
  hipMemcpy(h.data(), cyc, blocks * 8 * sizeof(long long), hipMemcpyDeviceToHost);
  long long mx = 0, sum = 0;
  for (int i = 0; i < blocks * 8; ++i) { mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
  printf("%-34s blocks %3d: slowest wave %6lld cycles, mean %6lld; 80 KiB per CU -> %.1f B/cycle/CU; kernel %.1f us\n",
         name, blocks, mx, sum / (blocks * 8), 81920.0 / mx, best_ms * 1e3);
}

int main() {
  char* out; long long* cyc;
  hipMalloc(&out, (size_t)12800 * 768 * 2);
  hipMalloc(&cyc, 240 * 8 * sizeof(long long));
  for (int blocks : {240, 30}) {
    run<0>("16 rows x 64 B per store (dwordx4)", out, cyc, blocks);
    run<1>("8 rows x 128 B per store (dwordx4)", out, cyc, blocks);
    run<2>("2 rows x 512 B per store (dwordx4)", out, cyc, blocks);
    run<3>("16 rows x 32 B per store (dwordx2)", out, cyc, blocks);
  }
  return 0;
}
