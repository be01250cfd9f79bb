// common.h — shared device helpers for the OAKE gfx950 kernels.
// Wavefront = 64 lanes everywhere; MFMA operands are 16-bit (f16 or bf16), accumulation fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <atomic>

namespace oake {

// Kernel-level timing for the profiler (api.hip, RUNK): when the two events are set, a launch made
// through OAKE_LAUNCH stamps them with the kernel's own begin / end (hipExtLaunchKernelGGL) — the
// same interval rocprofv3's kernel trace reports.  Bracketing a launch with hipEventRecord instead
// includes ~5 us of dispatch latency per kernel.  Null events: a plain launch.  (Per host thread: set
// around one launch by the thread that makes it.)
extern thread_local hipEvent_t g_launch_start, g_launch_stop;
#define OAKE_LAUNCH(kern, grid, block, lds, stream, ...)                                         \
  hipExtLaunchKernelGGL(kern, grid, block, lds, stream, oake::g_launch_start, oake::g_launch_stop, 0, \
                        __VA_ARGS__)


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device and
// handles may be driven from several host threads, so the "done" flag is an atomic bit per device (setting it
// twice is harmless; devices beyond 63 set it on every launch).
struct DynLdsAttr {
  std::atomic<uint64_t> done{0};
  hipError_t ensure(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = (dev >= 0 && dev < 64) ? 1ull << dev : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// compute units of the current device (cached per device; persistent kernels launch one block per CU)
inline hipError_t device_cu_count(int* cus) {
  static std::atomic<int> cache[64];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) { *cus = c; return hipSuccess; }
  }
  int n = 0;
  e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64) cache[dev].store(n, std::memory_order_relaxed);
  *cus = n;
  return hipSuccess;
}

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;
constexpr int kHeadDim = 64;  // ViT-B/32, B/16, L/14 all use 64

// ---- 16-bit operand traits -------------------------------------------------------------
template <typename T>
struct T16;

template <>
struct T16<f16_t> {
  typedef f16x8 vec8;
  typedef f16x4 vec4;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <>
struct T16<bf16_t> {
  typedef bf16x8 vec8;
  typedef bf16x4 vec4;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ T to16(float x) {
  return (T)x;  // round-to-nearest-even for both _Float16 and __bf16
}
template <typename T>
__device__ __forceinline__ float to32(T x) {
  return (float)x;
}

// 4 floats -> 4 x 16-bit packed into 8 bytes
template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  typename T16<T>::vec4 v;
  v[0] = to16<T>(a);
  v[1] = to16<T>(b);
  v[2] = to16<T>(c);
  v[3] = to16<T>(d);
  return __builtin_bit_cast(uint2, v);
}

// ---- wave reductions --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Reductions over lanes l, l^16, l^32, l^48 — the four 16-lane rows of a wave, i.e. the four lanes
// that hold one row (column) of a 16x16 MFMA accumulator tile — with the gfx950 row swaps:
// v_permlane16_swap / v_permlane32_swap are plain VALU instructions, where __shfl_xor compiles to
// ds_bpermute_b32, a round trip through the LDS crossbar with an lgkmcnt wait per butterfly level.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows16_sum(float v) {
  u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rows16_max(float v) {
  u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Full-line global accesses from MFMA-fragment layouts.  In a 16x16x32 operand / accumulator-pair
// layout lane (r = l & 15, g = l >> 4) owns the 16-B pieces g (k or column half 0) and 4 + g (half 1)
// of row r, and a row is one 128-B line: an instruction per half touches 16 half lines, which the
// vector memory path serves at about half the rate of 8 full lines (tools/ubench/store_bench.hip).
// Instead lane (r, g) accesses piece A = (row r & 7, piece 4 (r >> 3) + g) and piece B = (row
// 8 + (r & 7), same piece) — 8 rows x 128 B per instruction — and swaps one of them with lane r ^ 8:
//     half 0 of the own row = swap_piece(A, B, true),   half 1 = swap_piece(B, A, false)   (loads)
//     A = swap_piece(half 0, half 1, true),              B = swap_piece(half 1, half 0, false) (stores)
// Row-octet exchange of 16-byte pieces between lanes r and r ^ 8 of each 16-lane row.  For the
// lanes selected by `low` (true: rows 8-15, false: rows 0-7) the result is the partner's `theirs`;
// the other lanes keep `mine`.  One bank-masked row_ror:8 DPP move per register.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ V swap_piece(V mine, V theirs, bool take_in_upper_rows) {
  static_assert(sizeof(V) == 16, "16-byte pieces");
  u32x4_t m, t, o;
  __builtin_memcpy(&m, &mine, 16);
  __builtin_memcpy(&t, &theirs, 16);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = take_in_upper_rows
               ? (unsigned)__builtin_amdgcn_update_dpp((int)m[i], (int)t[i], 0x128, 0xF, 0xC, false)
               : (unsigned)__builtin_amdgcn_update_dpp((int)m[i], (int)t[i], 0x128, 0xF, 0x3, false);
  V out;
  __builtin_memcpy(&out, &o, 16);
  return out;
}

// 16-byte global stores with an explicit cache policy (gemm.hip documents why: a kernel's output is read by the
// NEXT kernel from any XCD, so it must reach the fabric anyway — written through (sc1) it leaves while the kernel is
// still computing instead of at its end-of-kernel release).  POLICY 0 = plain, 1 = sc1 (write-through, line dropped
// from the XCD's L2), 2 = nt, 3 = sc0 sc1.  Inline asm (no builtin carries the cache bits of a flat global store);
// the trailing s_nop keeps hipcc from overwriting the data registers before the store has read them, the "memory"
// clobber keeps later loads of the same location behind the store in program order.
template <int POLICY, typename P>
__device__ __forceinline__ void store16_policy(P* p, u32x4_t v) {
  if constexpr (POLICY == 1)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POLICY == 2)
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POLICY == 3)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else
    *reinterpret_cast<u32x4_t*>(p) = v;
}
// The same with the address as a wave-uniform base (SGPR pair) + a 32-bit byte offset per lane: no 64-bit address
// registers per store (kernels at their register limit: attn_out.hip)
template <int POLICY>
__device__ __forceinline__ void store16_policy_s(const void* sbase, unsigned voff, u32x4_t v) {
  if constexpr (POLICY == 1)
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
  else
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ u32x4_t as_u32x4(uint4 v) { return u32x4_t{v.x, v.y, v.z, v.w}; }
// the kernels beside the GEMMs whose output is a whole activation / operand buffer (attention output, im2col
// matrix): same reasoning, separate switch for A/B builds (-DOAKE_AUX_STORE_POLICY=n)
#ifndef OAKE_AUX_STORE_POLICY
#define OAKE_AUX_STORE_POLICY 1
#endif
template <typename P>
__device__ __forceinline__ void aux_store16(P* p, uint4 v) {
  store16_policy<OAKE_AUX_STORE_POLICY>(p, as_u32x4(v));
}

// Loads of data that is read exactly once by exactly one CU (the L <= 64 attention's q / k / v rows): marked
// non-temporal (aux 2 / __builtin_nontemporal_load) so that 59 MB of them per launch do not displace the weight
// panels and activation slabs the OTHER lane's GEMM keeps in the L2s.  One lane: neutral (96.7 vs 96.9 k images/s);
// two lanes: +0.75 % (111.3 vs 110.5 k, four interleaved rounds, profiles/r03/ab_session_m_attention_nt.log).
// -DOAKE_STREAM_AUX=0 = default policy (A/B builds).
#ifndef OAKE_STREAM_AUX
#define OAKE_STREAM_AUX 2
#endif
template <typename V>
__device__ __forceinline__ V stream_load16(const V* p) {
#if OAKE_STREAM_AUX == 2
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// QuickGELU (OpenAI CLIP): x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)).
// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an IEEE division: the result is rounded to 16 bits
// right after, and the c_fc epilogue evaluates this 80 times per lane per tile.
__device__ __forceinline__ float quick_gelu(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -2.4554669595930156f);  // 1.702 * log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// Four values at once, the full-rate steps as packed fp32 (v_pk_mul_f32 / v_pk_add_f32: two values per instruction; the
// same operations in the same order as quick_gelu, so the same bits): the c_fc epilogue is bound by the SIMD's issue
// slots — per value 16 + 16 cycles of v_exp / v_rcp and, unpacked, 5 x 4 of the rest.
#ifndef OAKE_GELU_PACKED
#define OAKE_GELU_PACKED 1  // (0: value by value, A/B builds)
#endif
__device__ __forceinline__ void quick_gelu4(f32x4& v) {
#if !OAKE_GELU_PACKED
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
  return;
#endif
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 k = f32x2{-2.4554669595930156f, -2.4554669595930156f}, one = f32x2{1.0f, 1.0f};
  f32x2 a = f32x2{v[0], v[1]}, b = f32x2{v[2], v[3]};
  const f32x2 ta = a * k, tb = b * k;
  const f32x2 da = f32x2{__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1])} + one;
  const f32x2 db = f32x2{__builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])} + one;
  a = a * f32x2{__builtin_amdgcn_rcpf(da[0]), __builtin_amdgcn_rcpf(da[1])};
  b = b * f32x2{__builtin_amdgcn_rcpf(db[0]), __builtin_amdgcn_rcpf(db[1])};
  v = f32x4{a[0], a[1], b[0], b[1]};
}

// XCD-aware bijective block remap (guide T1): hardware places block b on XCD b % 8; give each
// XCD a contiguous range of logical tiles so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  const int q = nwg / nx, r = nwg % nx;
  const int xcd = bid % nx, idx = bid / nx;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace oake
