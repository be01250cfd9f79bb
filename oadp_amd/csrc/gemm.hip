// gemm.hip — C[M,N] = A[M,K] * W[N,K]^T on the gfx950 matrix cores (v_mfma_f32_16x16x32_{f16,bf16}).
//
// This one kernel family carries 98.9 % of encode_image's FLOPs (SURVEY.md §8 A15a,d,f,g): conv1 as
// an im2col GEMM, QKV in-proj, attention out-proj, MLP c_fc (+QuickGELU) and c_proj (+residual).
// Both operands are K-contiguous ("B^T input"), which is exactly PyTorch's nn.Linear / Conv2d
// weight layout, so no weight transposition is needed at load time.
//
// Design (wave64):
//   * Block tile BM x BN x 64, WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile of 16x16 MFMA
//     tiles.  The encoder's GEMMs have M = 12800 (= 256 crops x 50 tokens) and N in {768, 2304,
//     3072}; with 256 CUs the tile shape decides the tail: 160x256 gives 240 / 720 / 960 tiles
//     (94 % of whole CU rounds) where 256x256 gives 150 / 450 / 600 (59 / 88 / 78 %).
//   * Global -> LDS with the LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no
//     staging VGPRs and no ds_write pass.  Double-buffered, one barrier per K-tile; the DMA of tile
//     t+1 is in flight under the MFMAs of tile t.
//   * LDS image: [rows][64] 16-bit = 128 B per row, 16-B chunk index XOR-swizzled with (row>>1)&7
//     (a 256-B bank row holds two tile rows): ds_read_b128 fragment reads are bank-conflict free
//     (SQ_LDS_BANK_CONFLICT = 0 measured).  The DMA writes LDS linearly, so the swizzle is applied
//     to the per-lane SOURCE address (same involution on the read side).
//   * MFMA operands are swapped (W fragment as the A operand, activation fragment as the B operand)
//     so each lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row: the
//     epilogue stores 8 B (16-bit out) or 16 B (fp32 out) per lane.
//   * Tile order: N is cut into panels of `pn` tile-columns, tiles are walked row-major inside a
//     panel, and each XCD (block b runs on XCD b % 8) gets a contiguous range of that order — so the
//     blocks resident on one XCD share a W panel and a few A rows in that XCD's 4 MiB L2 (the naive
//     order streamed all of W through every L2: 26 % L2 misses and a DRAM-bound kernel).
#include "common.h"
#include "kernels.h"

namespace oake {

int g_gemm_variant = -1;  // -1 = auto (per-shape), else forced tile config (debug / A-B runs)

namespace {

constexpr int BK = 64;
constexpr int kRowBytes = BK * 2;  // 128

struct EpiParams {
  const float* bias;
  void* out;
  int ldo;
  const float* pos;
  int P2;
  int L;
};

struct TileMap {
  int tiles_m, tiles_n, pn, nwg;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void tile_of_block(const TileMap& tmap, int bid, int& tm, int& tn) {
  const int t = xcd_remap(bid, tmap.nwg);
  const int full = tmap.tiles_n / tmap.pn;
  const int per_panel = tmap.tiles_m * tmap.pn;
  int panel, pw, rem;
  if (t < full * per_panel) {
    panel = t / per_panel;
    rem = t - panel * per_panel;
    pw = tmap.pn;
  } else {
    panel = full;
    rem = t - full * per_panel;
    pw = tmap.tiles_n - full * tmap.pn;
  }
  tm = rem / pw;
  tn = panel * tmap.pn + (rem - tm * pw);
}

// Wave-level epilogue.  Lane holds, for every (mi, ni), C[mbase + 16 mi][nbase + 16 ni + 0..3]
// (4 consecutive columns).  All bias / residual / pos-emb loads of a row are issued before the
// first store so the wave waits once per row instead of once per 16-B load.
template <typename T, int EPI, int MI, int NI>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[MI][NI], int mbase, int nbase, int M,
                                              int N, const EpiParams& ep, bool reset) {
  float4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = nbase + ni * 16;
    bv[ni] = (EPI != EPI_PATCH && ep.bias != nullptr && n < N)
                 ? *reinterpret_cast<const float4*>(ep.bias + n)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mbase + mi * 16;
    const bool mok = m < M;
    size_t orow = (size_t)m;
    const float* addrow = nullptr;  // residual row (EPI_RESID) or pos-emb row (EPI_PATCH)
    if (EPI == EPI_PATCH) {
      const int img = m / ep.P2;
      const int p = m - img * ep.P2;
      orow = (size_t)img * ep.L + 1 + p;
      addrow = ep.pos + (size_t)(1 + p) * N;
    } else if (EPI == EPI_RESID) {
      addrow = reinterpret_cast<const float*>(ep.out) + orow * ep.ldo;
    }
    float4 rv[NI];
    if (EPI == EPI_PATCH || EPI == EPI_RESID) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = nbase + ni * 16;
        rv[ni] = (mok && n < N) ? *reinterpret_cast<const float4*>(addrow + n)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = nbase + ni * 16;
      f32x4 v = acc[mi][ni];
      if (reset) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!(mok && n < N)) continue;
      v[0] += bv[ni].x; v[1] += bv[ni].y; v[2] += bv[ni].z; v[3] += bv[ni].w;
      if (EPI == EPI_PATCH || EPI == EPI_RESID) {
        v[0] += rv[ni].x; v[1] += rv[ni].y; v[2] += rv[ni].z; v[3] += rv[ni].w;
      }
      if (EPI == EPI_F32_BIAS || EPI == EPI_PATCH || EPI == EPI_RESID) {
        float* o = reinterpret_cast<float*>(ep.out) + orow * ep.ldo + n;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
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

template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const T* __restrict__ A,
                                                           const T* __restrict__ W, int M, int N,
                                                           int K, EpiParams ep, TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;               // 1-KiB DMA pieces per stage
  constexpr int NSLOT = (NINST + NW - 1) / NW;       // pieces per wave (last may be idle)
  static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  int tm, tn;
  tile_of_block(tmap, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA staging: piece ii covers stage rows [8 ii, 8 ii + 8); lane -> (row 8 ii + lane/8,
  //      LDS chunk lane%8) fetching source chunk (lane%8) ^ ((row>>1)&7).
  const char* src[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int ii = wid + NW * j;
    const int r = 8 * ii + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    if (r < BM) {
      int gr = m0 + r;
      gr = gr < M ? gr : M - 1;
      src[j] = reinterpret_cast<const char*>(A + (size_t)gr * K) + chunk * 16;
    } else {
      int gr = n0 + (r - BM);
      gr = gr < N ? gr : N - 1;
      gr = gr < 0 ? 0 : gr;
      src[j] = reinterpret_cast<const char*>(W + (size_t)gr * K) + chunk * 16;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * kStageBytes;
    const size_t koff = (size_t)kt * (BK * 2);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int ii = wid + NW * j;
      if (j < NSLOT - 1 || NINST % NW == 0 || ii < NINST)  // only the last slot can be idle
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + koff), (lds_ptr_t)(base + ii * 1024),
                                         16, 0, 0);
    }
  };

  // ---- fragment reads: lane -> (row = lane&15, k-chunk = lane>>4) ----
  const int frow = lane & 15;
  const int fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;  // TM, TN multiples of 16 keep (row>>1)&7 == (frow>>1)&7
  const int a_base = (wm * TM + frow) * kRowBytes;
  const int b_base = kATileBytes + (wn * TN + frow) * kRowBytes;
  const int koff0 = ((0 * 4 + fg) ^ fsw) << 4;
  const int koff1 = ((1 * 4 + fg) ^ fsw) << 4;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int cur) {
    const char* st = smem + cur * kStageBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int koff = kk == 0 ? koff0 : koff1;
      vec8 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        bf[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = T16<T>::mfma(bf[ni], af[mi], acc[mi][ni]);
    }
  };

  const int nk = K / BK;
  stage(0, 0);
  __syncthreads();  // (carries the vmcnt(0) that retires the DMA)
  for (int kt = 0; kt < nk - 1; ++kt) {  // branch-free body; last tile peeled below
    const int cur = kt & 1;
    stage(kt + 1, cur ^ 1);
    compute(cur);
    __syncthreads();
  }
  compute((nk - 1) & 1);

  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, false);
}

// ---------------------------------------------------------------------------------------------
// Pipelined persistent kernel ("v3"): same tile / LDS image / fragment layout as gemm_kernel, plus
//   * 3 LDS stages: the DMA of K-tile t+2 is issued right after the barrier of iteration t, so it has
//     two full MFMA phases to land — the barrier's vmcnt(0) never waits in steady state;
//   * fragment software pipelining across the barrier: the kk=1 fragments of tile t are read before
//     the kk=0 MFMAs, the kk=0 fragments of tile t+1 right after the barrier — every MFMA block
//     starts with its operands already in registers, so LDS latency and the post-barrier bubble are
//     covered by matrix work;
//   * persistent blocks: one block per CU walks tiles (XCD-contiguous order); the K pipeline runs
//     straight across tile boundaries, so the next tile's first loads fly under the current tile's
//     last MFMAs and its epilogue stores.
template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm3_kernel(const T* __restrict__ A,
                                                            const T* __restrict__ W, int M, int N,
                                                            int K, EpiParams ep, TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;
  constexpr int NSLOT = (NINST + NW - 1) / NW;
  constexpr int NSTAGE = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  // ---- this block's tile list: XCD x owns logical tiles [xb, xb + xc); block (b/8) of that XCD
  //      takes xb + b/8 + i * (blocks per XCD).
  const int nx = 8;
  const int xcd = blockIdx.x % nx, xslot = blockIdx.x / nx;
  const int per_xcd = gridDim.x / nx;  // host launches a multiple of 8 blocks
  const int q = tmap.nwg / nx, r = tmap.nwg % nx;
  const int xb = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xc = xcd < r ? q + 1 : q;
  const int my_tiles = xslot < xc ? (xc - xslot + per_xcd - 1) / per_xcd : 0;
  if (my_tiles == 0) return;
  const int nk = K / BK;
  const int total = my_tiles * nk;

  auto tile_coords = [&](int i, int& m0, int& n0) {
    const int t = xb + xslot + i * per_xcd;
    const int full = tmap.tiles_n / tmap.pn;
    const int per_panel = tmap.tiles_m * tmap.pn;
    int panel, pw, rem;
    if (t < full * per_panel) {
      panel = t / per_panel; rem = t - panel * per_panel; pw = tmap.pn;
    } else {
      panel = full; rem = t - full * per_panel; pw = tmap.tiles_n - full * tmap.pn;
    }
    const int tm = rem / pw;
    m0 = tm * BM;
    n0 = (panel * tmap.pn + (rem - tm * pw)) * BN;
  };

  // ---- producer state: the tile / k-tile being staged ----
  const char* src[NSLOT];
  auto set_src = [&](int tile_i) {
    int m0, n0;
    tile_coords(tile_i, m0, n0);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int ii = wid + NW * j;
      const int rr = 8 * ii + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rr >> 1) & 7);
      if (rr < BM) {
        int gr = m0 + rr;
        gr = gr < M ? gr : M - 1;
        src[j] = reinterpret_cast<const char*>(A + (size_t)gr * K) + chunk * 16;
      } else {
        int gr = n0 + (rr - BM);
        gr = gr < N ? gr : N - 1;
        gr = gr < 0 ? 0 : gr;
        src[j] = reinterpret_cast<const char*>(W + (size_t)gr * K) + chunk * 16;
      }
    }
  };
  int s_it = 0;      // flat iteration being staged
  int s_kt = 0;      // its k-tile
  int s_tile = 0;    // its tile ordinal
  int s_buf = 0;
  auto stage_next = [&]() {
    if (s_it < total) {
      char* base = smem + s_buf * kStageBytes;
      const size_t koff = (size_t)s_kt * (BK * 2);
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        const int ii = wid + NW * j;
        if (j < NSLOT - 1 || NINST % NW == 0 || ii < NINST)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + koff),
                                           (lds_ptr_t)(base + ii * 1024), 16, 0, 0);
      }
      ++s_it;
      s_buf = s_buf == NSTAGE - 1 ? 0 : s_buf + 1;
      if (++s_kt == nk) {
        s_kt = 0;
        ++s_tile;
        if (s_tile < my_tiles) set_src(s_tile);
      }
    }
  };

  // ---- fragment addressing ----
  const int frow = lane & 15;
  const int fg = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int a_base = (wm * TM + frow) * kRowBytes;
  const int b_base = kATileBytes + (wn * TN + frow) * kRowBytes;
  const int koff0 = ((0 * 4 + fg) ^ fsw) << 4;
  const int koff1 = ((1 * 4 + fg) ^ fsw) << 4;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  vec8 a0[MI], b0[NI], a1[MI], b1[NI];
#define OAKE_LOAD_FRAGS(af_, bf_, buf_, koff_)                                              \
  do {                                                                                      \
    const char* _st = smem + (buf_) * kStageBytes;                                          \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
        af_[i] = *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + (koff_)); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                          \
        bf_[i] = *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + (koff_)); \
  } while (0)
#define OAKE_MFMA_BLOCK(af_, bf_)                                                           \
  do {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
            acc[mi][ni] = T16<T>::mfma(bf_[ni], af_[mi], acc[mi][ni]);                      \
  } while (0)

  // ---- prologue: two K-tiles in flight, first fragments in registers ----
  set_src(0);
  stage_next();
  stage_next();
  // wait for the first tile only (the second stays in flight): vmcnt counts this wave's pieces
  if (total > 1) {
    if (NINST % NW == 0 || wid + NW * (NSLOT - 1) < NINST)
      __builtin_amdgcn_s_waitcnt(0x0F70 | (NSLOT & 15) | ((NSLOT >> 4) << 14));
    else
      __builtin_amdgcn_s_waitcnt(0x0F70 | ((NSLOT - 1) & 15) | (((NSLOT - 1) >> 4) << 14));
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  __builtin_amdgcn_s_barrier();
  OAKE_LOAD_FRAGS(a0, b0, 0, koff0);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): same loop-entry state as the back edge

  int c_buf = 0;
  int c_kt = 0;
  int c_tile = 0;
  for (int it = 0; it < total; ++it) {
    const int nbuf = c_buf == NSTAGE - 1 ? 0 : c_buf + 1;
    // sched_barrier(0) pins the block order: without it hipcc hoists the barrier above the MFMAs
    // (they touch no memory) and the wave then waits for LDS with an idle matrix pipe.
    OAKE_LOAD_FRAGS(a1, b1, c_buf, koff1);   // kk = 1 of this K-tile (lands under the kk = 0 MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    OAKE_MFMA_BLOCK(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                          // K-tile it+1 has landed; everyone is done with it-1
    __builtin_amdgcn_sched_barrier(0);
    stage_next();                             // K-tile it+2 -> the buffer K-tile it-1 occupied
    if (it + 1 < total) OAKE_LOAD_FRAGS(a0, b0, nbuf, koff0);
    __builtin_amdgcn_sched_barrier(0);
    OAKE_MFMA_BLOCK(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    // Retire the a0/b0 reads here (free: they landed under the MFMAs above) so that next
    // iteration's kk=0 MFMAs need no lgkmcnt wait behind the freshly issued kk=1 reads.
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    c_buf = nbuf;
    if (++c_kt == nk) {
      c_kt = 0;
      int m0, n0;
      tile_coords(c_tile, m0, n0);
      ++c_tile;
      tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, true);
    }
  }
#undef OAKE_LOAD_FRAGS
#undef OAKE_MFMA_BLOCK
}

template <typename T, int EPI, int BM, int BN, int WM, int WN>
hipError_t launch_cfg3(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * kRowBytes;
  static bool attr_set = false;
  static int num_cu = 0;
  auto kern = gemm3_kernel<T, EPI, BM, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    int dev = 0;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    num_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  TileMap tmap;
  tmap.tiles_m = (a.M + BM - 1) / BM;
  tmap.tiles_n = (a.N + BN - 1) / BN;
  tmap.nwg = tmap.tiles_m * tmap.tiles_n;
  int pn = 768 / BN;
  pn = pn < 1 ? 1 : pn;
  tmap.pn = pn > tmap.tiles_n ? tmap.tiles_n : pn;
  int grid = (num_cu / 8) * 8;           // one persistent block per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  const int need = ((tmap.nwg + 7) / 8) * 8;
  if (grid > need) grid = need;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, s, reinterpret_cast<const T*>(a.A),
                     reinterpret_cast<const T*>(a.W), a.M, a.N, a.K, ep, tmap);
  return hipGetLastError();
}

template <typename T, int EPI, int BM, int BN, int WM, int WN>
hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * kRowBytes;
  static bool attr_set = false;
  auto kern = gemm_kernel<T, EPI, BM, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  TileMap tmap;
  tmap.tiles_m = (a.M + BM - 1) / BM;
  tmap.tiles_n = (a.N + BN - 1) / BN;
  tmap.nwg = tmap.tiles_m * tmap.tiles_n;
  int pn = 768 / BN;  // ~768-column panels: a W panel of K=768 is ~1.2 MB of an XCD's 4 MiB L2
  pn = pn < 1 ? 1 : pn;
  tmap.pn = pn > tmap.tiles_n ? tmap.tiles_n : pn;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L};
  hipLaunchKernelGGL(kern, dim3(tmap.nwg), dim3(WM * WN * 64), lds, s,
                     reinterpret_cast<const T*>(a.A), reinterpret_cast<const T*>(a.W), a.M, a.N,
                     a.K, ep, tmap);
  return hipGetLastError();
}

// Tile configurations.  0: 128x128 (4 waves)  1: 160x256 (8 waves 2x4)  2: 320x128 (8 waves 4x2)
//                       3: 256x256 (8 waves 2x4)   4: pipelined persistent 160x256   5: pipelined 128x128
template <typename T, int EPI>
hipError_t launch_variant(int variant, const GemmArgs& a, hipStream_t s) {
  switch (variant) {
    case 0: return launch_cfg<T, EPI, 128, 128, 2, 2>(a, s);
    case 1: return launch_cfg<T, EPI, 160, 256, 2, 4>(a, s);
    case 2: return launch_cfg<T, EPI, 320, 128, 4, 2>(a, s);
    case 3: return launch_cfg<T, EPI, 256, 256, 2, 4>(a, s);
    case 4: return launch_cfg3<T, EPI, 160, 256, 2, 4>(a, s);
    case 5: return launch_cfg3<T, EPI, 128, 128, 2, 2>(a, s);
    default: return hipErrorInvalidValue;
  }
}

int pick_variant(const GemmArgs& a) {
  if (g_gemm_variant >= 0) return g_gemm_variant;
  if (a.M <= 1024 || a.N < 256) return 0;
  return 2;
}

template <typename T>
hipError_t launch_epi(int epi, const GemmArgs& a, hipStream_t s) {
  const int v = pick_variant(a);
  switch (epi) {
    case EPI_F32_BIAS: return launch_variant<T, EPI_F32_BIAS>(v, a, s);
    case EPI_T16_BIAS: return launch_variant<T, EPI_T16_BIAS>(v, a, s);
    case EPI_T16_GELU: return launch_variant<T, EPI_T16_GELU>(v, a, s);
    case EPI_RESID: return launch_variant<T, EPI_RESID>(v, a, s);
    case EPI_PATCH: return launch_variant<T, EPI_PATCH>(v, a, s);
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
