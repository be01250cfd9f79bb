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
#include <cstdio>

#include "common.h"
#include "kernels.h"

namespace oake {

int g_gemm_variant = -1;  // -1 = auto (per-shape), else forced tile config (debug / A-B runs)
int g_gemm_krot = 0;      // rotate the K walk per tile (measured: worse — lockstep sharers merge in L2)
unsigned long long* g_gemm_trace = nullptr;  // debug: per-iteration s_memtime stamps (gemm_kernel)

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
  int krot;  // 1 = rotate the K walk per tile (see gemm_kernel)
  unsigned long long* trace;  // debug: [block][iter][4] cycle stamps for blocks < 16, or nullptr
};

__device__ __forceinline__ bool g_krot_enabled_dev(const TileMap& t) { return t.krot != 0; }

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
template <typename T, int EPI, int MI, int NI, bool FULL>
__device__ __forceinline__ void tile_epilogue_impl(f32x4 (&acc)[MI][NI], int mbase, int nbase, int M,
                                                   int N, const EpiParams& ep, bool reset) {
  float4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = nbase + ni * 16;
    bv[ni] = (EPI != EPI_PATCH && ep.bias != nullptr && (FULL || n < N))
                 ? *reinterpret_cast<const float4*>(ep.bias + n)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mbase + mi * 16;
    const bool mok = FULL || m < M;
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
        rv[ni] = (FULL || (mok && n < N)) ? *reinterpret_cast<const float4*>(addrow + n)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = nbase + ni * 16;
      f32x4 v = acc[mi][ni];
      if (reset) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!FULL && !(mok && n < N)) continue;
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

// Interior tiles (the common case) take a branch-free epilogue: with per-store exec-mask branches
// hipcc put an s_waitcnt vmcnt(0) in front of every store, serialising ~20 store latencies per tile.
template <typename T, int EPI, int MI, int NI>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[MI][NI], int mbase, int nbase, int M,
                                              int N, const EpiParams& ep, bool reset, bool interior) {
  if (interior)
    tile_epilogue_impl<T, EPI, MI, NI, true>(acc, mbase, nbase, M, N, ep, reset);
  else
    tile_epilogue_impl<T, EPI, MI, NI, false>(acc, mbase, nbase, M, N, ep, reset);
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
  // K-tiles are walked in a per-tile rotated order: blocks that share an A row-panel or a W
  // column-panel on an XCD then touch different K-slices at any moment, so each slice is pulled
  // into L2 by ONE block and is a hit for the others (instead of every block first-touching, and
  // stalling on, the same slice at the same time).
  const int nk_rot = K / BK;
  int krot = 0;
  if (g_krot_enabled_dev(tmap)) {
    const int step = nk_rot >= 6 ? nk_rot / 6 : 1;
    krot = ((tm + tn) * step) % nk_rot;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * kStageBytes;
    int ks = kt + krot;
    ks = ks >= nk_rot ? ks - nk_rot : ks;
    const size_t koff = (size_t)ks * (BK * 2);
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
  if (tmap.trace != nullptr) {
    // traced copy of the loop (debug only): stamps = top, after DMA issue, after MFMAs, after barrier
    const bool rec = blockIdx.x < 16 && tid == 0;
    unsigned long long* tr = tmap.trace + (size_t)blockIdx.x * 64 * 4;
    for (int kt = 0; kt < nk - 1; ++kt) {
      const int cur = kt & 1;
      const unsigned long long t0 = __builtin_readcyclecounter();
      stage(kt + 1, cur ^ 1);
      const unsigned long long t1 = __builtin_readcyclecounter();
      compute(cur);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t2 = __builtin_readcyclecounter();
      __syncthreads();
      const unsigned long long t3 = __builtin_readcyclecounter();
      if (rec && kt < 64) { tr[kt * 4 + 0] = t0; tr[kt * 4 + 1] = t1; tr[kt * 4 + 2] = t2; tr[kt * 4 + 3] = t3; }
    }
  } else {
    for (int kt = 0; kt < nk - 1; ++kt) {  // branch-free body; last tile peeled below
      const int cur = kt & 1;
      stage(kt + 1, cur ^ 1);
      compute(cur);
      __syncthreads();
    }
  }
  compute((nk - 1) & 1);

  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, false,
                                m0 + BM <= M && n0 + BN <= N);
}

// ---------------------------------------------------------------------------------------------
// Pipelined persistent kernel ("v3"): same tile / LDS image / fragment layout as gemm_kernel, plus
//   * a 3-slot LDS ring with the DMA running TWO K-tiles ahead: the K-tile read in iteration j+3 is
//     issued right after the barrier of iteration j (its slot's last reader finished before that
//     barrier), and the barrier of iteration j waits with a COUNTED vmcnt that leaves the newest
//     K-tile in flight — the DMA latency has two full MFMA phases to hide in;
//   * fragment software pipelining across the barrier: the kk=1 fragments of K-tile j are read
//     before the kk=0 MFMAs, the kk=0 fragments of K-tile j+1 right after the barrier — every MFMA
//     block starts with its operands already in registers;
//   * the DMA pieces are issued between MFMAs (an LDS-DMA piece costs ~50 issue cycles), not as a
//     burst in front of them;
//   * persistent blocks, one per CU, walking tiles in XCD-contiguous order; the K pipeline runs
//     straight across tile boundaries (only the epilogue's own loads drain it).
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
  constexpr int NFULL = NINST - NW * (NSLOT - 1);  // waves with wid < NFULL own NSLOT pieces
  constexpr int NSTAGE = 3;
  static_assert(NSLOT <= 15 && NSLOT >= 2, "vmcnt immediates below assume 2..15 pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bool full_wave = (NINST % NW == 0) || (wid < NFULL);

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

  // ---- producer state: the K-tile being staged next ----
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
  int s_it = 0;      // flat K-tile index being staged
  int s_kt = 0;      // its k position inside the output tile
  int s_tile = 0;    // its output-tile ordinal
  int s_buf = 0;
  char* s_base = smem;
  size_t s_koff = 0;
  // one DMA piece (slot j) of the K-tile being staged; a no-op once everything is staged
  auto stage_piece = [&](int j) {
    const int ii = wid + NW * j;
    if (s_it < total && (j < NSLOT - 1 || full_wave))
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + s_koff), (lds_ptr_t)(s_base + ii * 1024),
                                       16, 0, 0);
  };
  auto stage_advance = [&]() {
    if (s_it < total) {
      ++s_it;
      s_buf = s_buf == NSTAGE - 1 ? 0 : s_buf + 1;
      s_base = smem + s_buf * kStageBytes;
      if (++s_kt == nk) {
        s_kt = 0;
        ++s_tile;
        if (s_tile < my_tiles) set_src(s_tile);
      }
      s_koff = (size_t)s_kt * (BK * 2);
    }
  };
  auto stage_all = [&]() {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) stage_piece(j);
    stage_advance();
  };
  // wait until at most `tiles_left_in_flight` of this wave's newest K-tiles are outstanding
  auto wait_tiles = [&](int tiles_in_flight) {
    // vmcnt immediate: bits [3:0] | [15:14]; expcnt 7 and lgkmcnt 15 = "don't wait"
    if (tiles_in_flight <= 0) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else if (tiles_in_flight == 1) {
      if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | NSLOT);
      else __builtin_amdgcn_s_waitcnt(0x0F70 | (NSLOT - 1));
    } else {
      if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * NSLOT) & 15) | (((2 * NSLOT) >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * NSLOT - 2) & 15) | (((2 * NSLOT - 2) >> 4) << 14));
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

  // ---- prologue: three K-tiles in flight, first fragments in registers ----
  set_src(0);
  stage_all();
  stage_all();
  stage_all();
  wait_tiles(total >= 3 ? 2 : total - 1);   // K-tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  OAKE_LOAD_FRAGS(a0, b0, 0, koff0);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)

  const bool rec = tmap.trace != nullptr && blockIdx.x < 16 && tid == 0;
  unsigned long long* tr = tmap.trace != nullptr ? tmap.trace + (size_t)blockIdx.x * 64 * 4 : nullptr;

  int c_buf = 0;
  int c_kt = 0;
  int c_tile = 0;
  for (int it = 0; it < total; ++it) {
    const int nbuf = c_buf == NSTAGE - 1 ? 0 : c_buf + 1;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (rec) t0 = __builtin_readcyclecounter();
    // -- block 1: kk=1 fragment reads land under the kk=0 MFMAs (sched_barrier pins the order:
    //    hipcc otherwise hoists the barrier above the MFMAs and parks the wave on LDS latency)
    OAKE_LOAD_FRAGS(a1, b1, c_buf, koff1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T16<T>::mfma(b0[ni], a0[mi], acc[mi][ni]);
    __builtin_amdgcn_sched_barrier(0);
    if (rec) t1 = __builtin_readcyclecounter();
    // -- barrier: K-tile it+1 landed everywhere (K-tile it+2 may still be in flight); all waves are
    //    done reading K-tile it's slot
    wait_tiles(s_it - (it + 2));               // in flight beyond it+1: 0, 1 (steady state) ...
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): a1/b1 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (rec) t2 = __builtin_readcyclecounter();
    // -- block 2: kk=0 fragments of the next K-tile + kk=1 MFMAs, DMA of K-tile it+3 in between
    if (it + 1 < total) OAKE_LOAD_FRAGS(a0, b0, nbuf, koff0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[mi][ni] = T16<T>::mfma(b1[ni], a1[mi], acc[mi][ni]);
        constexpr int kEvery = (MI * NI) / NSLOT > 0 ? (MI * NI) / NSLOT : 1;
        const int idx = mi * NI + ni;
        if (idx % kEvery == kEvery - 1 && idx / kEvery < NSLOT) {
          __builtin_amdgcn_sched_barrier(0);
          stage_piece(idx / kEvery);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    stage_advance();
    __builtin_amdgcn_sched_barrier(0);
    // Retire the a0/b0 reads here (free: they landed under the MFMAs above) so that next
    // iteration's kk=0 MFMAs need no lgkmcnt wait behind the freshly issued kk=1 reads.
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    if (rec && it < 64) {
      tr[it * 4 + 0] = t0; tr[it * 4 + 1] = t1; tr[it * 4 + 2] = t2;
      tr[it * 4 + 3] = __builtin_readcyclecounter();
    }
    c_buf = nbuf;
    if (++c_kt == nk) {
      c_kt = 0;
      int m0, n0;
      tile_coords(c_tile, m0, n0);
      ++c_tile;
      tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, true,
                                    m0 + BM <= M && n0 + BN <= N);
      // drain the epilogue's own loads/stores so no VMEM result is pending at the loop header
      // (otherwise hipcc guards the loop-top ds_reads with vmcnt(0) every iteration)
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
  }
#undef OAKE_LOAD_FRAGS
}

// ---------------------------------------------------------------------------------------------
// Ping-pong kernel ("v5").  Measured on v2/v3 (tools/gemm_trace.py): an LDS-DMA piece costs the
// issuing wave ~80 cycles, and because the two waves that share a SIMD ran the SAME phase at the same
// time (the per-K-tile barrier re-aligns them), those stalls were never covered by the partner's
// MFMAs: per K-tile 1280 cycles of matrix work + ~900 of DMA/LDS issue + barrier = 2250.
// Here every wave runs the same 4-phase loop per K-tile —
//     LOAD0: read kk0 fragments, issue half of its DMA pieces of K-tile j+2      | barrier
//     MFMA0: 20 MFMAs                                                            | barrier
//     LOAD1: read kk1 fragments, issue the other half, counted vmcnt for K-tile j+1 | barrier
//     MFMA1: 20 MFMAs                                                            | barrier
// — but waves 4-7 (one per SIMD, like waves 0-3) execute ONE extra barrier before the loop and waves
// 0-3 one after it, so the two waves of a SIMD are always one phase apart: while one issues nothing
// but MFMAs the other does its LDS reads and DMA issue.  3-slot LDS ring: the barrier that ends a
// wave's LOAD1(j) publishes its pieces of K-tile j+1; K-tile j+2 is written into the slot K-tile
// j-1 occupied, whose last readers (LOAD1(j-1)) are behind a barrier by then.
template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm5_kernel(const T* __restrict__ A,
                                                            const T* __restrict__ W, int M, int N,
                                                            int K, EpiParams ep, TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = WM * WN;
  static_assert(NW == 8, "two groups of four waves");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;
  constexpr int NSLOT = (NINST + NW - 1) / NW;
  constexpr int NFULL = NINST - NW * (NSLOT - 1);
  constexpr int NHALF = (NSLOT + 1) / 2;  // pieces issued in LOAD0; the rest in LOAD1
  constexpr int NSTAGE = 3;
  static_assert(NSLOT <= 15 && NSLOT >= 2, "vmcnt immediates assume 2..15 pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bool late_group = wid >= 4;
  const bool full_wave = (NINST % NW == 0) || (wid < NFULL);

  int tm, tn;
  tile_of_block(tmap, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = K / BK;

  const char* src[NSLOT];
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
  // pieces [j0, j1) of K-tile kt into ring slot kt % 3 (no-op past the end)
#define OAKE_STAGE(kt_, j0_, j1_)                                                            \
  do {                                                                                       \
    if ((kt_) < nk) {                                                                        \
      char* _base = smem + ((kt_) % NSTAGE) * kStageBytes;                                   \
      const size_t _koff = (size_t)(kt_) * (BK * 2);                                         \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j) {                             \
        const int _ii = wid + NW * _j;                                                       \
        if (_j < NSLOT - 1 || full_wave)                                                     \
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[_j] + _koff),                     \
                                           (lds_ptr_t)(_base + _ii * 1024), 16, 0, 0);       \
      }                                                                                      \
    }                                                                                        \
  } while (0)

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
  vec8 af[MI], bf[NI];

#define OAKE_LOAD_FRAGS(kt_, koff_)                                                         \
  do {                                                                                      \
    const char* _st = smem + ((kt_) % NSTAGE) * kStageBytes;                                \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
        af[i] = *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + (koff_)); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                          \
        bf[i] = *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + (koff_)); \
  } while (0)
#define OAKE_MFMA_BLOCK()                                                                   \
  do {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
            acc[mi][ni] = T16<T>::mfma(bf[ni], af[mi], acc[mi][ni]);                        \
  } while (0)
#define OAKE_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define OAKE_PIN() __builtin_amdgcn_sched_barrier(0)
#define OAKE_BAR()                   \
  do {                               \
    OAKE_PIN();                      \
    __builtin_amdgcn_s_barrier();    \
    OAKE_PIN();                      \
  } while (0)

  // ---- prologue: K-tiles 0 and 1 in flight; publish K-tile 0 ----
  OAKE_STAGE(0, 0, NSLOT);
  OAKE_STAGE(1, 0, NSLOT);
  if (nk >= 2) {
    if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | NSLOT);
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (NSLOT - 1));
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  OAKE_BAR();
  if (late_group) OAKE_BAR();  // waves 4-7 run one phase behind waves 0-3

  const bool rec = tmap.trace != nullptr && blockIdx.x < 16 && (tid == 0 || tid == 256);
  unsigned long long* tr =
      tmap.trace != nullptr ? tmap.trace + ((size_t)blockIdx.x * 2 + (tid >> 8)) * 32 * 4 : nullptr;
  for (int j = 0; j < nk; ++j) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (rec) t0 = __builtin_readcyclecounter();
    // LOAD0
    OAKE_LOAD_FRAGS(j, koff0);
    OAKE_STAGE(j + 2, 0, NHALF);
    OAKE_LGKM0();
    OAKE_BAR();
    if (rec) t1 = __builtin_readcyclecounter();
    // MFMA0
    OAKE_MFMA_BLOCK();
    OAKE_BAR();
    if (rec) t2 = __builtin_readcyclecounter();
    // LOAD1
    OAKE_LOAD_FRAGS(j, koff1);
    OAKE_STAGE(j + 2, NHALF, NSLOT);
    OAKE_LGKM0();
    // own pieces of K-tile j+1 landed; K-tile j+2 (if any) stays in flight
    if (j + 2 < nk) {
      if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | NSLOT);
      else __builtin_amdgcn_s_waitcnt(0x0F70 | (NSLOT - 1));
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    OAKE_BAR();
    if (rec && j < 32) {
      tr[j * 4 + 0] = t0; tr[j * 4 + 1] = t1; tr[j * 4 + 2] = t2;
      tr[j * 4 + 3] = __builtin_readcyclecounter();
    }
    // MFMA1
    OAKE_MFMA_BLOCK();
    OAKE_BAR();
  }
  if (!late_group) OAKE_BAR();
#undef OAKE_STAGE
#undef OAKE_LOAD_FRAGS
#undef OAKE_MFMA_BLOCK
#undef OAKE_LGKM0
#undef OAKE_PIN
#undef OAKE_BAR
  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, false,
                                m0 + BM <= M && n0 + BN <= N);
}

// ---------------------------------------------------------------------------------------------
// "v6": ping-pong (see gemm5_kernel) with the fragment reads moved INTO the MFMA phases
// (double-buffered fragments: the kk1 fragments of K-tile j are read between the kk0 MFMAs, the kk0
// fragments of K-tile j+1 between the kk1 MFMAs), so a LOAD phase is nothing but ~3 DMA pieces and a
// counted wait and is shorter than the partner's 20-MFMA phase.  Traced phase lengths (v5): LOAD
// ~500 cycles vs MFMA ~320 — the LOAD phases were the critical path.
// Early waves (0-3) and late waves (4-7, one barrier behind) place their DMA / waits differently —
// the early wave is constrained by RAW (the late wave publishes its pieces one barrier later), the
// late wave by nothing, so it issues earlier:
//     early: L0(j): 1st half of K-tile j+2          L1(j): 2nd half of j+2, wait own pieces of j+1
//     late : L0(j): 2nd half of j+2, wait j+1       L1(j): 1st half of K-tile j+3
template <typename T, int EPI, int BM, int BN, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm6_kernel(const T* __restrict__ A,
                                                            const T* __restrict__ W, int M, int N,
                                                            int K, EpiParams ep, TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = WM * WN;
  static_assert(NW == 8, "two groups of four waves");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;
  constexpr int NSLOT = (NINST + NW - 1) / NW;
  constexpr int NFULL = NINST - NW * (NSLOT - 1);
  constexpr int NHALF = (NSLOT + 1) / 2;
  constexpr int NSTAGE = 3;
  static_assert(NSLOT <= 15 && NSLOT >= 2 && NHALF < NSLOT, "vmcnt immediates assume 2..15 pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bool late = wid >= 4;
  const bool full_wave = (NINST % NW == 0) || (wid < NFULL);
  constexpr bool abl_nodma = ABL == 1;   // timing ablations (wrong results by design)
  constexpr bool abl_nomfma = ABL == 2;

  int tm, tn;
  tile_of_block(tmap, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = K / BK;

  const char* src[NSLOT];
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
#define OAKE_STAGE(kt_, j0_, j1_)                                                            \
  do {                                                                                       \
    if ((kt_) < nk && !(abl_nodma && (kt_) >= 3)) {                                                                        \
      char* _base = smem + ((kt_) % NSTAGE) * kStageBytes;                                   \
      const size_t _koff = (size_t)(kt_) * (BK * 2);                                         \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j) {                             \
        const int _ii = wid + NW * _j;                                                       \
        if (_j < NSLOT - 1 || full_wave)                                                     \
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[_j] + _koff),                     \
                                           (lds_ptr_t)(_base + _ii * 1024), 16, 0, 0);       \
      }                                                                                      \
    }                                                                                        \
  } while (0)
  // wait until only this wave's pieces of ONE (the newest, complete) K-tile may be outstanding
#define OAKE_WAIT_LEAVE_ONE(newer_exists_)                                                   \
  do {                                                                                       \
    if (!(newer_exists_)) __builtin_amdgcn_s_waitcnt(0x0F70);                                \
    else if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | NSLOT);                          \
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (NSLOT - 1));                                   \
  } while (0)

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

#define OAKE_FRAG_PTR(kt_) (smem + ((kt_) % NSTAGE) * kStageBytes)
#define OAKE_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define OAKE_PIN() __builtin_amdgcn_sched_barrier(0)
#define OAKE_BAR()                   \
  do {                               \
    OAKE_PIN();                      \
    __builtin_amdgcn_s_barrier();    \
    OAKE_PIN();                      \
  } while (0)
  // 20 MFMAs on (af_, bf_) with the MI+NI fragment reads of the NEXT block spread between them
#define OAKE_MFMA_AND_READ(af_, bf_, naf_, nbf_, st_, koff_, do_read_)                        \
  do {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                     \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                   \
        if (!abl_nomfma) acc[mi][ni] = T16<T>::mfma(bf_[ni], af_[mi], acc[mi][ni]);         \
        const int _idx = mi * NI + ni;                                                      \
        if ((do_read_) && (_idx & 1) && (_idx >> 1) < MI + NI) {                            \
          const int _f = _idx >> 1;                                                         \
          if (_f < MI)                                                                      \
            naf_[_f] = *reinterpret_cast<const vec8*>((st_) + a_base + _f * 16 * kRowBytes + (koff_)); \
          else                                                                              \
            nbf_[_f - MI] = *reinterpret_cast<const vec8*>((st_) + b_base + (_f - MI) * 16 * kRowBytes + (koff_)); \
        }                                                                                   \
      }                                                                                     \
    }                                                                                       \
  } while (0)
  static_assert(2 * (MI + NI) <= MI * NI, "not enough MFMA slots to hide the fragment reads");

  // ---- prologue ----
  OAKE_STAGE(0, 0, NSLOT);
  OAKE_STAGE(1, 0, NSLOT);
  if (late) OAKE_STAGE(2, 0, NHALF);
  {
    // own pieces of K-tile 0 landed; everything newer may stay in flight
    const bool t1 = nk >= 2, t2h = late && nk >= 3;
    if (!t1) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else if (!t2h) {
      if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | NSLOT);
      else __builtin_amdgcn_s_waitcnt(0x0F70 | (NSLOT - 1));
    } else {
      if (full_wave) __builtin_amdgcn_s_waitcnt(0x0F70 | ((NSLOT + NHALF) & 15) | (((NSLOT + NHALF) >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70 | ((NSLOT - 1 + NHALF) & 15) | (((NSLOT - 1 + NHALF) >> 4) << 14));
    }
  }
  OAKE_BAR();
  {
    const char* st = OAKE_FRAG_PTR(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
      a0[i] = *reinterpret_cast<const vec8*>(st + a_base + i * 16 * kRowBytes + koff0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      b0[i] = *reinterpret_cast<const vec8*>(st + b_base + i * 16 * kRowBytes + koff0);
    OAKE_LGKM0();
  }
  if (late) OAKE_BAR();  // waves 4-7 run one phase behind waves 0-3

  const bool rec = tmap.trace != nullptr && blockIdx.x < 16 && (tid == 0 || tid == 256);
  unsigned long long* tr =
      tmap.trace != nullptr ? tmap.trace + ((size_t)blockIdx.x * 2 + (tid >> 8)) * 32 * 4 : nullptr;
  for (int j = 0; j < nk; ++j) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (rec) t0 = __builtin_readcyclecounter();
    // ---- L0(j) ----
    if (!late) {
      OAKE_STAGE(j + 2, 0, NHALF);
    } else {
      OAKE_STAGE(j + 2, NHALF, NSLOT);
      OAKE_WAIT_LEAVE_ONE(j + 2 < nk);  // own pieces of K-tile j+1
    }
    OAKE_BAR();
    if (rec) t1 = __builtin_readcyclecounter();
    // ---- M0(j): kk0 MFMAs, kk1 fragment reads of K-tile j in between ----
    {
      const char* st = OAKE_FRAG_PTR(j);
      OAKE_MFMA_AND_READ(a0, b0, a1, b1, st, koff1, true);
      OAKE_PIN();  // keep the MFMAs above the wait: they do not depend on these reads
      OAKE_LGKM0();
    }
    OAKE_BAR();
    if (rec) t2 = __builtin_readcyclecounter();
    // ---- L1(j) ----
    if (!late) {
      OAKE_STAGE(j + 2, NHALF, NSLOT);
      OAKE_WAIT_LEAVE_ONE(j + 2 < nk);  // own pieces of K-tile j+1
    } else {
      OAKE_STAGE(j + 3, 0, NHALF);
    }
    OAKE_BAR();
    if (rec && j < 32) {
      tr[j * 4 + 0] = t0; tr[j * 4 + 1] = t1; tr[j * 4 + 2] = t2;
      tr[j * 4 + 3] = __builtin_readcyclecounter();
    }
    // ---- M1(j): kk1 MFMAs, kk0 fragment reads of K-tile j+1 in between ----
    {
      const char* st = OAKE_FRAG_PTR(j + 1);
      const bool more = j + 1 < nk;
      OAKE_MFMA_AND_READ(a1, b1, a0, b0, st, koff0, more);
      OAKE_PIN();
      OAKE_LGKM0();
    }
    OAKE_BAR();
  }
  if (!late) OAKE_BAR();
#undef OAKE_STAGE
#undef OAKE_WAIT_LEAVE_ONE
#undef OAKE_FRAG_PTR
#undef OAKE_LGKM0
#undef OAKE_PIN
#undef OAKE_BAR
#undef OAKE_MFMA_AND_READ
  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, false,
                                m0 + BM <= M && n0 + BN <= N);
}

// ---------------------------------------------------------------------------------------------
// "v7": ping-pong compute waves + dedicated DMA waves.  Phase traces of v5 (tools/gemm_trace5.py):
// MFMA phase ~320 cycles of matrix work, LOAD phase ~500 = ~250 for the 9 fragment reads + ~250 for
// the wave's 3-4 LDS-DMA pieces (an LDS-DMA piece stalls its issuing wave ~80 cycles) — the LOAD
// phases set the pace.  Here 4 extra waves (one per SIMD, no accumulators) issue ALL DMA pieces
// (13 per K-tile each) and do the counted vmcnt waits; the 8 compute waves only read fragments and
// issue MFMAs, still in two groups one phase apart.  12 waves per block (3 per SIMD), 3-slot ring.
template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((WM * WN + 4) * 64) void gemm7_kernel(const T* __restrict__ A,
                                                                   const T* __restrict__ W, int M,
                                                                   int N, int K, EpiParams ep,
                                                                   TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = WM * WN;
  static_assert(NW == 8, "two compute groups of four waves");
  constexpr int NL = 4;  // loader waves
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;
  static_assert(NINST % NL == 0, "pieces must split evenly over the loader waves");
  constexpr int NPL = NINST / NL;  // pieces per loader wave per K-tile
  static_assert(NPL <= 31, "vmcnt immediate");
  constexpr int NSTAGE = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned long long t_entry = __builtin_readcyclecounter();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  tile_of_block(tmap, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = K / BK;

#define OAKE_PIN() __builtin_amdgcn_sched_barrier(0)
#define OAKE_BAR()                   \
  do {                               \
    OAKE_PIN();                      \
    __builtin_amdgcn_s_barrier();    \
    OAKE_PIN();                      \
  } while (0)

  if (wid >= NW) {
    // ================= loader wave =================
    const int lw = wid - NW;
    const char* src[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int ii = lw + NL * j;
      const int rr = 8 * ii + (lane >> 3);
      const int chunk = (lane & 7) ^ ((rr >> 1) & 7);
      if (rr < BM) {
        int gr = m0 + rr;
        gr = gr < M ? gr : M - 1;
        src[j] = reinterpret_cast<const char*>(A + (size_t)gr * K) + chunk * 16;
      } else {
        int gr = n0 + (rr - BM);
        gr = gr < N ? gr : N - 1;
        src[j] = reinterpret_cast<const char*>(W + (size_t)gr * K) + chunk * 16;
      }
    }
#define OAKE_STAGE(kt_, j0_, j1_)                                                            \
  do {                                                                                       \
    if ((kt_) < nk) {                                                                        \
      char* _base = smem + ((kt_) % NSTAGE) * kStageBytes;                                   \
      const size_t _koff = (size_t)(kt_) * (BK * 2);                                         \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j)                               \
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[_j] + _koff),                     \
                                           (lds_ptr_t)(_base + (lw + NL * _j) * 1024), 16, 0, 0); \
    }                                                                                        \
  } while (0)
#define OAKE_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
    constexpr int Q1 = (NPL + 3) / 4, Q2 = (2 * NPL + 3) / 4, Q3 = (3 * NPL + 3) / 4;
    OAKE_STAGE(0, 0, NPL);
    OAKE_STAGE(1, 0, NPL);
    if (nk >= 2) OAKE_VMCNT(NPL); else OAKE_VMCNT(0);
    OAKE_BAR();  // b0: K-tile 0 published
    for (int j = 0; j < nk; ++j) {
      OAKE_STAGE(j + 2, 0, Q1);
      OAKE_BAR();
      OAKE_STAGE(j + 2, Q1, Q2);
      OAKE_BAR();
      OAKE_STAGE(j + 2, Q2, Q3);
      OAKE_BAR();
      OAKE_STAGE(j + 2, Q3, NPL);
      if (j + 2 < nk) OAKE_VMCNT(NPL); else OAKE_VMCNT(0);  // K-tile j+1 landed
      OAKE_BAR();  // publishes K-tile j+1; K-tile j's slot is free from here on
    }
    OAKE_BAR();  // pairs with the late compute group's last phase
#undef OAKE_STAGE
#undef OAKE_VMCNT
    return;
  }

  // ================= compute wave =================
  const int wm = wid / WN, wn = wid % WN;
  const bool late = wid >= 4;
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
  vec8 af[MI], bf[NI];
#define OAKE_LOAD_FRAGS(kt_, koff_)                                                         \
  do {                                                                                      \
    const char* _st = smem + ((kt_) % NSTAGE) * kStageBytes;                                \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
        af[i] = *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + (koff_)); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                          \
        bf[i] = *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + (koff_)); \
  } while (0)
#define OAKE_MFMA_BLOCK()                                                                   \
  do {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
            acc[mi][ni] = T16<T>::mfma(bf[ni], af[mi], acc[mi][ni]);                        \
  } while (0)
#define OAKE_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

  OAKE_BAR();                // b0
  const unsigned long long t_loop = __builtin_readcyclecounter();
  if (late) OAKE_BAR();      // waves 4-7 run one phase behind waves 0-3 (and the loaders)
  for (int j = 0; j < nk; ++j) {
    OAKE_LOAD_FRAGS(j, koff0);
    OAKE_LGKM0();
    OAKE_BAR();
    OAKE_MFMA_BLOCK();
    OAKE_BAR();
    OAKE_LOAD_FRAGS(j, koff1);
    OAKE_LGKM0();
    OAKE_BAR();
    OAKE_MFMA_BLOCK();
    OAKE_BAR();
  }
  if (!late) OAKE_BAR();
#undef OAKE_LOAD_FRAGS
#undef OAKE_MFMA_BLOCK
#undef OAKE_LGKM0
#undef OAKE_PIN
#undef OAKE_BAR
  const unsigned long long t_end = __builtin_readcyclecounter();
  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, false,
                                m0 + BM <= M && n0 + BN <= N);
  if (tmap.trace != nullptr && tid == 0 && blockIdx.x < 256) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    unsigned long long* tr = tmap.trace + (size_t)blockIdx.x * 4;
    tr[0] = t_entry; tr[1] = t_loop; tr[2] = t_end; tr[3] = __builtin_readcyclecounter();
  }
}

// ---------------------------------------------------------------------------------------------
// "v8": v7 made persistent.  One block per CU walks its tiles (XCD-contiguous order) and the K-tile
// pipeline — loader waves two K-tiles ahead, compute groups one phase apart — runs straight across
// tile boundaries: no per-tile prologue (first DMA latency) and the epilogue's stores drain under the
// next tile's MFMAs instead of all CUs storing at once at the end of a wave of blocks (with K = 768
// the un-overlapped prologue + epilogue were ~40 % of a tile's time).
template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((WM * WN + 4) * 64) void gemm8_kernel(const T* __restrict__ A,
                                                                   const T* __restrict__ W, int M,
                                                                   int N, int K, EpiParams ep,
                                                                   TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NW = WM * WN;
  static_assert(NW == 8, "two compute groups of four waves");
  constexpr int NL = 4;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;
  static_assert(NINST % NL == 0, "pieces must split evenly over the loader waves");
  constexpr int NPL = NINST / NL;
  static_assert(NPL <= 31, "vmcnt immediate");
  constexpr int NSTAGE = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this block's tile list (see gemm3_kernel) ----
  const int nx = 8;
  const int xcd = blockIdx.x % nx, xslot = blockIdx.x / nx;
  const int per_xcd = gridDim.x / nx;
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

#define OAKE_PIN() __builtin_amdgcn_sched_barrier(0)
#define OAKE_BAR()                   \
  do {                               \
    OAKE_PIN();                      \
    __builtin_amdgcn_s_barrier();    \
    OAKE_PIN();                      \
  } while (0)

  if (wid >= NW) {
    // ================= loader wave =================
    const int lw = wid - NW;
    const char* src[NPL];
    auto set_src = [&](int tile_i) {
      int m0, n0;
      tile_coords(tile_i, m0, n0);
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int ii = lw + NL * j;
        const int rr = 8 * ii + (lane >> 3);
        const int chunk = (lane & 7) ^ ((rr >> 1) & 7);
        if (rr < BM) {
          int gr = m0 + rr;
          gr = gr < M ? gr : M - 1;
          src[j] = reinterpret_cast<const char*>(A + (size_t)gr * K) + chunk * 16;
        } else {
          int gr = n0 + (rr - BM);
          gr = gr < N ? gr : N - 1;
          src[j] = reinterpret_cast<const char*>(W + (size_t)gr * K) + chunk * 16;
        }
      }
    };
    // producer cursor: flat K-tile s_g = s_tile * nk + s_kt goes to ring slot s_buf
    int s_g = 0, s_kt = 0, s_tile = 0, s_buf = 0;
#define OAKE_STAGE(j0_, j1_)                                                                 \
  do {                                                                                       \
    if (s_g < total) {                                                                       \
      char* _base = smem + s_buf * kStageBytes;                                              \
      const size_t _koff = (size_t)s_kt * (BK * 2);                                          \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j)                               \
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[_j] + _koff),                     \
                                           (lds_ptr_t)(_base + (lw + NL * _j) * 1024), 16, 0, 0); \
    }                                                                                        \
  } while (0)
#define OAKE_ADVANCE()                                    \
  do {                                                    \
    if (s_g < total) {                                    \
      ++s_g;                                              \
      s_buf = s_buf == NSTAGE - 1 ? 0 : s_buf + 1;        \
      if (++s_kt == nk) {                                 \
        s_kt = 0;                                         \
        ++s_tile;                                         \
        if (s_tile < my_tiles) set_src(s_tile);           \
      }                                                   \
    }                                                     \
  } while (0)
#define OAKE_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
    constexpr int Q1 = (NPL + 3) / 4, Q2 = (2 * NPL + 3) / 4, Q3 = (3 * NPL + 3) / 4;
    set_src(0);
    OAKE_STAGE(0, NPL);
    OAKE_ADVANCE();
    OAKE_STAGE(0, NPL);
    OAKE_ADVANCE();
    if (total >= 2) OAKE_VMCNT(NPL); else OAKE_VMCNT(0);
    OAKE_BAR();  // b0: flat K-tile 0 published
    for (int g = 0; g < total; ++g) {
      OAKE_STAGE(0, Q1);           // flat K-tile g+2
      OAKE_BAR();
      OAKE_STAGE(Q1, Q2);
      OAKE_BAR();
      OAKE_STAGE(Q2, Q3);
      OAKE_BAR();
      OAKE_STAGE(Q3, NPL);
      const bool newer = g + 2 < total;
      OAKE_ADVANCE();
      if (newer) OAKE_VMCNT(NPL); else OAKE_VMCNT(0);  // flat K-tile g+1 landed
      OAKE_BAR();  // publishes K-tile g+1; K-tile g's slot is free from here on
    }
    OAKE_BAR();
#undef OAKE_STAGE
#undef OAKE_ADVANCE
#undef OAKE_VMCNT
    return;
  }

  // ================= compute wave =================
  const int wm = wid / WN, wn = wid % WN;
  const bool late = wid >= 4;
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
  vec8 af[MI], bf[NI];
#define OAKE_LOAD_FRAGS(buf_, koff_)                                                        \
  do {                                                                                      \
    const char* _st = smem + (buf_) * kStageBytes;                                          \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
        af[i] = *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + (koff_)); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                          \
        bf[i] = *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + (koff_)); \
  } while (0)
#define OAKE_MFMA_BLOCK()                                                                   \
  do {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
            acc[mi][ni] = T16<T>::mfma(bf[ni], af[mi], acc[mi][ni]);                        \
  } while (0)
#define OAKE_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

  OAKE_BAR();                // b0
  if (late) OAKE_BAR();      // waves 4-7 run one phase behind waves 0-3 (and the loaders)
  int c_buf = 0, c_kt = 0, c_tile = 0;
  for (int g = 0; g < total; ++g) {
    OAKE_LOAD_FRAGS(c_buf, koff0);
    OAKE_LGKM0();
    OAKE_BAR();
    OAKE_MFMA_BLOCK();
    OAKE_BAR();
    OAKE_LOAD_FRAGS(c_buf, koff1);
    OAKE_LGKM0();
    OAKE_BAR();
    OAKE_MFMA_BLOCK();
    c_buf = c_buf == NSTAGE - 1 ? 0 : c_buf + 1;
    if (++c_kt == nk) {
      // tile done: the stores below drain while the pipeline carries on with the next tile
      c_kt = 0;
      int m0, n0;
      tile_coords(c_tile, m0, n0);
      ++c_tile;
      OAKE_PIN();
      tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN + 4 * fg, M, N, ep, true,
                                    m0 + BM <= M && n0 + BN <= N);
    }
    OAKE_BAR();
  }
  if (!late) OAKE_BAR();
#undef OAKE_LOAD_FRAGS
#undef OAKE_MFMA_BLOCK
#undef OAKE_LGKM0
#undef OAKE_PIN
#undef OAKE_BAR
}

template <typename T, int EPI, int BM, int BN, int WM, int WN, bool PERSIST>
hipError_t launch_cfg7(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * kRowBytes;
  static bool attr_set = false;
  static int num_cu = 0;
  auto kern = PERSIST ? gemm8_kernel<T, EPI, BM, BN, WM, WN> : gemm7_kernel<T, EPI, BM, BN, WM, WN>;
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
  tmap.krot = 0;
  tmap.trace = g_gemm_trace;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L};
  int grid = tmap.nwg;
  if (PERSIST) {
    grid = (num_cu / 8) * 8;  // one persistent block per CU, a multiple of the 8 XCDs
    if (grid < 8) grid = 8;
    const int need = ((tmap.nwg + 7) / 8) * 8;
    if (grid > need) grid = need;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3((WM * WN + 4) * 64), lds, s,
                     reinterpret_cast<const T*>(a.A), reinterpret_cast<const T*>(a.W), a.M, a.N,
                     a.K, ep, tmap);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) fprintf(stderr, "launch_cfg7 persist=%d grid=%d: %s\n", (int)PERSIST, grid, hipGetErrorString(le));
  return le;
}

template <typename T, int EPI, int BM, int BN, int WM, int WN, int VER>
hipError_t launch_cfg5(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * kRowBytes;
  static bool attr_set = false;
  auto kern = VER == 5   ? gemm5_kernel<T, EPI, BM, BN, WM, WN>
              : VER == 6 ? gemm6_kernel<T, EPI, BM, BN, WM, WN, 0>
              : VER == 61 ? gemm6_kernel<T, EPI, BM, BN, WM, WN, 1>
                          : gemm6_kernel<T, EPI, BM, BN, WM, WN, 2>;
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
  int pn = 768 / BN;
  pn = pn < 1 ? 1 : pn;
  tmap.pn = pn > tmap.tiles_n ? tmap.tiles_n : pn;
  tmap.krot = 0;
  tmap.trace = g_gemm_trace;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L};
  hipLaunchKernelGGL(kern, dim3(tmap.nwg), dim3(WM * WN * 64), lds, s,
                     reinterpret_cast<const T*>(a.A), reinterpret_cast<const T*>(a.W), a.M, a.N,
                     a.K, ep, tmap);
  return hipGetLastError();
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
  tmap.krot = 0;
  tmap.trace = g_gemm_trace;
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
  tmap.krot = g_gemm_krot;
  tmap.trace = g_gemm_trace;
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
    case 6: return launch_cfg5<T, EPI, 160, 256, 2, 4, 5>(a, s);
    case 7: return launch_cfg5<T, EPI, 160, 256, 2, 4, 6>(a, s);
    case 10: return launch_cfg7<T, EPI, 160, 256, 2, 4, false>(a, s);
    case 11: return launch_cfg7<T, EPI, 160, 256, 2, 4, true>(a, s);
    case 8: if (EPI == EPI_F32_BIAS) return launch_cfg5<T, EPI_F32_BIAS, 160, 256, 2, 4, 61>(a, s); return hipErrorInvalidValue;
    case 9: if (EPI == EPI_F32_BIAS) return launch_cfg5<T, EPI_F32_BIAS, 160, 256, 2, 4, 62>(a, s); return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
}

int pick_variant(const GemmArgs& a) {
  if (g_gemm_variant >= 0) return g_gemm_variant;
  if (a.M <= 1024 || a.N < 256) return 0;
  return 11;
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
