// gemm.hip — C[M,N] = A[M,K] * W[N,K]^T on the gfx950 matrix cores (v_mfma_f32_16x16x32_{f16,bf16}).
//
// This kernel family carries 98.9 % of encode_image's FLOPs (SURVEY.md §8 A15a,d,f,g): conv1 as an
// im2col GEMM, QKV in-proj, attention out-proj, MLP c_fc (+QuickGELU) and c_proj (+residual).
// Both operands are K-contiguous ("B^T input"), which is exactly PyTorch's nn.Linear / Conv2d
// weight layout, so no weight transposition is needed at load time.
//
// Common design (wave64):
//   * Block tile BM x BN x 64; a wave owns a (BM/WM) x (BN/WN) sub-tile of 16x16 MFMA tiles.
//     The encoder's GEMMs have M = 12800 (= 256 crops x 50 tokens), N in {768, 2304, 3072}; with
//     256 CUs the tile shape decides the tail: 160x256 / 320x128 give 240 / 720 / 960 tiles (94 % of
//     whole CU rounds), 256x256 gives 150 / 450 / 600 (59 / 88 / 78 %).
//   * Global -> LDS with the LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no
//     staging VGPRs, no ds_write pass.
//   * LDS image: [rows][64] 16-bit = 128 B per row, 16-B chunk index XOR-swizzled with (row>>1)&7
//     (a 256-B bank row holds two tile rows): ds_read_b128 fragment reads are bank-conflict free
//     (SQ_LDS_BANK_CONFLICT = 0 measured).  The DMA writes LDS linearly, so the swizzle is applied
//     to the per-lane SOURCE address (same involution on the read side).
//   * MFMA operands are swapped (W fragment as the A operand, activation fragment as the B operand)
//     so a lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row.  For 16-bit
//     outputs the W rows of each pair of 16-column MFMA tiles are additionally interleaved in LDS
//     (sigma below) so that a lane's 8 values of a tile pair are 8 consecutive columns: one 16-B
//     store instead of two 8-B ones — epilogue stores are issue-bound (~300 cycles per store
//     instruction whatever its width), so halving their number halves the epilogue.
//   * Tile order: N is cut into panels of `pn` tile-columns, tiles are walked row-major inside a
//     panel, and each XCD (block b runs on XCD b % 8) gets a contiguous range of that order, so the
//     blocks resident on one XCD share a W panel and a few A rows in that XCD's 4 MiB L2 (the naive
//     order streamed all of W through every L2: 26 % L2 misses and a fabric-bound 280 TFLOP/s).
//
// Two kernels:
//   gemm_kernel     simple: one tile per block, 2-slot LDS ring, one barrier per K-tile.  Used for
//                   small problems (object-token stream, tiny models) and as the A/B baseline.
//   gemm_pp_kernel  production: persistent, ping-pong compute waves + dedicated DMA waves (below).
#include "common.h"
#include "kernels.h"

// -DOAKE_LAB=1 builds liboake_hip_lab.so: the production kernels PLUS every tile configuration, kernel form and
// measurement epilogue that lost its A/B (tools/, the variant tests).  The production library carries only what
// pick_variant() can select.
#ifndef OAKE_LAB
#define OAKE_LAB 0
#endif

namespace oake {

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
  const float2* rowstat;  // EPI_*_LN, simple kernel: per-row (rstd, -mean * rstd) (launch_rowstat)
  const float* colsum;    // EPI_*_LN: per-column sum of the (gamma-folded) weights
  // Row statistics carried between the persistent kernels instead of a separate pass over x:
  // the residual epilogue (EPI_RESID16) leaves (sum x, sum x^2) of every 64-column slice of a row in
  // rowpart_out[m][kRowParts] (slice = column / 64); the next LN-folded GEMM's DMA waves add up the
  // first `nparts` slices of its rows (fixed order) and put (rstd, -mean rstd) into LDS.
  // (Round 3 measured the slice-major layout [slice][m] — an MFMA row tile's 16 rows as ONE 128-byte line per wave
  // instead of 16 lines of 8 bytes, coalesced 8-byte loads on the consumer side: -1.0 % on the bench, five
  // interleaved rounds, profiles/r03/ab_rowpart_xl_a32.log.  The row-major form keeps a row's 12 slices in one
  // line, which the consumer fetches with six 16-byte loads per lane; it stays.)
  float2* rowpart_out;
  const float2* rowpart_in;
  int nparts;
  float inv_k;            // 1 / K
  int patch_S, patch_P, patch_G;  // A = NCHW image batch, gathered patch-wise (GemmArgs::patch_*); 0 = matrix
  int patch_T, patch_H;           // patch stride and rows per colour plane (0: = patch_P / patch_S, the unpadded form)
};
constexpr int kRowParts = 16;  // float2 slots per row (128 B): N <= 1024 residual width

// (sum x, sum x^2) of a row of K values -> LayerNorm's (rstd, -mean rstd), eps = 1e-5 [REF clip's LayerNorm = nn.LayerNorm
// evaluated in fp32].  One definition for both persistent kernels: their results must agree bit for bit.
__device__ __forceinline__ float2 ln_rowstat(float s1, float s2, float inv_k) {
  const float mean = s1 * inv_k;
  const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f);
  return make_float2(rstd, -mean * rstd);
}

struct TileMap {
  int tiles_m, tiles_n, pn, nwg;
  int by_m;  // 0: panels of pn tile-columns, row-major inside;  1: slabs of pn tile-rows, column-major inside
  unsigned long long* trace;  // [block < 64][group 2][tile < 8][4] cycle stamps, or nullptr
};

// Cache policy of the 16-byte tile stores.  1 (production) = sc1: write-through, the line is not kept in the XCD's
// L2.  A GEMM's output is read by the NEXT kernel, from any XCD, so it has to reach the fabric anyway; with plain
// (write-back) stores it does so at the kernel's end-of-kernel release — up to 20 MB of dirty lines that every CU
// waits for with nothing left to compute (ours 'bias' - 'none' epilogue = 4.4-5.2 us for 19.7 MB on the
// one-tile-per-CU shapes) — with write-through stores it leaves under the K loops that are still running.
// Measured (tools/ab_env.py, one session, four interleaved rounds, profiles/r03/ab_session_f_store_policy.log):
// plain 107.1 k images/s, sc1 109.2 k (+2.0 %; one lane +2.6 %), sc0 sc1 109.2 k, nt 106.8 k.
// Measurement builds of the other policies: OAKE_EXTRA_FLAGS=-DOAKE_STORE_POLICY=n with OAKE_LIB_OUT
// (0 = plain, 2 = nt, 3 = sc0 sc1).  The stores are inline asm (no builtin carries the cache bits for a flat
// global store): they end with the s_nop that keeps hipcc from overwriting the data registers early, and the
// "memory" clobber keeps the residual epilogue's loads of x behind them in program order.
#ifndef OAKE_STORE_POLICY
#define OAKE_STORE_POLICY 1
#endif
// ... for the stores that write FULL 128-byte lines (the lane-swapped tile stores of interior tiles: c_fc, out_proj,
// c_proj).  The epilogues that store in accumulator layout — 16 rows x 64 bytes per instruction: the trickled
// pieces of qkv, conv1's token-remapped rows, edge tiles — keep plain stores: the two half lines of a row are
// written by different instructions (or K-tiles apart) and only a write-back L2 merges them into one line
// (written through, they were -1.9 % in objects mode, where a kernel has 15 tiles per CU and its end-of-kernel
// release hardly matters: profiles/r03/ab_session_g_store_policy_objects.log).
// measurement switch: cache policy bits of the A operand's LDS-DMA pieces (2 = nt; activations are read by the few
// tiles of one N panel and never again, the W panel by every tile).  Measured: -8.6 % (102.4 vs 112.1 k images/s,
// profiles/r03/ab_session_n_gemm_a_operand_nt.log) — the panel's tiles do re-read their A rows from the L2.
#ifndef OAKE_GEMM_A_AUX
#define OAKE_GEMM_A_AUX 0
#endif
// ... and of the W operand's (round 4: with M slabs of 10 tile rows — OAKE_OPT_GEMM_PANEL = -10 — an XCD owns one A
// slab (2.46 MB at K = 768) and streams every W panel past it exactly once; W fetched non-temporal should leave the
// slab in the L2)
#ifndef OAKE_GEMM_W_AUX
#define OAKE_GEMM_W_AUX 0
#endif
// ... and of the A operand in the residual epilogue's GEMMs alone (out_proj, c_proj: N = 768 = ONE panel of three
// tiles, so an A row slab is read by three tiles that run side by side on one XCD and never again, while W — 1.2 / 4.7 MB
// — is what every tile of the XCD re-reads).  Measured with 2 = nt: -1.0 % objects, -1.8 % globals
// (profiles/r04/ab_session_resid_gemm_a_operand_nt.log): the three tiles are not in step closely enough for an
// evict-first line to survive until its last reader.  Off.
#ifndef OAKE_GEMM_A_AUX_RESID
#define OAKE_GEMM_A_AUX_RESID OAKE_GEMM_A_AUX
#endif
#ifndef OAKE_STORE_POLICY_PARTIAL
#define OAKE_STORE_POLICY_PARTIAL 0
#endif
// Measurement build (-DOAKE_TRICKLE_FULLLINE=1): the trickled pieces of the pure-store epilogues (qkv) lane-swapped
// into full lines at pack time and written through as well.  Neutral in all three modes (110.1 vs 109.9 k images/s,
// 82.0 vs 82.0 images/s objects, 3900 vs 3899 blocks: profiles/r03/ab_session_i_trickle_fullline_*.log): the 40 DPP
// moves per tile cost what the merged lines save; production keeps the accumulator layout + plain stores.
#ifndef OAKE_TRICKLE_FULLLINE
#define OAKE_TRICKLE_FULLLINE 0
#endif
// measurement switch: the residual tile (read once, by the one tile that owns it) with non-temporal loads
#ifndef OAKE_RESID_NT
#define OAKE_RESID_NT 0
#endif
// the persistent kernel's residual epilogue: x as the accumulators' initial values (1, production) or loaded and added at
// the tile end (0: the form of rounds 1-4, A/B builds)
#ifndef OAKE_RESID_INIT
#define OAKE_RESID_INIT 1
#endif
// Measurement builds of the persistent kernel's K loop (WRONG results; tools/kloop_ablate.sh, profiles/r06/kloop_ablation.md):
// which resource a K-tile's time follows.  Bits: 1 = half of the fragment reads (the odd fragments are copies of the even
// ones: 5 of 9 ds_read_b128 per kk), 2 = half of the LDS-DMA pieces (the odd pieces of a DMA wave are never issued: 7 of
// 13), 4 = half of the MFMAs (the odd column tiles are skipped), 8 = every LDS-DMA piece issued, but the odd pieces fetch
// the SAME source lines as their even neighbour (full request / LDS-write traffic, half of the distinct bytes: separates
// what the dropped pieces of form 2 save in data movement from what the stale operands save in MFMA operand toggling).
// Barriers, waits and phases are unchanged in every form.
#ifndef OAKE_KLOOP_ABLATE
#define OAKE_KLOOP_ABLATE 0
#endif
// c_fc (EPI_T16_GELU_LN) on the persistent kernel: 0 = two long phases per K-tile, the whole epilogue at the tile end;
// 1 = four phases, the tile packed as 16-bit PRE-activations and its QuickGELU + stores deferred into the load phases of
// the next tile's K loop (gemm_pp_kernel<..., DG = true>; lab variant 12 in either build).  A/B: profiles/r06.
#ifndef OAKE_DEFER_GELU
#define OAKE_DEFER_GELU 0
#endif
template <typename V>
__device__ __forceinline__ V resid_load16(const V* p) {
#if OAKE_RESID_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename P>
__device__ __forceinline__ void tile_store16(P* p, u32x4_t v) {  // full-line instruction
  store16_policy<OAKE_STORE_POLICY>(p, v);
}
template <typename P>
__device__ __forceinline__ void tile_store16_partial(P* p, u32x4_t v) {  // half lines per instruction
  store16_policy<OAKE_STORE_POLICY_PARTIAL>(p, v);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int EPI>
struct EpiTraits {
  // 16-bit outputs use the paired column mapping (see header comment)
  static constexpr bool kLn = (EPI == EPI_T16_BIAS_LN || EPI == EPI_T16_GELU_LN);
  static constexpr bool kGelu = (EPI == EPI_T16_GELU || EPI == EPI_T16_GELU_LN);
  // pure stores (no read-modify-write): the persistent kernel may hold them back and trickle them
  static constexpr bool kNone = (EPI == EPI_T16_NONE);   // measurement: no epilogue at all
  static constexpr bool kRaw = (EPI == EPI_T16_RAW);     // measurement: pack + store only
  static constexpr bool kTrickle = (EPI == EPI_T16_BIAS || EPI == EPI_T16_GELU || kLn || kRaw);
  static constexpr bool kPaired = (kTrickle || EPI == EPI_RESID16 || EPI == EPI_PATCH16 || kNone);
};

// acc -> acc + bias, or the folded LayerNorm  rstd * acc + (-mean rstd) * colsum + bias
template <bool LN>
__device__ __forceinline__ float epi_affine(float acc, float bias, float cs, float2 rs) {
  return LN ? fmaf(acc, rs.x, fmaf(rs.y, cs, bias)) : acc + bias;
}

// Column (relative to the wave's TN-column block) held by LDS row l = 16*ni + rho of that block.
// identity: l.   paired: tiles (2t, 2t+1) interleave 4-column groups so lane-group g of the MFMA
// output owns columns 32t + 8g .. +7  (rho = 4g + r:  tile 2t -> +r, tile 2t+1 -> +4 + r).
template <bool PAIRED>
__device__ __forceinline__ int sigma_col(int l) {
  if (!PAIRED) return l;
  const int ni = l >> 4, rho = l & 15;
  return 32 * (ni >> 1) + 8 * (rho >> 2) + 4 * (ni & 1) + (rho & 3);
}

// logical tile t -> tile origin.  The "outer" dimension (N for by_m = 0, M for by_m = 1) is cut into
// panels of pn tiles; inside a panel the outer index runs fastest.
__device__ __forceinline__ void tile_origin(const TileMap& tmap, int t, int BM, int BN, int& m0,
                                            int& n0) {
  const int outer = tmap.by_m ? tmap.tiles_m : tmap.tiles_n;
  const int inner = tmap.by_m ? tmap.tiles_n : tmap.tiles_m;
  const int full = outer / tmap.pn;
  const int per_panel = inner * tmap.pn;
  int panel, pw, rem;
  if (t < full * per_panel) {
    panel = t / per_panel;
    rem = t - panel * per_panel;
    pw = tmap.pn;
  } else {
    panel = full;
    rem = t - full * per_panel;
    pw = outer - full * tmap.pn;
  }
  const int ti = rem / pw;
  const int to = panel * tmap.pn + (rem - ti * pw);
  m0 = (tmap.by_m ? to : ti) * BM;
  n0 = (tmap.by_m ? ti : to) * BN;
}

// Per-lane source pointer of DMA piece `ii` (stage rows [8 ii, 8 ii + 8)): lane -> (row 8 ii +
// lane/8, LDS chunk lane%8) fetching source chunk (lane%8) ^ ((row>>1)&7) of A row m0+row or of the
// W row that sigma assigns to that LDS row.
//
// Patch-gather form (ep.patch_S != 0; conv1 without an im2col pass): A row m = (image, py, px) and a
// K-tile is 64 / P rows of P pixels of one channel, so chunk c (8 pixels) of the K-tile starting at k0
// lies at pixel row ky0 + (8 c) / P, column (8 c) % P of that patch — a per-lane part (here) plus a part
// that depends only on the K-tile (patch_koff), exactly like the matrix form's kt * 128 bytes.
template <typename T, int BM, int TN, bool PAIRED>
__device__ __forceinline__ const char* piece_src(const T* A, const T* W, int M, int N, int K, int m0,
                                                 int n0, int ii, int lane, int pS = 0, int pP = 0,
                                                 int pG = 0, int pT = 0, int pH = 0) {
  const int rr = 8 * ii + (lane >> 3);
  const int chunk = (lane & 7) ^ ((rr >> 1) & 7);
  if (rr < BM) {
    int gr = m0 + rr;
    gr = gr < M ? gr : M - 1;
    if (pS != 0) {
      const int img = gr / (pG * pG), p = gr - img * pG * pG;
      const int py = p / pG, px = p - py * pG;
      const int ky = (chunk * 8) / pP, kx = (chunk * 8) - ky * pP;
      // (pS = row stride, pH = rows per plane, pT = distance between patch origins: a zero-padded buffer lets
      // overlapping / offset patches — objects mode: stride 16, padding 15 — take the same path)
      return reinterpret_cast<const char*>(A + (size_t)img * 3 * pH * pS + (size_t)(py * pT + ky) * pS + px * pT + kx);
    }
    return reinterpret_cast<const char*>(A + (size_t)gr * K) + chunk * 16;
  }
  const int l = rr - BM;
  int gr = n0 + (l / TN) * TN + sigma_col<PAIRED>(l % TN);
  gr = gr < N ? gr : N - 1;
  return reinterpret_cast<const char*>(W + (size_t)gr * K) + chunk * 16;
}

// byte offset of K-tile kt in the patch-gather form: channel kt*64 / P^2, pixel row (kt*64 % P^2) / P
__device__ __forceinline__ size_t patch_koff(int kt, int pS, int pP, int pH) {
  const int k0 = kt * BK, pp = pP * pP;
  const int ch = k0 / pp, ky0 = (k0 - ch * pp) / pP;
  return ((size_t)ch * pH * pS + (size_t)ky0 * pS) * 2;
}

// Wave-level epilogue.  mbase = first row of the lane (m0 + wm*TM + lane&15), nwave = first column
// of the wave's block (n0 + wn*TN), g = lane>>4.  All bias / residual / pos-emb loads of a row are
// issued before its first store so the wave waits once per row; interior tiles take a branch-free
// path (with per-store exec-mask branches hipcc put an s_waitcnt vmcnt(0) in front of every store).
//
// ELDS (persistent kernel): bias / colsum / rowstat of the tile were staged into LDS by the DMA waves
// (EpiLds layout below) — `elds` points at that block, lcol / lrow are the lane's first column / row
// relative to the tile.  Otherwise they are read from global memory.
struct EpiLds {
  static constexpr int kBias = 0;        // float[BN <= 256]
  static constexpr int kColsum = 1024;   // float[BN <= 256]
  static constexpr int kRowstat = 2048;  // float2[BM <= 160]
  static constexpr int kBytes = 2048 + 160 * 8;
};

template <bool ELDS>
__device__ __forceinline__ float4 epi_vec4(const float* gptr, int n, const char* elds, int off, int lc) {
  if constexpr (ELDS)
    return *reinterpret_cast<const float4*>(elds + off + lc * 4);
  else
    return *reinterpret_cast<const float4*>(gptr + n);
}

// RIA ("residual in accumulators", EPI_RESID16 in the persistent kernel): the tile's x values were the accumulators'
// INITIAL values (tile_resid_init below), so the epilogue neither loads nor adds them
template <typename T, int EPI, int MI, int NI, bool FULL, bool ELDS, bool RIA = false>
__device__ __forceinline__ void tile_epilogue_impl(f32x4 (&acc)[MI][NI], int mbase, int nwave, int g,
                                                   int M, int N, const EpiParams& ep, bool reset,
                                                   const char* elds, int lcol, int lrow) {
  constexpr bool PAIRED = EpiTraits<EPI>::kPaired;
  if constexpr (EpiTraits<EPI>::kNone) {  // measurement: the MFMAs stay live, nothing is computed or stored
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
    return;
  }
  if (PAIRED) {
    static_assert(!PAIRED || NI % 2 == 0, "paired mapping needs an even number of column tiles");
    constexpr int NP = NI / 2;
    typedef typename T16<T>::vec8 vec8;
    constexpr bool LN = EpiTraits<EPI>::kLn;
    float4 b0[NP], b1[NP], c0[LN ? NP : 1], c1[LN ? NP : 1];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const int n = nwave + 32 * t + 8 * g;
      const bool ok = EPI != EPI_PATCH16 && ep.bias != nullptr && (FULL || n < N);
      const int lc = lcol + 32 * t + 8 * g;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      b0[t] = ok ? epi_vec4<ELDS>(ep.bias, n, elds, EpiLds::kBias, lc) : z4;
      b1[t] = ok ? epi_vec4<ELDS>(ep.bias, n + 4, elds, EpiLds::kBias, lc + 4) : z4;
      if constexpr (LN) {
        const bool okc = FULL || n < N;
        c0[t] = okc ? epi_vec4<ELDS>(ep.colsum, n, elds, EpiLds::kColsum, lc) : z4;
        c1[t] = okc ? epi_vec4<ELDS>(ep.colsum, n + 4, elds, EpiLds::kColsum, lc + 4) : z4;
      }
    }
    // Residual stream, RIA = false (the non-persistent kernels; -DOAKE_RESID_INIT=0): a row block's 16-B pieces are fetched
    // kAhead row blocks before they are used.
    // x is read and written through the same pointer, so hipcc keeps every load behind the stores
    // that precede it in program order: fetched row by row at the point of use, that was MI
    // dependent round trips to L2 per tile (10k cycles of epilogue); all MI rows up front would not fit
    // the persistent kernel's 168 registers next to the accumulators.
    constexpr int kAhead = 2;
    // Full-line accesses (interior tiles; tools/ubench/store_bench.hip): a lane's two 16-B pieces of
    // row r are the two halves of ONE 128-B line, so a store instruction in accumulator layout
    // writes 16 half lines — measured 5.6k cycles for the CU's 80 KiB tile-end burst against 3.8k
    // (all CUs; 1.6k with few) for instructions of 8 full lines.  Lanes r and r ^ 8 therefore swap
    // one piece each (one row_ror:8 DPP move per register, bank-masked so no selects): piece A goes
    // to / comes from row (r & 7), piece B row 8 + (r & 7), both at column 32 (r >> 3) + 8 g.
    constexpr bool SWAP = FULL && NP == 2 && EPI != EPI_PATCH16;
    const int frow_ = threadIdx.x & 15;
    const int swap_row = SWAP ? (frow_ & 7) - frow_ : 0;       // row of piece A relative to the lane's own
    const int swap_col = SWAP ? ((frow_ & 8) ? 32 : 0) : 0;    // column of both pieces relative to 8 g
    vec8 xres[EPI == EPI_RESID16 ? MI : 1][NP];
#define OAKE_FETCH_RESID(mi_)                                                                   \
  do {                                                                                          \
    _Pragma("unroll") for (int _t = 0; _t < NP; ++_t) {                                         \
      const int _m = mbase + (mi_) * 16 + (SWAP ? swap_row + 8 * _t : 0);                       \
      const int _n = nwave + 8 * g + (SWAP ? swap_col : 32 * _t);                               \
      if (FULL || (_m < M && _n < N))                                                           \
        xres[mi_][_t] = resid_load16(reinterpret_cast<const vec8*>(reinterpret_cast<const T*>(ep.out) + \
                                                                   (size_t)_m * ep.ldo + _n));  \
    }                                                                                           \
  } while (0)
    if constexpr (EPI == EPI_RESID16 && !RIA) {
#pragma unroll
      for (int mi = 0; mi < (kAhead < MI ? kAhead : MI); ++mi) OAKE_FETCH_RESID(mi);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mbase + mi * 16;
      const bool mok = FULL || m < M;
      float2 rs = make_float2(1.f, 0.f);
      if constexpr (LN) {
        if constexpr (ELDS)
          rs = *reinterpret_cast<const float2*>(elds + EpiLds::kRowstat + (lrow + mi * 16) * 8);
        else if (mok)
          rs = ep.rowstat[m];
      }
      size_t orow = (size_t)m;
      const float* posrow = nullptr;
      if (EPI == EPI_PATCH16) {
        const int img = m / ep.P2;
        const int p = m - img * ep.P2;
        orow = (size_t)img * ep.L + 1 + p;
        posrow = ep.pos + (size_t)(1 + p) * N;
      }
      T* orow_ptr = reinterpret_cast<T*>(ep.out) + orow * ep.ldo;
      float ps1 = 0.f, ps2 = 0.f;  // this lane's share of the row's (sum x, sum x^2)
      // this row's residual (8 x 16-bit) / pos-emb (8 x fp32) loads first, then the stores
      vec8 xr[NP];
      u32x4_t qv[SWAP ? NP : 1];
      float4 p0[NP], p1[NP];
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const int n = nwave + 32 * t + 8 * g;
        const bool ok = FULL || (mok && n < N);
        if (EPI == EPI_RESID16 && !RIA) {
          // own piece t: fetched by this lane (A for rows < 8, B for rows >= 8) or by lane r ^ 8
          const vec8 own = xres[EPI == EPI_RESID16 ? mi : 0][t];
          xr[t] = SWAP ? swap_piece(own, xres[EPI == EPI_RESID16 ? mi : 0][NP - 1 - t], t == 0) : own;
        } else if (EPI == EPI_PATCH16) {
          p0[t] = ok ? *reinterpret_cast<const float4*>(posrow + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          p1[t] = ok ? *reinterpret_cast<const float4*>(posrow + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const int n = nwave + 32 * t + 8 * g;
        f32x4 lo = acc[mi][2 * t], hi = acc[mi][2 * t + 1];
        if (reset) {
          acc[mi][2 * t] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[mi][2 * t + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!FULL && !(mok && n < N)) continue;
        {
          const float4 cl = c0[LN ? t : 0], ch = c1[LN ? t : 0];
          lo[0] = epi_affine<LN>(lo[0], b0[t].x, cl.x, rs); lo[1] = epi_affine<LN>(lo[1], b0[t].y, cl.y, rs);
          lo[2] = epi_affine<LN>(lo[2], b0[t].z, cl.z, rs); lo[3] = epi_affine<LN>(lo[3], b0[t].w, cl.w, rs);
          hi[0] = epi_affine<LN>(hi[0], b1[t].x, ch.x, rs); hi[1] = epi_affine<LN>(hi[1], b1[t].y, ch.y, rs);
          hi[2] = epi_affine<LN>(hi[2], b1[t].z, ch.z, rs); hi[3] = epi_affine<LN>(hi[3], b1[t].w, ch.w, rs);
        }
        if (EpiTraits<EPI>::kGelu) {
          quick_gelu4(lo);
          quick_gelu4(hi);
        } else if (EPI == EPI_RESID16 && !RIA) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lo[r] += to32<T>(xr[t][r]);
            hi[r] += to32<T>(xr[t][4 + r]);
          }
        } else if (EPI == EPI_PATCH16) {
          lo[0] += p0[t].x; lo[1] += p0[t].y; lo[2] += p0[t].z; lo[3] += p0[t].w;
          hi[0] += p1[t].x; hi[1] += p1[t].y; hi[2] += p1[t].z; hi[3] += p1[t].w;
        }
        if constexpr (ELDS && EPI == EPI_RESID16) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ps1 += lo[r] + hi[r];
            ps2 = fmaf(lo[r], lo[r], fmaf(hi[r], hi[r], ps2));
          }
        }
        const uint2 q0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]);
        const uint2 q1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
        if constexpr (SWAP)
          qv[t] = u32x4_t{q0.x, q0.y, q1.x, q1.y};
        else
          tile_store16_partial(orow_ptr + n, u32x4_t{q0.x, q0.y, q1.x, q1.y});
      }
      if constexpr (SWAP) {
        T* pa = orow_ptr + (ptrdiff_t)swap_row * ep.ldo + nwave + 8 * g + swap_col;
        tile_store16(pa, swap_piece(qv[0], qv[1], true));
        tile_store16(pa + (size_t)8 * ep.ldo, swap_piece(qv[1], qv[0], false));
      }
      if constexpr (EPI == EPI_RESID16 && !RIA) {
        if (mi + kAhead < MI) OAKE_FETCH_RESID(mi + kAhead < MI ? mi + kAhead : 0);
      }
      if constexpr (ELDS && EPI == EPI_RESID16) {
        if (ep.rowpart_out != nullptr) {  // (uniform) the wave's 64-column slice of row m
          static_assert(NI * 16 == 64, "row-statistics slices are 64 columns wide");
          ps1 = rows16_sum(ps1);
          ps2 = rows16_sum(ps2);
          if (g == 0 && mok) ep.rowpart_out[(size_t)m * kRowParts + nwave / 64] = make_float2(ps1, ps2);
        }
      }
    }
#undef OAKE_FETCH_RESID
    return;
  }
  // fp32 outputs: lane owns columns nwave + 16 ni + 4 g .. +3 of each column tile
  const int nbase = nwave + 4 * g;
  float4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = nbase + ni * 16;
    bv[ni] = (EPI != EPI_PATCH && ep.bias != nullptr && (FULL || n < N))
                 ? epi_vec4<ELDS>(ep.bias, n, elds, EpiLds::kBias, lcol + 4 * g + ni * 16)
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
      float* o = reinterpret_cast<float*>(ep.out) + orow * ep.ldo + n;
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// Paired (16-bit) epilogue of an interior tile, split in two: pack now, store later.  The
// persistent kernel keeps the packed tile (MI * NI/2 x 16 B per lane) in registers and issues ONE
// store per K-tile of the NEXT tile: all blocks reach their tile ends together, so storing at once
// is a chip-wide write burst (20 MB at ~5.7 TB/s = 3.5 us with every compute wave stalled in store
// issue); trickled, the same bytes ride under the next tile's MFMAs at ~2 TB/s.
// DG ("deferred GELU", c_fc): the pending rows are packed as PRE-activations — the LayerNorm affine rounded to 16 bits,
// which is what the reference's fp16 Linear hands its QuickGELU [REF oadp/oake/globals.py:57 -> clip's ResidualAttentionBlock
// .mlp: c_fc -> QuickGELU on the fp16 tensor] — and the activation itself is applied piece by piece in the load phases
// of the NEXT tile's K loop (gelu_piece_half below), where a compute wave otherwise waits at the barrier for its
// partner's MFMA phase; the rows stored at the tile end (mi < MI0) get it here.
template <typename T, int EPI, int MI, int NI, int MI0, bool DG = false>
__device__ __forceinline__ void tile_pack_paired(f32x4 (&acc)[MI][NI], uint4 (&pend)[MI - MI0][NI / 2],
                                                 const EpiParams& ep, T* row0_ptr, const char* elds,
                                                 int lcol, int lrow) {
  constexpr int NP = NI / 2;
  constexpr bool LN = EpiTraits<EPI>::kLn;
  // Epilogue constants come from LDS (staged by the DMA waves): no global-load latency here, and no
  // VMEM results pending at the K-loop header (hipcc would guard the loop-top ds_reads with
  // s_waitcnt vmcnt(0) on every iteration, which then waits for the previous trickled store).
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    const int lc = lcol + 32 * t;
    const bool ok = ep.bias != nullptr && !EpiTraits<EPI>::kRaw;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b0 = ok ? epi_vec4<true>(nullptr, 0, elds, EpiLds::kBias, lc) : z4;
    const float4 b1 = ok ? epi_vec4<true>(nullptr, 0, elds, EpiLds::kBias, lc + 4) : z4;
    const float4 cl = LN ? epi_vec4<true>(nullptr, 0, elds, EpiLds::kColsum, lc) : z4;
    const float4 ch = LN ? epi_vec4<true>(nullptr, 0, elds, EpiLds::kColsum, lc + 4) : z4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f32x4 lo = acc[mi][2 * t], hi = acc[mi][2 * t + 1];
      float2 r = make_float2(1.f, 0.f);
      if constexpr (LN) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(3))) f32x2* lds_f2_t;
        const f32x2 rv = *(lds_f2_t)(elds + EpiLds::kRowstat + (lrow + mi * 16) * 8);
        r = make_float2(rv[0], rv[1]);
      }
      if constexpr (!EpiTraits<EPI>::kRaw) {
        lo[0] = epi_affine<LN>(lo[0], b0.x, cl.x, r); lo[1] = epi_affine<LN>(lo[1], b0.y, cl.y, r);
        lo[2] = epi_affine<LN>(lo[2], b0.z, cl.z, r); lo[3] = epi_affine<LN>(lo[3], b0.w, cl.w, r);
        hi[0] = epi_affine<LN>(hi[0], b1.x, ch.x, r); hi[1] = epi_affine<LN>(hi[1], b1.y, ch.y, r);
        hi[2] = epi_affine<LN>(hi[2], b1.z, ch.z, r); hi[3] = epi_affine<LN>(hi[3], b1.w, ch.w, r);
      }
      if (EpiTraits<EPI>::kGelu && (!DG || mi < MI0)) {
        if constexpr (DG) {  // (the same two roundings as the deferred rows: pre-activation to 16 bits, then the activation)
          const uint2 q0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]), q1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
          const typename T16<T>::vec4 v0 = __builtin_bit_cast(typename T16<T>::vec4, q0),
                                      v1 = __builtin_bit_cast(typename T16<T>::vec4, q1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lo[r] = to32<T>(v0[r]);
            hi[r] = to32<T>(v1[r]);
          }
        }
        quick_gelu4(lo);
        quick_gelu4(hi);
      }
      const uint2 p0 = pack4<T>(lo[0], lo[1], lo[2], lo[3]);
      const uint2 p1 = pack4<T>(hi[0], hi[1], hi[2], hi[3]);
      const uint4 pk = make_uint4(p0.x, p0.y, p1.x, p1.y);
      if (mi < MI0)  // the 168-VGPR budget (3 waves per SIMD) holds MI - MI0 rows; the rest go now
        tile_store16_partial(row0_ptr + (size_t)mi * 16 * ep.ldo + t * 32, as_u32x4(pk));
      else
        pend[mi - MI0][t] = pk;
    }
  }
#if OAKE_TRICKLE_FULLLINE
  // the pending pieces leave as FULL lines (and can then be written through, see OAKE_STORE_POLICY): lanes r and
  // r ^ 8 swap one piece each, exactly as the lane-swapped stores of tile_epilogue_impl — piece [mi][0] becomes
  // (row 16 mi + (r & 7), columns 32 (r >> 3) + 8 g ..), piece [mi][1] the same columns 8 rows further down
  static_assert(NP == 2, "piece swap pairs the two 32-column halves of a wave's 64 columns");
#pragma unroll
  for (int mi = MI0; mi < MI; ++mi) {
    const u32x4_t a = as_u32x4(pend[mi - MI0][0]), b = as_u32x4(pend[mi - MI0][1]);
    const u32x4_t sa = swap_piece(a, b, true), sb = swap_piece(b, a, false);
    pend[mi - MI0][0] = make_uint4(sa[0], sa[1], sa[2], sa[3]);
    pend[mi - MI0][1] = make_uint4(sb[0], sb[1], sb[2], sb[3]);
  }
#endif
}

// QuickGELU of four packed 16-bit values (two registers of a pending piece), in place: unpack -> fp32 quick_gelu4 -> round
template <typename T>
__device__ __forceinline__ void gelu_packed4(unsigned& r0, unsigned& r1) {
  const typename T16<T>::vec4 v = __builtin_bit_cast(typename T16<T>::vec4, make_uint2(r0, r1));
  f32x4 f = f32x4{to32<T>(v[0]), to32<T>(v[1]), to32<T>(v[2]), to32<T>(v[3])};
  quick_gelu4(f);
  const uint2 q = pack4<T>(f[0], f[1], f[2], f[3]);
  r0 = q.x;
  r1 = q.y;
}

template <typename T, int EPI, int MI, int NI>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[MI][NI], int mbase, int nwave, int g,
                                              int M, int N, const EpiParams& ep, bool reset,
                                              bool interior) {
  if (interior)
    tile_epilogue_impl<T, EPI, MI, NI, true, false>(acc, mbase, nwave, g, M, N, ep, reset, nullptr, 0, 0);
  else
    tile_epilogue_impl<T, EPI, MI, NI, false, false>(acc, mbase, nwave, g, M, N, ep, reset, nullptr, 0, 0);
}

template <typename T, int EPI, int MI, int NI, bool RIA = false>
__device__ __forceinline__ void tile_epilogue_lds(f32x4 (&acc)[MI][NI], int mbase, int nwave, int g,
                                                  int M, int N, const EpiParams& ep, bool reset,
                                                  bool interior, const char* elds, int lcol, int lrow) {
  if (interior)
    tile_epilogue_impl<T, EPI, MI, NI, true, true, RIA>(acc, mbase, nwave, g, M, N, ep, reset, elds, lcol, lrow);
  else
    tile_epilogue_impl<T, EPI, MI, NI, false, true, RIA>(acc, mbase, nwave, g, M, N, ep, reset, elds, lcol, lrow);
}

// EPI_RESID16 in the persistent kernel: x + (A W^T + bias) with the tile's x values as the accumulators' INITIAL values.
// The ten 16-byte loads per lane are issued where a compute wave has nothing else to do — at kernel entry, while the DMA
// waves fill the ring (4-6 k cycles), or right behind the previous tile's epilogue — instead of at the tile end, where they
// were three dependent round trips to the Infinity Cache (x was last touched a GEMM ago) in front of every wave's stores.
// Paired column mapping as the epilogue's: lane (frow, g) holds x[mbase + 16 mi][nwave + 32 t + 8 g .. + 7] in
// acc[mi][2 t] (first four) and acc[mi][2 t + 1].  fp32 accumulation starts from x instead of ending with it: the sums
// differ from the epilogue-add form by fp32 rounding only (both round to 16 bits once, at the store).
template <typename T, int MI, int NI>
__device__ __forceinline__ void tile_resid_init(f32x4 (&acc)[MI][NI], int mbase, int nwave, int g, int M, int N,
                                                const EpiParams& ep, bool interior) {
  typedef typename T16<T>::vec8 vec8;
  constexpr int NP = NI / 2;
  vec8 x[MI][NP];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const int m = mbase + mi * 16, n = nwave + 32 * t + 8 * g;
      vec8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = to16<T>(0.f);
      if (interior || (m < M && n < N))
        v = resid_load16(reinterpret_cast<const vec8*>(reinterpret_cast<const T*>(ep.out) + (size_t)m * ep.ldo + n));
      x[mi][t] = v;
    }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[mi][2 * t][r] = to32<T>(x[mi][t][r]);
        acc[mi][2 * t + 1][r] = to32<T>(x[mi][t][4 + r]);
      }
}

// ---------------------------------------------------------------------------------------------
// Simple kernel: one tile per block, WM x WN waves, 2-slot ring, one barrier per K-tile; the DMA of
// K-tile t+1 is in flight under the MFMAs of K-tile t.
template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const T* __restrict__ A,
                                                           const T* __restrict__ W, int M, int N,
                                                           int K, EpiParams ep, TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr bool PAIRED = EpiTraits<EPI>::kPaired;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;          // 1-KiB DMA pieces per stage
  constexpr int NSLOT = (NINST + NW - 1) / NW;  // pieces per wave (the last may be idle)
  static_assert(TM % 16 == 0 && TN % 32 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  int m0, n0;
  tile_origin(tmap, xcd_remap(blockIdx.x, tmap.nwg), BM, BN, m0, n0);

  const char* src[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j)
    src[j] = piece_src<T, BM, TN, PAIRED>(A, W, M, N, K, m0, n0, wid + NW * j, lane);
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

  // fragment reads: lane -> (row = lane&15, k-chunk = lane>>4)
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
  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN, fg, M, N, ep, false,
                                m0 + BM <= M && n0 + BN <= N);
}

// ---------------------------------------------------------------------------------------------
// Small-problem kernel: gemm_kernel with a 4-slot ring and counted waits.  With a few hundred rows
// (the CLS rows of the last block, the object-token stream, the head projection) a launch is a
// handful of 64x64 tiles per CU and its time is nk x (DMA latency): gemm_kernel's barrier retires the
// NEXT K-tile's DMA every iteration (vmcnt(0)), so one L2 round trip is exposed per K-tile.  Here
// three K-tiles are in flight and an iteration waits only for the oldest (s_waitcnt vmcnt(2 x pieces)),
// raw s_barrier instead of __syncthreads (whose fence would wait for everything).
template <typename T, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_deep_kernel(const T* __restrict__ A,
                                                                const T* __restrict__ W, int M, int N,
                                                                int K, EpiParams ep, TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr bool PAIRED = EpiTraits<EPI>::kPaired;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;  // 1-KiB DMA pieces per stage
  static_assert(NINST % NW == 0, "every wave issues the same number of pieces (counted vmcnt)");
  constexpr int NSLOT = NINST / NW;
  constexpr int NSTAGE = 4;
  static_assert(2 * NSLOT <= 63, "vmcnt immediate");
  static_assert(TM % 16 == 0 && TN % 32 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  int m0, n0;
  tile_origin(tmap, xcd_remap(blockIdx.x, tmap.nwg), BM, BN, m0, n0);

  const char* src[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j)
    src[j] = piece_src<T, BM, TN, PAIRED>(A, W, M, N, K, m0, n0, wid + NW * j, lane);
#define OAKE_DEEP_STAGE(kt_)                                                                   \
  do {                                                                                         \
    char* _base = smem + ((kt_) % NSTAGE) * kStageBytes;                                       \
    const size_t _koff = (size_t)(kt_) * (BK * 2);                                             \
    _Pragma("unroll") for (int _j = 0; _j < NSLOT; ++_j)                                       \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[_j] + _koff),                         \
                                         (lds_ptr_t)(_base + (wid + NW * _j) * 1024), 16, 0, 0); \
  } while (0)
  // vmcnt immediate: bits [3:0] | [15:14]; expcnt 7 = "don't wait"; lgkmcnt 0 (own LDS reads done)
#define OAKE_DEEP_WAIT(n_) __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15) | (((n_) >> 4) << 14))

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

  const int nk = K / BK;
  OAKE_DEEP_STAGE(0);
  if (nk > 1) OAKE_DEEP_STAGE(1);
  if (nk > 2) OAKE_DEEP_STAGE(2);
  for (int kt = 0; kt < nk; ++kt) {
    // K-tile kt has landed when at most the pieces of the (up to two) younger K-tiles are in flight
    const int younger = nk - 1 - kt;
    if (younger >= 2) OAKE_DEEP_WAIT(2 * NSLOT);
    else if (younger == 1) OAKE_DEEP_WAIT(NSLOT);
    else OAKE_DEEP_WAIT(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // ... in every wave; and everybody is done with K-tile kt - 1's slot
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 3 < nk) OAKE_DEEP_STAGE(kt + 3);
    const char* st = smem + (kt % NSTAGE) * kStageBytes;
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
  }
#undef OAKE_DEEP_STAGE
#undef OAKE_DEEP_WAIT
  tile_epilogue<T, EPI, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN, fg, M, N, ep, false,
                                m0 + BM <= M && n0 + BN <= N);
}

#if OAKE_LAB
#include "gemm_lab_q4.inc"
#endif

// ---------------------------------------------------------------------------------------------
// Production kernel: persistent, ping-pong compute waves + dedicated DMA waves.
//
// What the cycle traces of the simple kernel showed (s_memtime stamps per K-tile, 160x256 tile):
// 1280 cycles of matrix work per SIMD per K-tile, but 2250-2460 cycles per K-tile, because
//   (1) an LDS-DMA piece stalls its issuing wave ~80 cycles and a wave's 9 fragment reads ~250, and
//       the two waves that share a SIMD ran the SAME phase at the same time (the per-K-tile barrier
//       re-aligns them), so those stalls were never covered by the partner's MFMAs;
//   (2) per tile, ~6.7k cycles of prologue (first DMA latency) and 7-10k cycles of epilogue (store
//       issue) were exposed — with K = 768 (12 K-tiles) almost half of a tile's time.
// Structure:
//   * 12 waves: 8 compute waves in two groups of four (one wave of each group per SIMD) + 4 DMA
//     waves (one per SIMD, no accumulators) that issue ALL LDS-DMA pieces and do the counted vmcnt
//     waits.  Compute waves only read fragments and issue MFMAs.
//   * every compute wave runs the same 4-phase loop per K-tile —
//         LOAD0 (read kk0 fragments) | MFMA0 (20 MFMAs) | LOAD1 (read kk1 fragments) | MFMA1
//     with a barrier after each phase — but group 1 executes ONE extra barrier before the loop and
//     group 0 one after it, so the two waves of a SIMD are always one phase apart: while one issues
//     nothing but MFMAs the other does its LDS reads.
//   * 3-slot LDS ring, DMA two K-tiles ahead: the barrier that ends a DMA wave's 4th phase of K-tile
//     g publishes K-tile g+1 (counted vmcnt leaves K-tile g+2 in flight) and frees K-tile g's slot.
//   * persistent: one block per CU walks its tiles (XCD-contiguous order); the K-tile pipeline runs
//     straight across tile boundaries, so there is no per-tile prologue and the epilogue's stores
//     drain under the next tile's MFMAs.
// Measured: 1520 cycles per K-tile in the loop = 84 % MFMA utilisation.
template <typename T, int EPI, int BM, int BN, int WM, int WN, bool PH2 = false, bool A32 = false, bool DG = false>
__global__ __launch_bounds__((WM * WN + 4) * 64) void gemm_pp_kernel(const T* __restrict__ A,
                                                                     const T* __restrict__ W, int M,
                                                                     int N, int K, EpiParams ep,
                                                                     TileMap tmap) {
  typedef typename T16<T>::vec8 vec8;
  constexpr bool PAIRED = EpiTraits<EPI>::kPaired;
  constexpr int NW = WM * WN;
  static_assert(NW == 8, "two compute groups of four waves");
  constexpr int NL = 4;  // DMA waves
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int kATileBytes = BM * kRowBytes;
  constexpr int kStageBytes = (BM + BN) * kRowBytes;
  constexpr int NINST = (BM + BN) / 8;
  static_assert(NINST % NL == 0, "pieces must split evenly over the DMA waves");
  constexpr int NPL = NINST / NL;  // pieces per DMA wave per K-tile
  static_assert(NPL <= 31, "vmcnt immediate");
  static_assert(TM % 16 == 0 && TN % 32 == 0, "tile shape");
  constexpr int NSTAGE = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // this block's tile list: XCD x owns logical tiles [xb, xb + xc); block (b/8) of that XCD takes
  // xb + b/8 + i * (blocks per XCD)
  const int nx = 8;
  const int xcd = blockIdx.x % nx, xslot = blockIdx.x / nx;
  const int per_xcd = gridDim.x / nx;  // host launches a multiple of 8 blocks
  const int q = tmap.nwg / nx, r = tmap.nwg % nx;
  const int xb = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xc = xcd < r ? q + 1 : q;
  const int my_tiles = xslot < xc ? (xc - xslot + per_xcd - 1) / per_xcd : 0;
  if (my_tiles == 0) return;
  const int nk = K / BK;
  const int total = my_tiles * nk;  // flat K-tile count of this block
#define OAKE_PIN() __builtin_amdgcn_sched_barrier(0)
#define OAKE_BAR()                   \
  do {                               \
    OAKE_PIN();                      \
    __builtin_amdgcn_s_barrier();    \
    OAKE_PIN();                      \
  } while (0)

  if (wid >= NW) {
    // ================= DMA wave =================
    const int lw = wid - NW;
    const char* src[NPL];
    auto set_src = [&](int tile_i) {
      int m0, n0;
      tile_origin(tmap, xb + xslot + tile_i * per_xcd, BM, BN, m0, n0);
#pragma unroll
      for (int j = 0; j < NPL; ++j)
        if (!A32 || j >= BM / 8 / NL)  // (A32: the A rows go through registers, a32_set_src)
          src[j] = piece_src<T, BM, TN, PAIRED>(A, W, M, N, K, m0, n0, lw + NL * j, lane, ep.patch_S,
                                                ep.patch_P, ep.patch_G, ep.patch_T, ep.patch_H);
    };
    // pieces lw + NL j with j < kAPieces are A rows for every DMA wave (BM / 8 is a multiple of NL)
    static_assert((BM / 8) % NL == 0, "A pieces split evenly over the DMA waves");
    constexpr int kAPieces = BM / 8 / NL;
    // A32 (conv1 on an fp32 NCHW batch, patch 32): the A half of a stage does not come by LDS-DMA (a raw copy) but
    // through this wave's registers.  A K-tile is two pixel rows ("runs") of 32 floats = one 128-byte line each per
    // tile row; piece j = 8 tile rows, lane (row = lane >> 3, c = lane & 7) fetches floats [4c, 4c+4) of run 0 and of
    // run 1 — two instructions of 8 FULL lines each (fetching 8 consecutive floats per lane instead touches 16 half
    // lines per instruction, which the vector memory path serves at half the rate).  Adjacent lanes then swap one
    // float4 (quad_perm DPP): the even lane ends up with 8 consecutive floats of run 0, the odd lane with 8 of run
    // 1, i.e. one 16-byte chunk of the 16-bit LDS image each; rounded to T (RNE, as im2col's cast) and written with
    // ds_write_b128 into the same swizzled image the LDS-DMA would have produced.  The rows of flat K-tile g+2 are
    // requested in iteration g and written at the top of iteration g+1 (one register set; a second set for two
    // K-tiles of look-ahead does not fit the 168-register budget: 152 bytes of scratch).
    static_assert(!A32 || (PH2 && EPI == EPI_PATCH16 && sizeof(T) == 2), "A32: conv1 on the long-phase kernel");
    const float* asrc[A32 ? kAPieces : 1];  // lane's float4 of run 0 of piece j at K-tile 0
    int adst[A32 ? kAPieces : 1];           // byte offset of the lane's 16-byte chunk inside a stage
    float4 areg[A32 ? 2 * kAPieces : 1];
    int a_buf = 0;                          // ring slot the next write fills
    auto a32_set_src = [&](int tile_i) {
      if constexpr (A32) {
        int m0, n0;
        tile_origin(tmap, xb + xslot + tile_i * per_xcd, BM, BN, m0, n0);
        const float* A32p = reinterpret_cast<const float*>(A);
        const int pS = ep.patch_S, pP = ep.patch_P, pG = ep.patch_G;
#pragma unroll
        for (int j = 0; j < kAPieces; ++j) {
          const int rr = 8 * (lw + NL * j) + (lane >> 3), c = lane & 7;
          int gr = m0 + rr;
          gr = gr < M ? gr : M - 1;
          const int img = gr / (pG * pG), p = gr - img * pG * pG;
          const int py = p / pG, px = p - py * pG;
          asrc[j] = A32p + (size_t)img * 3 * pS * pS + (size_t)(py * pP) * pS + px * pP + 4 * c;
          const int cc = (c & 1) * 4 + (c >> 1);  // the K-tile's 16-byte chunk this lane assembles
          adst[j] = rr * kRowBytes + ((cc ^ ((rr >> 1) & 7)) << 4);
        }
      }
    };
    auto a32_load = [&](int kt) {  // K-tile kt: channel kt / 16, pixel rows 2 (kt % 16), 2 (kt % 16) + 1 of the patch
      if constexpr (A32) {
        const int k0 = kt * BK, pp = ep.patch_P * ep.patch_P;
        const int ch = k0 / pp, ky0 = (k0 - ch * pp) / ep.patch_P;
        const size_t off = ((size_t)ch * ep.patch_S + ky0) * ep.patch_S;
#pragma unroll
        for (int j = 0; j < kAPieces; ++j) {
          areg[2 * j] = *reinterpret_cast<const float4*>(asrc[j] + off);
          areg[2 * j + 1] = *reinterpret_cast<const float4*>(asrc[j] + off + ep.patch_S);
        }
      }
    };
    auto a32_write = [&]() {
      if constexpr (A32) {
        typedef typename T16<T>::vec8 vec8w;
        char* base = smem + a_buf * kStageBytes;
        const bool odd = lane & 1;
#pragma unroll
        for (int j = 0; j < kAPieces; ++j) {
          const float4 r0 = areg[2 * j], r1 = areg[2 * j + 1];
          const float4 send = odd ? r0 : r1;  // what the neighbour needs
          float4 recv;
          recv.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.x), 0xB1, 0xF, 0xF, true));
          recv.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.y), 0xB1, 0xF, 0xF, true));
          recv.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.z), 0xB1, 0xF, 0xF, true));
          recv.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send.w), 0xB1, 0xF, 0xF, true));
          const float4 lo = odd ? recv : r0, hi = odd ? r1 : recv;
          vec8w v;
          v[0] = to16<T>(lo.x); v[1] = to16<T>(lo.y); v[2] = to16<T>(lo.z); v[3] = to16<T>(lo.w);
          v[4] = to16<T>(hi.x); v[5] = to16<T>(hi.y); v[6] = to16<T>(hi.z); v[7] = to16<T>(hi.w);
          *reinterpret_cast<vec8w*>(base + adst[j]) = v;
        }
        a_buf = a_buf == NSTAGE - 1 ? 0 : a_buf + 1;
      }
    };
    // producer cursor: flat K-tile s_g (k position s_kt of tile s_tile) goes to ring slot s_buf
    int s_g = 0, s_kt = 0, s_tile = 0, s_buf = 0;
    // consumer position (compute group 0): K-tile d_kt of tile d_tile.  In the last phase of a tile's
    // FIRST K-tile (group 1 finished the previous tile's epilogue two phases earlier) one DMA wave
    // each stages the tile's bias / colsum block into the EpiLds area; it is covered by
    // the next iteration's vmcnt wait + barrier, long before the tile's epilogue (nk >= 3).
    // (Two long phases per K-tile: group 1's epilogue of the previous tile runs IN the last phase of the
    // first K-tile, so the block is staged one K-tile later.)
    constexpr int kEpiStageKt = PH2 ? 1 : 0;
    int d_kt = 0, d_tile = 0;
#define OAKE_STAGE_EPI()                                                                        \
  do {                                                                                          \
    if (d_kt == kEpiStageKt) {                                                                  \
      int _m0, _n0;                                                                             \
      tile_origin(tmap, xb + xslot + d_tile * per_xcd, BM, BN, _m0, _n0);                       \
      char* _e = smem + NSTAGE * kStageBytes;                                                   \
      int _n = _n0 + 4 * lane;                                                                  \
      _n = _n + 4 <= N ? _n : N - 4;                                                            \
      if (lw == 0 && ep.bias != nullptr && EPI != EPI_PATCH && EPI != EPI_PATCH16)             \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ep.bias + _n),                             \
                                         (lds_ptr_t)(_e + EpiLds::kBias), 16, 0, 0);            \
      if (EpiTraits<EPI>::kLn && lw == 1)                                                       \
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ep.colsum + _n),                           \
                                         (lds_ptr_t)(_e + EpiLds::kColsum), 16, 0, 0);          \
    }                                                                                           \
    if (++d_kt == nk) {                                                                         \
      d_kt = 0;                                                                                 \
      ++d_tile;                                                                                 \
    }                                                                                           \
  } while (0)
// (form 8: piece j reads what piece j & ~1 reads — never across the A / W boundary: kAPieces = 5 is odd, piece 4 | 5 differ)
#define OAKE_ABL_SRC(j_) (((OAKE_KLOOP_ABLATE & 8) && ((j_) & 1) && (((j_) - 1 < kAPieces) == ((j_) < kAPieces))) ? (j_) - 1 : (j_))
#define OAKE_STAGE(j0_, j1_)                                                                 \
  do {                                                                                       \
    if (s_g < total) {                                                                       \
      char* _base = smem + s_buf * kStageBytes;                                              \
      const size_t _koff = (size_t)s_kt * (BK * 2);                                          \
      const size_t _koffa = ep.patch_S != 0 ? patch_koff(s_kt, ep.patch_S, ep.patch_P, ep.patch_H) : _koff; \
      _Pragma("unroll") for (int _j = (j0_); _j < (j1_); ++_j)                               \
          if ((!A32 || _j >= kAPieces) && !((OAKE_KLOOP_ABLATE & 2) && !A32 && (_j & 1))) {  \
            if (_j < kAPieces)                                                               \
              __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[OAKE_ABL_SRC(_j)] + _koffa),  \
                                               (lds_ptr_t)(_base + (lw + NL * _j) * 1024), 16, 0,           \
                                               EPI == EPI_RESID16 ? OAKE_GEMM_A_AUX_RESID : OAKE_GEMM_A_AUX); \
            else                                                                             \
              __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[OAKE_ABL_SRC(_j)] + _koff),   \
                                               (lds_ptr_t)(_base + (lw + NL * _j) * 1024), 16, 0, OAKE_GEMM_W_AUX); \
          }                                                                                  \
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
        if (s_tile < my_tiles) {                          \
          set_src(s_tile);                                \
          a32_set_src(s_tile);                            \
        }                                                 \
      }                                                   \
    }                                                     \
  } while (0)
    // vmcnt immediate: bits [3:0] | [15:14]; expcnt 7 and lgkmcnt 15 = "don't wait"
#define OAKE_VMCNT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
    constexpr int Q1 = (NPL + 3) / 4, Q2 = (2 * NPL + 3) / 4, Q3 = (3 * NPL + 3) / 4;
    // vmcnt entries one K-tile's worth of this wave's requests occupies
    constexpr int kPerKt = A32 ? (NPL - kAPieces) + 2 * kAPieces : (OAKE_KLOOP_ABLATE & 2) ? (NPL + 1) / 2 : NPL;
    static_assert(kPerKt <= 63, "vmcnt immediate");
    set_src(0);
    a32_set_src(0);
    if constexpr (A32) a32_load(0);  // A rows of flat K-tile 0 ...
    OAKE_STAGE(0, NPL);
    if constexpr (A32) a32_write();  // ... into slot 0 (hipcc waits for the loads here)
    OAKE_ADVANCE();
    if constexpr (A32) a32_load(s_kt);  // flat K-tile 1: stays in registers until iteration 0 writes it
    OAKE_STAGE(0, NPL);
    OAKE_ADVANCE();
    if (total >= 2) OAKE_VMCNT(kPerKt); else OAKE_VMCNT(0);
    if constexpr (A32) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's A rows of K-tile 0 are in LDS
    OAKE_BAR();  // b0: flat K-tile 0 published
    // LN-folded epilogues: DMA wave lw owns rows [lw * BM/4, +BM/4) of the tile, one row per lane.  At
    // the tile's first K-tile it loads the row's partial (sum x, sum x^2) slices, adds them up in
    // slot order and — two barriers later, when group 1 is done with the previous tile's epilogue (it
    // runs in the second phase of the new tile's first K-tile, group 0's in the first) —
    // writes (rstd, -mean rstd) into EpiLds::kRowstat, 11 K-tiles ahead of the epilogue that reads it.
    // (hipcc guards the loaded values with s_waitcnt vmcnt(0): once per tile this wave also waits
    // for its newest pieces.)
    constexpr bool LN = EpiTraits<EPI>::kLn;
    constexpr int RPW = BM / NL;  // rows per DMA wave
    static_assert(RPW <= 64, "one row per lane");
    float st_rstd = 0.f, st_shift = 0.f;
    for (int g = 0; g < total; ++g) {
      if constexpr (LN) {
        if (d_kt == 0 && lane < RPW) {
          int _m0, _n0;
          tile_origin(tmap, xb + xslot + d_tile * per_xcd, BM, BN, _m0, _n0);
          int _m = _m0 + lw * RPW + lane;
          _m = _m < M ? _m : M - 1;
          // all slices in flight at once (two per 16-B load), then summed in slot order
          const float4* _p = reinterpret_cast<const float4*>(ep.rowpart_in + (size_t)_m * kRowParts);
          float4 _v[kRowParts / 2];
#pragma unroll
          for (int i = 0; i < kRowParts / 2; ++i)
            _v[i] = 2 * i < ep.nparts ? _p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          float _s1 = 0.f, _s2 = 0.f;
#pragma unroll
          for (int i = 0; i < kRowParts / 2; ++i) {
            _s1 += _v[i].x;
            _s2 += _v[i].y;
            if (2 * i + 1 < ep.nparts) {
              _s1 += _v[i].z;
              _s2 += _v[i].w;
            }
          }
          const float2 _st = ln_rowstat(_s1, _s2, ep.inv_k);
          st_rstd = _st.x;
          st_shift = _st.y;
        }
      }
      if constexpr (A32) {
        // flat K-tile g+1's A rows (requested during iteration g-1) -> LDS; then request those of g+2 under this
        // iteration.  (The cursor s_kt / asrc stands at flat K-tile g+2.)
        if (g + 1 < total) a32_write();
        if (s_g < total) a32_load(s_kt);
      }
      if constexpr (PH2) {  // two long phases per K-tile: half of the pieces in each
        OAKE_STAGE(0, Q2);
        OAKE_BAR();
        if constexpr (LN) {  // (group 1's previous-tile epilogue ran in the first K-tile's last phase: one K-tile later)
          if (d_kt == 1 && lane < RPW) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) f32x2* lds_f2w_t;
            *(lds_f2w_t)(smem + NSTAGE * kStageBytes + EpiLds::kRowstat + (lw * RPW + lane) * 8) =
                f32x2{st_rstd, st_shift};
          }
        }
        OAKE_STAGE(Q2, NPL);
      } else {
      OAKE_STAGE(0, Q1);  // flat K-tile g+2, a quarter of the pieces per phase
      OAKE_BAR();
      OAKE_STAGE(Q1, Q2);
      OAKE_BAR();
      if constexpr (LN) {
        if (d_kt == 0 && lane < RPW) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          typedef __attribute__((address_space(3))) f32x2* lds_f2w_t;
          *(lds_f2w_t)(smem + NSTAGE * kStageBytes + EpiLds::kRowstat + (lw * RPW + lane) * 8) =
              f32x2{st_rstd, st_shift};
        }
      }
      OAKE_STAGE(Q2, Q3);
      OAKE_BAR();
      OAKE_STAGE(Q3, NPL);
      }
      OAKE_STAGE_EPI();
      // (with the residual tile staged behind the last K-tile, iteration total - 2 has issued a full set of
      // NPL pieces too — part 0 — and must not wait for them to publish K-tile total - 1)
      const bool newer = g + 2 < total;
      OAKE_ADVANCE();
      if (newer) OAKE_VMCNT(kPerKt); else OAKE_VMCNT(0);  // flat K-tile g+1 landed
      if constexpr (A32) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the A rows written at the top
      OAKE_BAR();  // publishes K-tile g+1; K-tile g's slot is free from here on
    }
    OAKE_BAR();  // pairs with compute group 1's last phase
#undef OAKE_STAGE
#undef OAKE_ABL_SRC
#undef OAKE_STAGE_EPI
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
  const char* elds = smem + NSTAGE * kStageBytes;
  // (cycle stamps are compiled out of the LN-folded variants: they sit exactly at the VGPR limit)
  unsigned long long* const trace = (EpiTraits<EPI>::kLn || PH2) ? nullptr : tmap.trace;

  // the residual epilogue starts its accumulators from the tile's x values (tile_resid_init)
  constexpr bool RIA = EPI == EPI_RESID16 && OAKE_RESID_INIT;
  f32x4 acc[MI][NI];
  if constexpr (RIA) {
    int m0, n0;
    tile_origin(tmap, xb + xslot, BM, BN, m0, n0);
    tile_resid_init<T, MI, NI>(acc, m0 + wm * TM + frow, n0 + wn * TN, fg, M, N, ep, m0 + BM <= M && n0 + BN <= N);
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  vec8 af[MI], bf[NI];
#define OAKE_LOAD_FRAGS(buf_, koff_)                                                        \
  do {                                                                                      \
    const char* _st = smem + (buf_) * kStageBytes;                                          \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
        af[i] = ((OAKE_KLOOP_ABLATE & 1) && (i & 1)) ? af[i - 1]                            \
                : *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + (koff_)); \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                          \
        bf[i] = ((OAKE_KLOOP_ABLATE & 1) && (i & 1)) ? bf[i - 1]                            \
                : *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + (koff_)); \
  } while (0)
// A compute wave raises its issue priority for its MFMA phase (s_setprio 1 .. 0 around the cluster: while it issues
// MFMAs its SIMD's other compute wave reads fragments and the DMA wave issues pieces).  Round 1 had only tried static
// priorities (+-0.5 %); per phase: 112.88 vs 112.59 k images/s, every round above every base round, one lane equal
// (profiles/r04/ab_session_setprio_globals.log), objects 83.7 vs 83.6.  -DOAKE_GEMM_SETPRIO=0: without.
#ifndef OAKE_GEMM_SETPRIO
#define OAKE_GEMM_SETPRIO 1
#endif
#define OAKE_PRIO(n_)                                             \
  do {                                                            \
    if (OAKE_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(n_);        \
  } while (0)
#define OAKE_MFMA_BLOCK()                                                                   \
  do {                                                                                      \
    OAKE_PRIO(1);                                                                           \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
            if (!((OAKE_KLOOP_ABLATE & 4) && (ni & 1)))                                     \
              acc[mi][OAKE_NI_AT(mi, ni, NI)] =                                             \
                  T16<T>::mfma(bf[OAKE_NI_AT(mi, ni, NI)], af[mi], acc[mi][OAKE_NI_AT(mi, ni, NI)]); \
    if (!PH2) OAKE_PRIO(0);                                                                 \
  } while (0)
  // (after the epilogue, not while it consumes the rows: zeroed early, the accumulators would stay
  // live — as zeros — next to the packed tile and the epilogue constants)
#define OAKE_ZERO_ACC()                                                                     \
  do {                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
            acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};                                        \
  } while (0)
#define OAKE_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

  // pending (packed, not yet stored) 16-bit tile: see tile_pack_paired
  constexpr int MI0 = DG ? 2 : 1;  // rows stored immediately at tile end: the K-loop holds acc + fragments +
                          // (MI - MI0) * NI/2 * 4 pending registers inside 168 VGPRs (3 waves per SIMD)
  constexpr bool TRICKLE = EpiTraits<EPI>::kTrickle && !PH2;  // (long phases: the second fragment set takes the pending tile's registers)
  static_assert(!DG || (TRICKLE && EpiTraits<EPI>::kGelu), "deferred GELU rides on the trickled pieces of a GELU epilogue");
  constexpr int NPEND = TRICKLE ? (MI - MI0) * (NI / 2) : 1;
  uint4 pend[TRICKLE ? MI - MI0 : 1][TRICKLE ? NI / 2 : 1];
  T* pend_ptr = nullptr;  // lane's address of the tile's first row-block (mi = 0, t = 0)
  int pend_next = NPEND;  // next pending piece to store (NPEND = none)
#if OAKE_TRICKLE_FULLLINE
  // (pend_ptr = the lane's address of piece [0][0] in the swapped form: row r & 7, columns 32 (r >> 3) + 8 g)
#define OAKE_STORE_PEND(i_)                                                                  \
  tile_store16(pend_ptr + (size_t)(((i_) / (NI / 2) + MI0) * 16 + ((i_) % (NI / 2)) * 8) * ep.ldo, \
               as_u32x4(pend[(i_) / (NI / 2)][(i_) % (NI / 2)]))
#else
#define OAKE_STORE_PEND(i_)                                                                  \
  tile_store16_partial(pend_ptr + (size_t)((i_) / (NI / 2) + MI0) * 16 * ep.ldo + ((i_) % (NI / 2)) * 32, \
               as_u32x4(pend[(i_) / (NI / 2)][(i_) % (NI / 2)]))
#endif

  const unsigned long long t_entry = trace ? __builtin_readcyclecounter() : 0;
  if (trace != nullptr && tid == 0) trace[4096 + blockIdx.x * 2] = wall_clock64();
  OAKE_BAR();            // b0
  if (late) OAKE_BAR();  // group 1 runs one phase behind group 0 (and the DMA waves)
  int c_buf = 0, c_kt = 0, c_tile = 0;
  unsigned long long t_tile = trace ? __builtin_readcyclecounter() : 0;
  for (int g = 0; g < total; ++g) {
    if constexpr (PH2) {
      // two long phases per K-tile (18 fragment reads | 40 MFMAs): half the barrier hand-overs, for the
      // epilogues that keep nothing pending across the K loop and can spare the second fragment set
      vec8 af1[MI], bf1[NI];
      {
        const char* _st = smem + c_buf * kStageBytes;
        constexpr bool kHalfReads = (OAKE_KLOOP_ABLATE & 1) != 0;  // (measurement build: odd fragments = copies)
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = (kHalfReads && (i & 1)) ? af[i - 1] : *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + koff0);
#pragma unroll
        for (int i = 0; i < NI; ++i) bf[i] = (kHalfReads && (i & 1)) ? bf[i - 1] : *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + koff0);
#pragma unroll
        for (int i = 0; i < MI; ++i) af1[i] = (kHalfReads && (i & 1)) ? af1[i - 1] : *reinterpret_cast<const vec8*>(_st + a_base + i * 16 * kRowBytes + koff1);
#pragma unroll
        for (int i = 0; i < NI; ++i) bf1[i] = (kHalfReads && (i & 1)) ? bf1[i - 1] : *reinterpret_cast<const vec8*>(_st + b_base + i * 16 * kRowBytes + koff1);
      }
      OAKE_LGKM0();
      OAKE_BAR();
      OAKE_MFMA_BLOCK();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          if (!((OAKE_KLOOP_ABLATE & 4) && (ni & 1)))
            acc[mi][OAKE_NI_AT(mi, ni, NI)] = T16<T>::mfma(bf1[OAKE_NI_AT(mi, ni, NI)], af1[mi], acc[mi][OAKE_NI_AT(mi, ni, NI)]);
      OAKE_PRIO(0);
    } else {
    OAKE_LOAD_FRAGS(c_buf, koff0);
    if constexpr (DG) if (pend_next < NPEND) {
      // deferred GELU: the first half of this K-tile's piece, under the fragment reads just issued (the wave would
      // otherwise wait for them and then at the barrier for its partner's MFMA phase)
#pragma unroll
      for (int i = 0; i < NPEND; ++i)
        if (i == pend_next) gelu_packed4<T>(pend[i / (NI / 2)][i % (NI / 2)].x, pend[i / (NI / 2)][i % (NI / 2)].y);
    }
    OAKE_LGKM0();
    OAKE_BAR();
    OAKE_MFMA_BLOCK();
    OAKE_BAR();
    OAKE_LOAD_FRAGS(c_buf, koff1);
    if constexpr (DG) if (pend_next < NPEND) {
#pragma unroll
      for (int i = 0; i < NPEND; ++i)
        if (i == pend_next) gelu_packed4<T>(pend[i / (NI / 2)][i % (NI / 2)].z, pend[i / (NI / 2)][i % (NI / 2)].w);
    }
    OAKE_LGKM0();
    if constexpr (TRICKLE) if (pend_next < NPEND) {
      // one trickled store per K-tile, in this wave's load phase (static register indices: a
      // runtime-indexed register array would live in scratch)
      OAKE_PIN();
#pragma unroll
      for (int i = 0; i < NPEND; ++i)
        if (i == pend_next) OAKE_STORE_PEND(i);
      ++pend_next;
    }
    OAKE_BAR();
    OAKE_MFMA_BLOCK();
    }
    c_buf = c_buf == NSTAGE - 1 ? 0 : c_buf + 1;
    // The epilogue runs AFTER the barrier that ends the tile's last MFMA phase, i.e. in the first phase
    // of the next tile's first K-tile: group 0's overlaps group 1's last MFMA phase, group 1's (one
    // phase later) group 0's first, instead of holding the partner wave at the barrier throughout.
    OAKE_BAR();
    if (++c_kt == nk) {
      // tile done
      c_kt = 0;
      int m0, n0;
      tile_origin(tmap, xb + xslot + c_tile * per_xcd, BM, BN, m0, n0);
      ++c_tile;
      // (the block's LAST tile is stored after the loop: done here, group 0's stores would sit in
      // front of a barrier and group 1 would start its own epilogue only once they are issued)
      if (c_tile < my_tiles) {
        const bool interior = m0 + BM <= M && n0 + BN <= N;
        OAKE_PIN();
        // re-derive the lane coordinates behind an opaque asm: hipcc would otherwise hoist every
        // epilogue address out of the K-loop and keep it in a VGPR across it
        int etid = tid;
        asm volatile("" : "+v"(etid));
        const int frow = etid & 15, fg = (etid & 63) >> 4;
        const unsigned long long t_ep = trace ? __builtin_readcyclecounter() : 0;
        bool deferred = false;
        if constexpr (TRICKLE) {
          // flush what is still pending from the previous tile (only when a tile has < NPEND K-tiles)
#pragma unroll
          for (int i = 0; i < NPEND; ++i)
            if (i >= pend_next) {
              if constexpr (DG) {
                gelu_packed4<T>(pend[i / (NI / 2)][i % (NI / 2)].x, pend[i / (NI / 2)][i % (NI / 2)].y);
                gelu_packed4<T>(pend[i / (NI / 2)][i % (NI / 2)].z, pend[i / (NI / 2)][i % (NI / 2)].w);
              }
              OAKE_STORE_PEND(i);
            }
          pend_next = NPEND;
          if (interior) {
            pend_ptr = reinterpret_cast<T*>(ep.out) + (size_t)(m0 + wm * TM + frow) * ep.ldo + n0 +
                       wn * TN + 8 * fg;
            tile_pack_paired<T, EPI, MI, NI, MI0, DG>(acc, pend, ep, pend_ptr, elds, wn * TN + 8 * fg,
                                                      wm * TM + frow);
#if OAKE_TRICKLE_FULLLINE
            pend_ptr += (ptrdiff_t)((frow & 7) - frow) * ep.ldo + ((frow & 8) ? 32 : 0);
#endif
            pend_next = 0;
            deferred = true;
          }
        }
        if (!deferred)  // edge tile, fp32 output or read-modify-write epilogue: store now
          tile_epilogue_lds<T, EPI, MI, NI, RIA>(acc, m0 + wm * TM + frow, n0 + wn * TN, fg, M, N, ep,
                                                 false, interior, elds, wn * TN, wm * TM + frow);
        if constexpr (RIA) {  // the next tile's x values (this block's own tile: nobody else writes it)
          int m1, n1;
          tile_origin(tmap, xb + xslot + c_tile * per_xcd, BM, BN, m1, n1);
          tile_resid_init<T, MI, NI>(acc, m1 + wm * TM + frow, n1 + wn * TN, fg, M, N, ep,
                                     m1 + BM <= M && n1 + BN <= N);
        } else {
          OAKE_ZERO_ACC();
        }
        if (trace != nullptr && (tid & 255) == 0 && blockIdx.x < 64 && c_tile <= 8) {
          OAKE_PIN();
          unsigned long long* tr =
              trace + (((size_t)blockIdx.x * 2 + (tid >> 8)) * 8 + (c_tile - 1)) * 4;
          tr[0] = t_entry; tr[1] = t_tile; tr[2] = t_ep; tr[3] = __builtin_readcyclecounter();
          t_tile = tr[3];
        }
      }
    }
  }
  if (!late) OAKE_BAR();
  if constexpr (TRICKLE) {
#pragma unroll
    for (int i = 0; i < NPEND; ++i)
      if (i >= pend_next) {
        if constexpr (DG) {
          gelu_packed4<T>(pend[i / (NI / 2)][i % (NI / 2)].x, pend[i / (NI / 2)][i % (NI / 2)].y);
          gelu_packed4<T>(pend[i / (NI / 2)][i % (NI / 2)].z, pend[i / (NI / 2)][i % (NI / 2)].w);
        }
        OAKE_STORE_PEND(i);
      }
  }
  {
    int m0, n0;
    tile_origin(tmap, xb + xslot + (my_tiles - 1) * per_xcd, BM, BN, m0, n0);
    const unsigned long long t_ep = trace ? __builtin_readcyclecounter() : 0;
    tile_epilogue_lds<T, EPI, MI, NI, RIA>(acc, m0 + wm * TM + frow, n0 + wn * TN, fg, M, N, ep, false,
                                           m0 + BM <= M && n0 + BN <= N, elds, wn * TN, wm * TM + frow);
    if (trace != nullptr && (tid & 255) == 0 && blockIdx.x < 64 && my_tiles <= 8) {
      unsigned long long* tr =
          trace + (((size_t)blockIdx.x * 2 + (tid >> 8)) * 8 + (my_tiles - 1)) * 4;
      tr[0] = t_entry; tr[1] = t_tile; tr[2] = t_ep; tr[3] = __builtin_readcyclecounter();
    }
  }
  if (trace != nullptr && tid == 256) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    trace[4096 + blockIdx.x * 2 + 1] = wall_clock64();
  }
#undef OAKE_ZERO_ACC
#undef OAKE_STORE_PEND
#undef OAKE_LOAD_FRAGS
#undef OAKE_MFMA_BLOCK
#undef OAKE_PRIO
#undef OAKE_LGKM0
#undef OAKE_PIN
#undef OAKE_BAR
}


#include "gemm_w8.inc"
#if OAKE_LAB
#include "gemm_lab_duo.inc"
#endif

// ---------------------------------------------------------------------------------------------
// Register-only MFMA stream (measurement: bench.py's `roofline.sustained`, tools/ubench/mfma_power.hip):
// every SIMD runs two waves of back-to-back v_mfma_f32_16x16x32_f16 on the production 5 x 4 wave tile
// with the operand fragments the caller supplies — no LDS, no memory traffic.  What it measures is the
// matrix rate the board sustains under its power cap for that operand DATA (zeros: the 2.4 PFLOP/s of the
// data sheet at 2.39 GHz / 0.75 kW; N(0, 0.25) halves: 1.9 PFLOP/s at 1.94 GHz / 1.33 kW, capped).
__global__ __launch_bounds__(512) void mfma_probe_kernel(const f16x8* __restrict__ frags, float* sink,
                                                         int iters) {
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
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (t == 12345.678f) sink[0] = t;  // (keeps the MFMAs alive; never true for the data the probe is given)
}

// The same probe with v_mfma_f32_32x32x16_f16: a 160 x 64 wave tile as 5 x 2 tiles of 32 x 32 (160 accumulator
// registers), ten MFMAs of 32 768 FLOP per 16 values of K — half the A / B operand reads per FLOP of the 16x16x32
// form, twice the accumulator traffic.  (Round 6: does the board sustain a different rate on the other shape?)
__global__ __launch_bounds__(512) void mfma_probe32_kernel(const f16x8* __restrict__ frags, float* sink, int iters) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
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
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  if (t == 12345.678f) sink[0] = t;
}

// ... and the ORDER of a wave's MFMAs over its 10 x 4 tiles of 16 x 16 (gemm_w8_kernel's wave tile, 160 accumulator
// registers): 0 = row by row (consecutive MFMAs share the A fragment; at a row change both operands change), 1 = serpentine
// (every consecutive pair shares one operand), 2 = column by column (shares the B fragment, ten MFMAs per column).
// The accumulation order of every tile is the same in all three: a power question only.
template <int ORDER>
__global__ __launch_bounds__(512) void mfma_probe_order_kernel(const f16x8* __restrict__ frags, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 af[10], bf[4];
#pragma unroll
  for (int i = 0; i < 10; ++i) af[i] = frags[(i % 9) * 64 + lane] + frags[((i + 3) % 9) * 64 + lane];
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = frags[(5 + j) * 64 + lane];
  f32x4 acc[10][4];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if constexpr (ORDER == 2 || ORDER == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ii = 0; ii < 10; ++ii) {
          const int i = (ORDER == 3 && (j & 1)) ? 9 - ii : ii;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = (ORDER == 1 && (i & 1)) ? 3 - jj : jj;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (t == 12345.678f) sink[0] = t;
}

// ---- host side ------------------------------------------------------------------------------
TileMap make_tilemap(const GemmArgs& a, int BM, int BN) {
  TileMap tmap;
  tmap.tiles_m = (a.M + BM - 1) / BM;
  tmap.tiles_n = (a.N + BN - 1) / BN;
  tmap.nwg = tmap.tiles_m * tmap.tiles_n;
  // N panels of ~1536 columns where K <= 1024 (a W panel of K = 768 is 2.4 MB of an XCD's 4 MiB L2; an XCD's 120 c_fc tiles
  // are then 20 row tiles x 6 column tiles: 4.9 MB of A + 2.4 MB of W per XCD instead of 9.8 + 1.2), ~768 columns for
  // longer K.  Round 2 had measured wider panels WORSE (HBM-side reads 118 / 109 / 114 / 175 MB at 3 / 4 / 6 / 12 tile
  // columns: docs/history/round2_what_bounds_the_gemm.md item 4) — with write-back tile stores, whose 78.6 MB of output
  // lines shared the L2 with the operands.  With the write-through stores of round 3 the same sweep reads 107.5 / 86.2 /
  // 71.3 MB (c_fc, profiles/r06/gemm_panel_fetch.txt) and the bench is equal or better in every mode (globals +0.0..0.4 %,
  // objects +0.2 %, blocks +0.6 %: profiles/r06/ab_gemm_panel_*.log).
  int pn = (a.K <= 1024 ? 1536 : 768) / BN;
  pn = pn < 1 ? 1 : pn;
  tmap.by_m = 0;
  int panel = a.opts ? a.opts->gemm_panel : 0;
  if (panel >= 1000) panel -= 1000;  // (>= 1000: launch_duo's one-workgroup-per-CU switch)
  if (panel > 0) pn = panel;
  if (panel < 0) {
    tmap.by_m = 1;
    pn = -panel;
  }
  const int outer = tmap.by_m ? tmap.tiles_m : tmap.tiles_n;
  tmap.pn = pn > outer ? outer : pn;
  tmap.trace = a.opts ? a.opts->gemm_trace : nullptr;  // per-tile s_memtime stamps (tools/gemm_trace.py)
  return tmap;
}

template <typename T, int EPI, int BM, int BN, int WM, int WN>
hipError_t launch_simple(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * kRowBytes;
  static DynLdsAttr attr;
  auto kern = gemm_kernel<T, EPI, BM, BN, WM, WN>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  const TileMap tmap = make_tilemap(a, BM, BN);
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L,
               reinterpret_cast<const float2*>(a.rowstat), a.colsum,
               reinterpret_cast<float2*>(a.rowpart_out), reinterpret_cast<const float2*>(a.rowpart_in),
               a.nparts, 1.0f / (float)a.K, 0, 0, 0};
  OAKE_LAUNCH(kern, dim3(tmap.nwg), dim3(WM * WN * 64), lds, s,
                     reinterpret_cast<const T*>(a.A), reinterpret_cast<const T*>(a.W), a.M, a.N,
                     a.K, ep, tmap);
  return hipGetLastError();
}

template <typename T, int EPI, int BM, int BN, int WM, int WN>
hipError_t launch_deep(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 4 * (BM + BN) * kRowBytes;
  static DynLdsAttr attr;
  auto kern = gemm_deep_kernel<T, EPI, BM, BN, WM, WN>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  const TileMap tmap = make_tilemap(a, BM, BN);
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L,
               reinterpret_cast<const float2*>(a.rowstat), a.colsum,
               reinterpret_cast<float2*>(a.rowpart_out), reinterpret_cast<const float2*>(a.rowpart_in),
               a.nparts, 1.0f / (float)a.K, 0, 0, 0};
  OAKE_LAUNCH(kern, dim3(tmap.nwg), dim3(WM * WN * 64), lds, s, reinterpret_cast<const T*>(a.A),
              reinterpret_cast<const T*>(a.W), a.M, a.N, a.K, ep, tmap);
  return hipGetLastError();
}

#if OAKE_LAB
template <typename T, int EPI, int BM, int BN>
hipError_t launch_q4(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * kRowBytes + EpiLds::kBytes;
  static DynLdsAttr attr;
  auto kern = gemm_q4_kernel<T, EPI, BM, BN>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  if (a.K < 2 * BK) return launch_simple<T, EPI, 128, 128, 2, 2>(a, s);
  const TileMap tmap = make_tilemap(a, BM, BN);
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L,
               reinterpret_cast<const float2*>(a.rowstat), a.colsum,
               reinterpret_cast<float2*>(a.rowpart_out), reinterpret_cast<const float2*>(a.rowpart_in),
               a.nparts, 1.0f / (float)a.K, 0, 0, 0};
  OAKE_LAUNCH(kern, dim3(tmap.nwg), dim3(512), lds, s, reinterpret_cast<const T*>(a.A),
              reinterpret_cast<const T*>(a.W), a.M, a.N, a.K, ep, tmap);
  return hipGetLastError();
}

#endif

template <typename T, int EPI, int BM, int BN, int WM, int WN, bool PH2 = false, bool A32 = false, bool DG = false>
hipError_t launch_pp(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (BM + BN) * kRowBytes + EpiLds::kBytes;
  static_assert(BM <= 160 && BN <= 256, "EpiLds layout");
  static DynLdsAttr attr;
  auto kern = gemm_pp_kernel<T, EPI, BM, BN, WM, WN, PH2, A32, DG>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  int num_cu = 0;
  if (hipError_t e = device_cu_count(&num_cu); e != hipSuccess) return e;
  if (a.opts && a.opts->cu_count > 0 && a.opts->cu_count < num_cu) num_cu = a.opts->cu_count;  // CU-masked stream
  if (a.K < 3 * BK) {  // the EpiLds staging needs >= 3 K-tiles per tile
    if (a.patch_S != 0) return hipErrorInvalidValue;
    return launch_simple<T, EPI, BM, BN, WM, WN>(a, s);
  }
  const TileMap tmap = make_tilemap(a, BM, BN);
  int grid = (num_cu / 8) * 8;  // one persistent block per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  const int need = ((tmap.nwg + 7) / 8) * 8;
  if (grid > need) grid = need;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L,
               reinterpret_cast<const float2*>(a.rowstat), a.colsum,
               reinterpret_cast<float2*>(a.rowpart_out), reinterpret_cast<const float2*>(a.rowpart_in),
               a.nparts, 1.0f / (float)a.K, 0, 0, 0};
  if (a.patch_S != 0) {
    if (EPI != EPI_PATCH16 && EPI != EPI_PATCH) return hipErrorInvalidValue;
    ep.patch_S = a.patch_S; ep.patch_P = a.patch_P; ep.patch_G = a.patch_G;
    ep.patch_T = a.patch_T ? a.patch_T : a.patch_P; ep.patch_H = a.patch_H ? a.patch_H : a.patch_S;
  }
  OAKE_LAUNCH(kern, dim3(grid), dim3((WM * WN + 4) * 64), lds, s,
                     reinterpret_cast<const T*>(a.A), reinterpret_cast<const T*>(a.W), a.M, a.N,
                     a.K, ep, tmap);
  return hipGetLastError();
}

// c_fc's kernel: the 320 x 256 tile, eight compute waves that issue their own LDS-DMA (gemm_w8.inc; GEMM variant 13).
// What it takes: a plain matrix A, the 16-bit tile epilogues, and — LayerNorm-folded — seven K-tiles to hand the row
// statistics over.  Anything else goes to the 160 x 256 kernel.
inline bool w8_takes(int epi, const GemmArgs& a) {
  const bool ln = epi == EPI_T16_BIAS_LN || epi == EPI_T16_GELU_LN;
  if (!ln && epi != EPI_T16_BIAS && epi != EPI_T16_GELU && epi != EPI_RESID16 && epi != EPI_T16_NONE && epi != EPI_T16_RAW) return false;
  return a.patch_S == 0 && a.K >= (ln ? 7 : 3) * BK;
}
// ... and what the automatic choice gives it: problems whose 320-row tiles fill the chip as well as the 160-row ones
// do (a 320-row tile costs two of those in this count, 1.78 measured).  At 12 800 rows that is c_fc — N = 3072: 480
// tiles = two rounds of 256 CUs against four — and not qkv (N = 2304: two rounds against three; measured 56 us
// against 47, profiles/r06/w8/).
inline bool w8_preferred(int epi, const GemmArgs& a) {
  if (!w8_takes(epi, a) || epi == EPI_T16_NONE || epi == EPI_T16_RAW) return false;
  int num_cu = 0;
  if (device_cu_count(&num_cu) != hipSuccess || num_cu <= 0) return false;
  if (a.opts && a.opts->cu_count > 0 && a.opts->cu_count < num_cu) num_cu = a.opts->cu_count;
  const long tn = (a.N + 255) / 256;
  const long t320 = (a.M + 319) / 320 * tn, t160 = (a.M + 159) / 160 * tn;
  return 2 * ((t320 + num_cu - 1) / num_cu) <= (t160 + num_cu - 1) / num_cu;
}

template <typename T, int EPI>
hipError_t launch_w8(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = 320, BN = 256;
  constexpr int lds = 2 * (BM + BN) * kRowBytes + W8Lds::kBytes;
  static DynLdsAttr attr;
  auto kern = gemm_w8_kernel<T, EPI>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  int num_cu = 0;
  if (hipError_t e = device_cu_count(&num_cu); e != hipSuccess) return e;
  if (a.opts && a.opts->cu_count > 0 && a.opts->cu_count < num_cu) num_cu = a.opts->cu_count;
  if (!w8_takes(EPI, a)) return launch_pp<T, EPI, 160, 256, 2, 4, EpiTraits<EPI>::kLn || EPI == EPI_RESID16>(a, s);
  const TileMap tmap = make_tilemap(a, BM, BN);
  int grid = (num_cu / 8) * 8;
  if (grid < 8) grid = 8;
  const int need = ((tmap.nwg + 7) / 8) * 8;
  if (grid > need) grid = need;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L,
               reinterpret_cast<const float2*>(a.rowstat), a.colsum,
               reinterpret_cast<float2*>(a.rowpart_out), reinterpret_cast<const float2*>(a.rowpart_in),
               a.nparts, 1.0f / (float)a.K, 0, 0, 0};
  OAKE_LAUNCH(kern, dim3(grid), dim3(512), lds, s, reinterpret_cast<const T*>(a.A), reinterpret_cast<const T*>(a.W), a.M,
              a.N, a.K, ep, tmap);
  return hipGetLastError();
}

#if OAKE_LAB
template <typename T, int EPI>
hipError_t launch_variant_lab(int variant, const GemmArgs& a, hipStream_t s);

template <typename T, int EPI, int BM, int BN>
hipError_t launch_duo(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * kRowBytes + EpiLds::kBytes;
  static_assert(BM <= 160 && BN <= 256 && 2 * lds <= 160 * 1024, "two workgroups per CU");
  static DynLdsAttr attr;
  auto kern = gemm_duo_kernel<T, EPI, BM, BN>;
  if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  int num_cu = 0;
  if (hipError_t e = device_cu_count(&num_cu); e != hipSuccess) return e;
  if (a.opts && a.opts->cu_count > 0 && a.opts->cu_count < num_cu) num_cu = a.opts->cu_count;  // CU-masked stream
  if (a.K < 3 * BK) {  // (as launch_pp: the callers' row-statistics hand-off assumes the same threshold)
    if (a.patch_S != 0) return hipErrorInvalidValue;
    return launch_simple<T, EPI, BM, BN, 2, 2>(a, s);
  }
  const TileMap tmap = make_tilemap(a, BM, BN);
  const int per_cu = a.opts && a.opts->gemm_panel >= 1000 ? 1 : 2;  // (1000: one workgroup per CU, the A/B of the overlap)
  int grid = (per_cu * num_cu / 8) * 8;
  if (grid < 8) grid = 8;
  const int need = ((tmap.nwg + 7) / 8) * 8;
  if (grid > need) grid = need;
  EpiParams ep{a.bias, a.out, a.ldo, a.pos, a.P2, a.L,
               reinterpret_cast<const float2*>(a.rowstat), a.colsum,
               reinterpret_cast<float2*>(a.rowpart_out), reinterpret_cast<const float2*>(a.rowpart_in),
               a.nparts, 1.0f / (float)a.K, 0, 0, 0};
  if (a.patch_S != 0) {
    if (EPI != EPI_PATCH16 && EPI != EPI_PATCH) return hipErrorInvalidValue;
    ep.patch_S = a.patch_S; ep.patch_P = a.patch_P; ep.patch_G = a.patch_G;
    ep.patch_T = a.patch_T ? a.patch_T : a.patch_P; ep.patch_H = a.patch_H ? a.patch_H : a.patch_S;
  }
  OAKE_LAUNCH(kern, dim3(grid), dim3(512), lds, s, reinterpret_cast<const T*>(a.A),
              reinterpret_cast<const T*>(a.W), a.M, a.N, a.K, ep, tmap);
  return hipGetLastError();
}

#endif

// Configurations.  0: simple 128x128 (4 waves)   1: simple 160x256 (8 waves 2x4)
//                  2: simple 320x128 (8 waves 4x2)   3: simple 256x256 (8 waves 2x4)
//                  4: ping-pong persistent 160x256 (8 compute + 4 DMA waves)  [production; the residual, conv1 and
//                     LayerNorm-folded (qkv, c_fc) epilogues run the K loop in two long phases per K-tile, the others in four]
//                  8: the same at 128x256 (experiment)   9: two long phases wherever they fit   10: four phases everywhere
//                  5: deep-ring 64x64 (4 waves, 4-slot ring) for the few-hundred-row problems (head,
//                     CLS rows of the last block, object stream)   6: simple 64x64 (2-slot ring)
//                  7: one compute wave per SIMD, 160x256, one tile per block (gemm_q4_kernel; experiment)
//                 11: two workgroups per CU, 160x128 tiles (gemm_duo_kernel; experiment, docs/history/round2_what_bounds_the_gemm.md item 10)
//                 12: variant 4 with c_fc's QuickGELU deferred into the next tile's load phases (four phases; OAKE_DEFER_GELU)
#if OAKE_LAB
template <typename T, int EPI>
hipError_t launch_variant_lab(int variant, const GemmArgs& a, hipStream_t s) {
  if constexpr (EpiTraits<EPI>::kNone || EpiTraits<EPI>::kRaw) {  // measurement epilogues: persistent kernels only
    if (variant == 8) return launch_pp<T, EPI, 128, 256, 2, 4>(a, s);
    if (variant == 11) return launch_duo<T, EPI, 160, 128>(a, s);
    if (variant == 13) return launch_w8<T, EPI>(a, s);
    if constexpr (EpiTraits<EPI>::kNone) {  // (9: the K loop alone in the production schedule, two long phases per K-tile)
      if (variant == 9) return launch_pp<T, EPI, 160, 256, 2, 4, true>(a, s);
    }
    return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);
  } else
  switch (variant) {
    case 0: return launch_simple<T, EPI, 128, 128, 2, 2>(a, s);
    case 1: return launch_simple<T, EPI, 160, 256, 2, 4>(a, s);
    case 2: return launch_simple<T, EPI, 320, 128, 4, 2>(a, s);
    case 3: return launch_simple<T, EPI, 256, 256, 2, 4>(a, s);
    case 4:  // production: two long phases per K-tile where the epilogue keeps no tile pending (residual, conv1)
      if constexpr (EPI == EPI_PATCH16) {
        if (a.patch_f32) return launch_pp<T, EPI, 160, 256, 2, 4, true, true>(a, s);
      }
      if constexpr (EPI == EPI_RESID16 || EPI == EPI_PATCH16)
        return launch_pp<T, EPI, 160, 256, 2, 4, true>(a, s);
      // ... and for c_fc's epilogue (LayerNorm affine + QuickGELU): long phases + all stores at the tile end
      // beat four phases + trickled stores there (+1.1 % on the bench, A/B of two builds in one session; for
      // qkv's lighter epilogue the same switch is neutral: it keeps the trickled stores)
      else if constexpr (EPI == EPI_T16_GELU_LN)
        return OAKE_DEFER_GELU ? launch_pp<T, EPI, 160, 256, 2, 4, false, false, true>(a, s)
                               : launch_pp<T, EPI, 160, 256, 2, 4, true>(a, s);
      // ... and, with the tile stores written through (round 3), for qkv's too: all of a tile's stores at its end
      // as full lines, nothing left dirty in the L2s — +0.7 % globals, +0.4 % blocks, +0.2 % objects
      // (profiles/r03/ab_session_k_*; before the write-through stores the same switch was neutral)
      else if constexpr (EPI == EPI_T16_BIAS_LN)
        return launch_pp<T, EPI, 160, 256, 2, 4, true>(a, s);
      else
        return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);
    case 11:  // two workgroups per CU, 160x128 tiles (the 16-bit epilogues that fit 128 registers)
      if constexpr (EpiTraits<EPI>::kPaired && !EpiTraits<EPI>::kLn && EPI != EPI_RESID16)
        return launch_duo<T, EPI, 160, 128>(a, s);
      else
        return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);
    case 10: return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);  // four short phases for every epilogue (A/B, cycle stamps)
    case 13:  // 320 x 256 tile, eight compute waves issuing their own LDS-DMA (gemm_w8_kernel; the 16-bit tile epilogues)
      if constexpr (EPI == EPI_T16_BIAS || EPI == EPI_T16_GELU || EPI == EPI_T16_BIAS_LN || EPI == EPI_T16_GELU_LN ||
                    EPI == EPI_RESID16)
        return launch_w8<T, EPI>(a, s);
      else
        return launch_variant_lab<T, EPI>(4, a, s);
    case 12:  // four phases + the QuickGELU of c_fc deferred into the next tile's load phases (round 6)
      if constexpr (EPI == EPI_T16_GELU_LN || EPI == EPI_T16_GELU)
        return launch_pp<T, EPI, 160, 256, 2, 4, false, false, true>(a, s);
      else
        return launch_variant_lab<T, EPI>(4, a, s);
    case 5: return launch_deep<T, EPI, 64, 64, 2, 2>(a, s);
    case 6: return launch_simple<T, EPI, 64, 64, 2, 2>(a, s);
    case 8: return launch_pp<T, EPI, 128, 256, 2, 4>(a, s);  // experiment: 64 x 64 wave tiles
    case 9:  // two long phases per K-tile (epilogues without a pending tile / LayerNorm statistics)
      if constexpr (EPI == EPI_RESID16 || EPI == EPI_PATCH16 || EPI == EPI_F32_BIAS)
        return launch_pp<T, EPI, 160, 256, 2, 4, true>(a, s);
      else
        return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);
    case 7:  // one compute wave per SIMD (experiment): residual / bias epilogues without LN statistics from LDS
      if constexpr (EPI == EPI_RESID16 || EPI == EPI_T16_BIAS || EPI == EPI_F32_BIAS)
        return launch_q4<T, EPI, 160, 256>(a, s);
      else
        return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);
    default: return hipErrorInvalidValue;
  }
}

#endif

// The production library: the configurations the automatic choice can make — 0 (simple 128x128: narrow N), 4 (the
// persistent ping-pong kernel, 160 x 256), 5 (deep-ring 64x64: few-hundred-row problems) and 13 (the 320 x 256 tile of
// gemm_w8.inc: c_fc).  Everything else is in the lab build.
template <typename T, int EPI>
hipError_t launch_variant(int variant, const GemmArgs& a, hipStream_t s) {
#if OAKE_LAB
  return launch_variant_lab<T, EPI>(variant, a, s);
#else
  if constexpr (EpiTraits<EPI>::kNone || EpiTraits<EPI>::kRaw) {
    return hipErrorInvalidValue;  // measurement epilogues: lab build only
  } else
  switch (variant) {
    case 0: return launch_simple<T, EPI, 128, 128, 2, 2>(a, s);
    case 13:  // the 320 x 256 tile for the 16-bit tile epilogues (c_fc; out_proj / c_proj); every other epilogue: as 4
      if constexpr (EPI == EPI_T16_BIAS || EPI == EPI_T16_GELU || EPI == EPI_T16_BIAS_LN || EPI == EPI_T16_GELU_LN ||
                    EPI == EPI_RESID16)
        return launch_w8<T, EPI>(a, s);
      [[fallthrough]];
    case 4:  // two long phases per K-tile where the epilogue keeps no tile pending (residual, conv1) and for the
             // LayerNorm-folded epilogues (qkv, c_fc: all of a tile's stores at its end, full lines, written through)
      if (a.patch_f32) return hipErrorInvalidValue;  // (fp32 conv1 gather through the DMA waves' registers: lab build)
#if OAKE_DEFER_GELU
      if constexpr (EPI == EPI_T16_GELU_LN) return launch_pp<T, EPI, 160, 256, 2, 4, false, false, true>(a, s);
#endif
      if constexpr (EPI == EPI_RESID16 || EPI == EPI_PATCH16 || EPI == EPI_T16_GELU_LN || EPI == EPI_T16_BIAS_LN)
        return launch_pp<T, EPI, 160, 256, 2, 4, true>(a, s);
      else
        return launch_pp<T, EPI, 160, 256, 2, 4>(a, s);
    case 5: return launch_deep<T, EPI, 64, 64, 2, 2>(a, s);
    default: return hipErrorInvalidValue;
  }
#endif
}

int pick_variant(const GemmArgs& a) {
  if (a.opts && a.opts->gemm_variant >= 0) return a.opts->gemm_variant;
  // few-hundred-row problems: 64x64 tiles spread over the CUs, deep ring against the per-K-tile latency
  if ((long)a.M * a.N <= 512 * 1024 || a.M <= 1024) return 5;
  if (a.N < 256) return 0;
  return 4;
}

template <typename T>
hipError_t launch_epi(int epi, const GemmArgs& a, hipStream_t s) {
  int v = pick_variant(a);
  if (v == 4 && !(a.opts && a.opts->gemm_variant != -1) && w8_preferred(epi, a)) v = 13;  // (-2: A/B runs without it)
  switch (epi) {
    case EPI_F32_BIAS: return launch_variant<T, EPI_F32_BIAS>(v, a, s);
    case EPI_T16_BIAS: return launch_variant<T, EPI_T16_BIAS>(v, a, s);
    case EPI_T16_GELU: return launch_variant<T, EPI_T16_GELU>(v, a, s);
    case EPI_RESID: return launch_variant<T, EPI_RESID>(v, a, s);
    case EPI_PATCH: return launch_variant<T, EPI_PATCH>(v, a, s);
    case EPI_RESID16: return launch_variant<T, EPI_RESID16>(v, a, s);
    case EPI_PATCH16: return launch_variant<T, EPI_PATCH16>(v, a, s);
    case EPI_T16_BIAS_LN: return launch_variant<T, EPI_T16_BIAS_LN>(v, a, s);
    case EPI_T16_GELU_LN: return launch_variant<T, EPI_T16_GELU_LN>(v, a, s);
    case EPI_T16_NONE: return launch_variant<T, EPI_T16_NONE>(v, a, s);
    case EPI_T16_RAW: return launch_variant<T, EPI_T16_RAW>(v, a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool gemm_variant_supported(int v) {
#if OAKE_LAB
  return v >= -2 && v <= 13;
#else
  return v == -2 || v == -1 || v == 0 || v == 4 || v == 5 || v == 13;
#endif
}

bool gemm_uses_persistent(int M, int N, int K, const LaunchOpts* opts) {
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K;
  a.opts = opts;
  const int v = pick_variant(a);
  return (v == 4 || v == 8 || v == 9 || v == 10 || v == 11 || v == 12 || v == 13) && K >= 3 * BK;
}

bool gemm_patch_direct_ok(int image, int patch, int stride, int padding, int M, int N, int K,
                          const LaunchOpts* opts) {
  // 8-pixel chunks stay inside a patch row, a K-tile is whole patch rows of one channel, every chunk is
  // 16-byte aligned, and the shape runs the persistent kernel (the only one with the gather)
  return stride == patch && padding == 0 && image % patch == 0 && patch % 8 == 0 && BK % patch == 0 &&
         (patch * patch) % BK == 0 && image % 8 == 0 && K == 3 * patch * patch &&
         gemm_uses_persistent(M, N, K, opts);
}

bool gemm_patch_padded_ok(int patch, int stride, int M, int N, int K, const LaunchOpts* opts) {
  // as gemm_patch_direct_ok, for patches read from a zero-padded buffer with 16-byte-aligned rows: the patch
  // origins (multiples of the stride) and the 8-pixel chunks must stay 16-byte aligned
  return patch % 8 == 0 && stride % 8 == 0 && BK % patch == 0 && (patch * patch) % BK == 0 &&
         K == 3 * patch * patch && gemm_uses_persistent(M, N, K, opts);
}

hipError_t launch_mfma_probe_order(const void* d_frags, float* d_sink, int iters, int order, double* flop, hipStream_t s) {
  int cus = 0;
  if (hipError_t e = device_cu_count(&cus); e != hipSuccess) return e;
  const f16x8* f = reinterpret_cast<const f16x8*>(d_frags);
  if (order == 0) OAKE_LAUNCH(mfma_probe_order_kernel<0>, dim3(cus), dim3(512), 0, s, f, d_sink, iters);
  else if (order == 1) OAKE_LAUNCH(mfma_probe_order_kernel<1>, dim3(cus), dim3(512), 0, s, f, d_sink, iters);
  else if (order == 2) OAKE_LAUNCH(mfma_probe_order_kernel<2>, dim3(cus), dim3(512), 0, s, f, d_sink, iters);
  else if (order == 3) OAKE_LAUNCH(mfma_probe_order_kernel<3>, dim3(cus), dim3(512), 0, s, f, d_sink, iters);
  else return hipErrorInvalidValue;
  if (flop != nullptr) *flop = (double)cus * 8 * 40 * 16384.0 * (double)iters;
  return hipGetLastError();
}

hipError_t launch_mfma_probe32(const void* d_frags, float* d_sink, int iters, double* flop, hipStream_t s) {
  int cus = 0;
  if (hipError_t e = device_cu_count(&cus); e != hipSuccess) return e;
  OAKE_LAUNCH(mfma_probe32_kernel, dim3(cus), dim3(512), 0, s, reinterpret_cast<const f16x8*>(d_frags), d_sink, iters);
  if (flop != nullptr) *flop = (double)cus * 8 * 10 * 32768.0 * (double)iters;
  return hipGetLastError();
}

hipError_t launch_mfma_probe(const void* d_frags, float* d_sink, int iters, double* flop, hipStream_t s) {
  int cus = 0;
  if (hipError_t e = device_cu_count(&cus); e != hipSuccess) return e;
  OAKE_LAUNCH(mfma_probe_kernel, dim3(cus), dim3(512), 0, s, reinterpret_cast<const f16x8*>(d_frags), d_sink,
              iters);
  if (flop != nullptr) *flop = (double)cus * 8 * 20 * 16384.0 * (double)iters;
  return hipGetLastError();
}

bool gemm_patch_f32_ok(int image, int patch, int stride, int padding, int M, int N, int K,
                       const LaunchOpts* opts) {
  // the 16-bit gather's geometry, on the production configuration of the persistent kernel (the one instantiated
  // with the register-staged A path), 8 floats per lane = 32 contiguous bytes of a patch row
#if OAKE_LAB
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.opts = opts;
  const int v = pick_variant(a);
  return gemm_patch_direct_ok(image, patch, stride, padding, M, N, K, opts) && v == 4 && patch == 32;
#else
  return false;  // (measured slower than im2col + GEMM with two lanes: lab build only)
#endif
}

hipError_t launch_gemm(int dtype16, int epi, const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
  if (a.patch_f32 && (epi != EPI_PATCH16 || a.patch_S == 0 || a.patch_T != 0 || a.patch_P != 32 ||
                      pick_variant(a) != 4))
    return hipErrorInvalidValue;
  if (a.patch_S != 0 && !gemm_uses_persistent(a.M, a.N, a.K, a.opts)) return hipErrorInvalidValue;
  if (a.K % BK != 0 || a.N % 4 != 0 || a.ldo % 4 != 0) return hipErrorInvalidValue;
  if ((epi == EPI_T16_BIAS || epi == EPI_T16_GELU || epi == EPI_RESID16 || epi == EPI_PATCH16 ||
       epi == EPI_T16_BIAS_LN || epi == EPI_T16_GELU_LN || epi == EPI_T16_NONE || epi == EPI_T16_RAW) &&
      (a.N % 8 != 0 || a.ldo % 8 != 0))
    return hipErrorInvalidValue;
  if (epi == EPI_T16_BIAS_LN || epi == EPI_T16_GELU_LN) {
    if (a.colsum == nullptr || a.bias == nullptr) return hipErrorInvalidValue;
    if (gemm_uses_persistent(a.M, a.N, a.K, a.opts)
            ? (a.rowpart_in == nullptr || a.nparts < 1 || a.nparts > kRowParts)
            : a.rowstat == nullptr)
      return hipErrorInvalidValue;
  }
  if (dtype16 == DT_F16) return launch_epi<f16_t>(epi, a, s);
  if (dtype16 == DT_BF16) return launch_epi<bf16_t>(epi, a, s);
  return hipErrorInvalidValue;
}

}  // namespace oake
