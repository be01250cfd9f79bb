// kernels.h — host-side launcher declarations shared by the translation units of liboake_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

// Order of a wave's MFMAs over its MI x NI tiles inside one K-step: 0 = row by row (at a row change both operands of
// consecutive MFMAs change), 1 = serpentine (every consecutive pair shares one operand fragment).  Every tile's own
// accumulation order is unchanged: results are bit-identical; the board's power is not (tools/mfma_shape_probe.py:
// the register-only stream sustains 1.94 PFLOP/s serpentine against 1.90 row by row on random operands).
#ifndef OAKE_MFMA_ORDER
#define OAKE_MFMA_ORDER 1
#endif
#define OAKE_NI_AT(mi_, ni_, NI_) ((OAKE_MFMA_ORDER == 1 && ((mi_) & 1)) ? (NI_) - 1 - (ni_) : (ni_))

namespace oake {

// 16-bit operand type selector (matches OAKE_F16 / OAKE_BF16 in include/oake_hip.h)
enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_U8 = 3 };

// ---- GEMM: C[M,N] = A[M,K] * W[N,K]^T, 16-bit operands, fp32 accumulate -------------------
enum GemmEpi {
  EPI_F32_BIAS = 0,   // out fp32 [M,ldo] = acc + bias                 (debug / tests)
  EPI_T16_BIAS = 1,   // out T16 [M,ldo] = acc + bias                  (QKV in-proj)
  EPI_T16_GELU = 2,   // out T16 [M,ldo] = quick_gelu(acc + bias)      (MLP c_fc)
  EPI_RESID = 3,      // out fp32 [M,ldo] += acc + bias                (attn out_proj, MLP c_proj)
  EPI_PATCH = 4,      // out fp32 x[(m/P2)*L + 1 + m%P2, :] = acc + pos[1 + m%P2, :]   (conv1)
  EPI_RESID16 = 5,    // EPI_RESID on a 16-bit residual stream (read-modify-write in T16)
  EPI_PATCH16 = 6,    // EPI_PATCH writing a 16-bit residual stream
  // LayerNorm folded in (rowops.hip, fold_ln_kernel): W carries gamma, `bias` is b + W beta, and
  // out = rowstat[m].x * acc + rowstat[m].y * colsum[n] + bias[n]   (then QuickGELU for _GELU_LN)
  EPI_T16_BIAS_LN = 7,
  EPI_T16_GELU_LN = 8,
  // measurement-only epilogues (tools/gemm_ablate.py): what a tile costs without its epilogue
  EPI_T16_NONE = 9,   // nothing is stored (accumulators kept alive, then zeroed)
  EPI_T16_RAW = 10    // out T16 = acc (no bias, no activation): the pack + store cost alone
};

// Kernel-selection switches of one caller (a handle, or the calling thread's handle-less oake_debug_* entry
// points).  No process-wide state: two handles / lanes never see each other's settings.
constexpr int kAttentionVariantDefault = 159;  // attention.hip: bits 1 | 2 | 4 | 8 | 16 | 128
constexpr int kQkvWalkDefault = 4;             // qkv_attn_obj.hip: tile walk in head blocks of 4 (0 = group-major)

struct LaunchOpts {
  int gemm_variant = -1;   // -1 = automatic per shape (-2: the same without the 320-row tile), else a forced tile configuration (csrc/gemm.hip)
  int gemm_panel = 0;      // tile order: 0 default, n > 0 N panels of n tiles, n < 0 M slabs of -n tiles
  unsigned long long* gemm_trace = nullptr;  // device buffer for per-tile cycle stamps, or nullptr
  int attention_variant = kAttentionVariantDefault;  // bits: see oake_debug_set_attention_variant
  int cu_count = 0;        // compute units the launch stream may use (0 = all of the device): a handle driven on a
                           // CU-masked stream (hipExtStreamCreateWithCUMask) sizes its persistent grids to that
  int qkv_walk = kQkvWalkDefault;  // fused qkv + attention kernel: heads per head block of the tile walk (walk_decode)
};

struct GemmArgs {
  const void* A;      // [M,K] row-major 16-bit
  const void* W;      // [N,K] row-major 16-bit
  const float* bias;  // [N] or nullptr
  void* out;
  int M, N, K;
  int ldo;            // leading dimension of out (elements)
  // EPI_PATCH only
  const float* pos;   // [L, N] positional embedding (fp32)
  int P2;             // patches per image
  int L;              // tokens per image (P2 + 1)
  // EPI_*_LN only
  const float* rowstat;  // [M + 1, 2] (rstd, -mean * rstd) of the raw A rows (launch_rowstat; row M is padding)
  const float* colsum;   // [N] sum_k W[n,k]
  // Row statistics handed from GEMM to GEMM when both run the persistent kernel
  // (gemm_uses_persistent): EPI_RESID16 writes (sum x, sum x^2) of each 64-column slice of its
  // output rows to rowpart_out [M, 16, 2] (nullptr: don't); EPI_*_LN reads the first `nparts`
  // slices of rowpart_in [M, 16, 2] instead of `rowstat`.
  float* rowpart_out;
  const float* rowpart_in;
  int nparts;
  const LaunchOpts* opts;  // nullptr = defaults
  // EPI_PATCH16 on the persistent kernel only: A is not a matrix but the 16-bit NCHW image batch itself
  // ([n,3,S,S], conv stride == patch P, no padding, grid G = S / P): row m = (image, py, px) of the implicit
  // im2col matrix, column k = (channel, ky, kx); the DMA waves gather the patch rows straight from the
  // images (8 pixels = 16 B per lane, source-side swizzle as for a matrix).  patch_S == 0: plain matrix A.
  int patch_S, patch_P, patch_G;
  // the general form: A = a zero-padded 16-bit buffer [n,3,patch_H,patch_S] (patch_S = row stride in pixels, a
  // multiple of 8), patch origins patch_T pixels apart (0: patch_T = patch_P, patch_H = patch_S, i.e. the above)
  int patch_T, patch_H;
  // patch_S != 0 in the plain geometry (patch_T == 0) with an FP32 image batch: the DMA waves fetch the patch
  // rows with ordinary 16-byte loads, round them to the 16-bit operand type and write the LDS image themselves
  // (no im2col pass and no 16-bit copy of the batch; EPI_PATCH16 on the persistent kernel only)
  int patch_f32;
};
// true when the conv1 GEMM can read its A operand straight from the NCHW batch (no im2col pass)
bool gemm_patch_direct_ok(int image, int patch, int stride, int padding, int M, int N, int K,
                          const LaunchOpts* opts = nullptr);
// ... and when that batch may be fp32 (cast on the way into LDS by the DMA waves)
bool gemm_patch_f32_ok(int image, int patch, int stride, int padding, int M, int N, int K,
                       const LaunchOpts* opts = nullptr);
// ... or from a zero-padded copy of it (strides that cut patches, padding: objects mode)
bool gemm_patch_padded_ok(int patch, int stride, int M, int N, int K, const LaunchOpts* opts = nullptr);

hipError_t launch_gemm(int dtype16, int epi, const GemmArgs& a, hipStream_t s);
// tile configurations this build of the library carries (production: -1 automatic, 0, 4, 5; lab build: -1 .. 11)
bool gemm_variant_supported(int v);
// true when launch_gemm runs this shape on the persistent kernel (row statistics via rowpart_*)
bool gemm_uses_persistent(int M, int N, int K, const LaunchOpts* opts = nullptr);

// ---- row kernels -------------------------------------------------------------------------
// y(16-bit)[rows,c] = LN(x [rows, c]); x is fp32 (x_dtype DT_F32) or the 16-bit type (x_dtype ==
// dtype16); x rows are `x_row_stride` ELEMENTS apart.  Statistics in fp32 either way.
hipError_t launch_layernorm(int dtype16, const void* x, int x_dtype, long x_row_stride,
                            const float* gamma, const float* beta, void* y, int rows, int c,
                            hipStream_t s);

// stat[row] = (rstd, -mean * rstd) of LayerNorm over x[row, :c]   (the EPI_*_LN GEMM epilogues)
hipError_t launch_rowstat(const void* x, int x_dtype, long x_row_stride, float* stat, int rows, int c,
                          hipStream_t s);
// wf = 16-bit(w32 * gamma[k]) [n_out,k]; colsum[n] = sum_k wf[n,k]; bf[n] = bias[n] + sum_k w32[n,k] beta[k]
hipError_t launch_fold_ln(int dtype16, const float* w32, const float* gamma, const float* beta,
                          const float* bias, void* wf, float* colsum, float* bf, int n_out, int k,
                          hipStream_t s);

// x[n*L + t, :] (in place, fp32 or 16-bit): t == 0 -> cls + pos[0]; then ln_pre over every row.
// rowpart (optional): [n*L, 16, 2] — slot 0 of each row receives (sum, sum of squares) of the output row
hipError_t launch_embed_ln_pre(void* x, int x_dtype, const float* cls, const float* pos,
                               const float* gamma, const float* beta, int n, int L, int c,
                               float* rowpart, hipStream_t s);
// rowpart[row][0] = (sum, sum of squares) of x[row, :c]   (tests / debug entry)
hipError_t launch_rowsums(const void* x, int x_dtype, long x_row_stride, float* rowpart, int rows,
                          int c, hipStream_t s);

// ---- text tower glue (oadp/prompts/vild.py -> clip encode_text) ------------------------------
// x[n*L + t, :] = tok_emb[tokens[n*L + t], :] + pos[t, :]  (residual-stream type);  rowpart as in
// launch_embed_ln_pre (slot 0 = row sums) or nullptr
hipError_t launch_text_embed(const int32_t* tokens, const float* tok_emb, const float* pos, void* x,
                             int x_dtype, int n, int L, int c, int vocab, float* rowpart, hipStream_t s);
// y[i, :] (fp32) = x[i*L + argmax_t tokens[i*L + t], :]   (the EOT position: highest token id, first hit)
hipError_t launch_gather_eot(const int32_t* tokens, const void* x, int x_dtype, float* y, int n, int L,
                             int c, hipStream_t s);

// im2col of NCHW images into the conv1 GEMM A operand [n*G*G, 3*P*P] (16-bit).
hipError_t launch_im2col(int dtype16, const void* img, int in_dtype, void* out, int n, int image,
                         int patch, int stride, int pad, int grid, hipStream_t s);

// zero-padded 16-bit copy [n,3,hp,ws] of an NCHW batch (image at (pad, pad); ws = row stride, a multiple of 8):
// what the conv1 GEMM gathers its patches from when the convolution pads or its stride cuts patches
hipError_t launch_pad_nchw(int dtype16, const void* img, int in_dtype, void* out, int n, int image, int pad, int hp,
                           int ws, hipStream_t s);
// ... and back: the dense 16-bit [n,3,image,image] batch out of the padded one (small passes of a padded-layout call)
hipError_t launch_unpad_nchw(const void* padded, void* out, int n, int image, int pad, int hp, int ws, hipStream_t s);

// ---- attention ---------------------------------------------------------------------------
// qkv [n*L, 3*H*64] 16-bit (q pre-scaled by 1/8) -> out [n*L, H*64] 16-bit. Full self-attention.
// causal != 0: key j is visible to query i only if j <= i (text tower)
// qkv_y / mask / out_y (optional, only when attention_fuses_object_token(L)): the object tokens'
// attention (launch_object_attention's job) done by the same launch.
hipError_t launch_attention(int dtype16, const void* qkv, void* out, int n, int L, int heads,
                            int causal, hipStream_t s, const void* qkv_y = nullptr,
                            const void* mask = nullptr, int mask_dtype = 0, void* out_y = nullptr,
                            const LaunchOpts* opts = nullptr);
bool attention_fuses_object_token(int L, const LaunchOpts* opts = nullptr);
// attention kernel forms this build carries (production: 159 and 31; lab build: every bit set)
bool attention_variant_supported(int v);

// Object-token attention (oadp/oake/objects.py:232-247): one query per crop (qkv_y row n),
// keys/values = patch rows 1..L-1 of qkv_x plus the object token's own k/v (qkv_y);
// additive bias = -100 * mask[n, p] on patch keys, 0 on the object token.
hipError_t launch_object_attention(int dtype16, const void* qkv_x, const void* qkv_y,
                                   const void* mask, int mask_dtype, void* out, int n, int L,
                                   int heads, hipStream_t s, const LaunchOpts* opts = nullptr);

// ---- attention + out_proj + residual in one kernel (attn_out.hip) ------------------------------
// For sequences of at most 64 tokens on a 16-bit residual stream, ViT-B geometry (12 heads, width 768):
// x[n*L + t, :] (16-bit, in place) += softmax(q k^T) v . W_out^T + bias, and rowpart [n*L, 16, 2] receives
// (sum, sum^2) of every 64-column slice of the new rows (12 slices: the LayerNorm statistics of the LN-folded
// c_fc GEMM that follows).  `wperm` = W_out in the kernel's fragment order (launch_permute_out_w).
bool attn_out_supported(int L, int heads, int width);
hipError_t launch_permute_out_w(int dtype16, const void* w, void* wp, hipStream_t s);
// trace (measurement only): device buffer of 4 x 12 x 64 uint64 receiving s_memtime stamps of the first workgroups
hipError_t launch_attn_out(int dtype16, const void* qkv, const void* wperm, const float* bias, void* x,
                           float* rowpart, int n, int L, hipStream_t s, unsigned long long* trace = nullptr);

// ---- LN-folded qkv in-projection + attention in one persistent kernel (qkv_attn.hip) ------------
// Sequences of at most 53 tokens (three images per 160-row tile) on a 16-bit residual stream, 64-wide heads:
// out [n*L, H*64] = softmax(q k^T) v with (q | k | v) = rstd * (x W'^T) + (-mean rstd) * colsum + bias' computed tile by
// tile and never written.  wp / biasp / colsump = the folded in-projection in head-major row order
// (launch_permute_qkv: row h * 192 + 64 m + j <- row m * C + 64 h + j); rowpart / nparts as for the LN-folded GEMMs.
bool qkv_attn_supported(int L, int heads, int width, int n_img);
hipError_t launch_permute_qkv(int dtype16, const void* w, const float* bias, const float* colsum, void* wp, float* biasp,
                              float* colsump, int width, hipStream_t s);
hipError_t launch_qkv_attn(int dtype16, const void* x, const void* wp, const float* biasp, const float* colsump,
                           const float* rowpart, int nparts, void* out, int n_img, int L, int heads,
                           const LaunchOpts* opts, hipStream_t s, unsigned long long* trace = nullptr);

// ... and objects mode (qkv_attn_obj.hip): 192 < L + 1 <= 200 rows per crop = its L tokens + its object token (row T + crop
// of x / rowpart / out, T = n_img * L); mask [n_img, L - 1] (1 = background) of mask_dtype DT_F16 | DT_F32.  out rows
// 0 .. T - 1 = the patch stream's attention, rows T .. T + n - 1 the object tokens' [REF oadp/oake/objects.py:223-247].
// wp / biasp / colsump in the kernel's own column order (launch_permute_qkv_obj).
bool qkv_attn_obj_supported(int L, int heads, int width, int n_img);
// ... and its QUAD form for short sequences: FOUR images of L <= 50 tokens per 208-row tile (plain self-attention per image;
// batch 256 x 12 heads = 768 tiles = exactly three rounds of 256 CUs, where three images per 160-row tile make 1032 = 4.03)
bool qkv_attn_quad_supported(int L, int heads, int width, int n_img);
hipError_t launch_qkv_attn_quad(int dtype16, const void* x, const void* wp, const float* biasp, const float* colsump,
                                const float* rowpart, int nparts, void* out, int n_img, int L, int heads,
                                const LaunchOpts* opts, hipStream_t s, unsigned long long* trace);
hipError_t launch_permute_qkv_obj(const void* w, const float* bias, const float* colsum, void* wp, float* biasp,
                                  float* colsump, int width, hipStream_t s);
hipError_t launch_qkv_attn_obj(int dtype16, const void* x, const void* wp, const float* biasp, const float* colsump,
                               const float* rowpart, int nparts, const void* mask, int mask_dtype, void* out, int n_img,
                               int L, int heads, const LaunchOpts* opts, hipStream_t s, unsigned long long* trace = nullptr);

// ---- head tail ---------------------------------------------------------------------------
// rows of [n, e] fp32 -> optional L2 normalise (F.normalize, eps 1e-12) -> out [n, e] (fp32 or f16)
hipError_t launch_l2norm_rows(const float* in, void* out, int out_dtype, int normalize, int n, int e,
                              hipStream_t s);

// ---- misc --------------------------------------------------------------------------------
hipError_t launch_cast_f32_to_16(int dtype16, const float* in, void* out, size_t numel, float scale,
                                 hipStream_t s);
// scale rows [row0,row1) of a [rows, cols] fp32 matrix / vector prefix in place
hipError_t launch_scale_f32(float* x, size_t numel, float scale, hipStream_t s);

hipError_t launch_crop_normalize(const uint8_t* img, int height, int width, const int32_t* boxes,
                                 int k, int out_size, const float* mean3, const float* inv_std3,
                                 void* out, int out_dtype, hipStream_t s);

// ---- Pillow-exact crop + bicubic resize (resample.hip) --------------------------------------
// One job = one crop of one uint8 HWC image; the jobs of a launch may come from different images
// (a whole flush of a sweep is three launches, not three per image).
struct ResampleJob {
  const uint8_t* img;  // source image (device), uint8 HWC
  int height, width;
  int sx0, sy0;        // crop origin in the source image (may be negative: PIL zero-fills)
  int cw, ch;          // crop size
  int rw, rh;          // size after Resize
  int cx, cy;          // CenterCrop offset into the resized image
  int kh, kv;          // coefficient taps per output index (horizontal / vertical)
  long coefh_off, coefv_off;    // offsets into the int32 coefficient table
  long boundh_off, boundv_off;  // offsets into the int32 bounds table
  long temp_off;       // byte offset of this job's horizontal-pass image (16-byte aligned)
  int tstride;         // bytes between its rows: 12 * ceil(rw / 4) (the horizontal pass stores 12-byte quads)
  long out_row;        // DT_F32 / DT_F16 output: the job's row of `out` [rows,3,out,out]
  uint8_t* u8_out;     // DT_U8 output: this job's rh x rw HWC destination (device)
  // Pass order.  Pillow runs the horizontal pass first — except that a source more than 100 times taller
  // than wide whose vertical pass reduces is resampled vertically first (Pillow 12.2, pinned empirically:
  // tests/test_resample.py).  Such a job is stored TRANSPOSED (tr = 1: cw/ch, rw/rh, sx0/sy0, cx/cy swapped,
  // run_resample does it): the kernels read the source and write the output through the transposition, so
  // "horizontal, then vertical" in job space is vertical, then horizontal in the image.
  int tr;
};
// out_dtype DT_F32 / DT_F16: normalised crops, job j -> out[j.out_row];
// DT_U8: every job writes its own uint8 HWC image (rh x rw) to job.u8_out (`out` unused).
// max_chq_rw / max_rh_rw: the largest ceil(ch / 4) * (rw rounded up to 4) (the horizontal pass: four rows per thread,
// whole quads of columns) / rh * rw over the jobs (grid sizing).
// out_ws > 0 (DT_F16, out_size % 4 == 0): job j writes rows out_pad .. out_pad + out_size - 1, columns out_pad .. of its
// three planes of the zero-padded batch [rows, 3, out_hp, out_ws] conv1 gathers its patches from (objects mode) — the
// padding columns inside the four-pixel groups it touches are written as zeros, everything else of the border is
// expected to BE zero already
hipError_t launch_resample(const ResampleJob* d_jobs, int njobs, int max_out, long max_chq_rw, long max_rh_rw,
                           int32_t* d_coef, int32_t* d_bounds, uint8_t* d_temp, int out_size,
                           const float* mean3, const float* std3, void* out, int out_dtype, hipStream_t s,
                           int out_pad = 0, int out_hp = 0, int out_ws = 0);

// Exact-size crops (no resampling) of many images in one launch: job j -> out[j.out_row].
struct CropJob {
  const uint8_t* img;
  int height, width;
  int x1, y1;          // crop origin (PIL zero fill outside the image)
  long out_row;
};
hipError_t launch_crop_normalize_jobs(const CropJob* d_jobs, int njobs, int out_size, const float* mean3,
                                      const float* std3, void* out, int out_dtype, hipStream_t s);

// ---- baseline JPEG decode (jpeg.hip) ---------------------------------------------------------
enum { JPEG_OK = 0, JPEG_INVALID = 1, JPEG_UNSUPPORTED = 2 };
struct JpegFrame {
  int width, height, ncomp, hmax, vmax, mcux, mcuy;
  int progressive;           // SOF2: coefficients arrive over several scans
  int comp_id[3];            // component identifiers of the frame header (scan headers refer to them)
  int h[3], v[3];            // sampling factors
  int bx[3], by[3];          // blocks per row / column of each (MCU-padded) component plane
  long coef_off[3];          // int16 element offset of each component's [by][bx][64] coefficients
  long plane_off[3];         // byte offset of each component's uint8 plane [by*8][bx*8]
  long total_coefs, total_plane_bytes;
  uint16_t q[3][64];         // quantisation table of each component, natural (row-major) order
};
// header walk only (dimensions, sampling, buffer sizes); host
int jpeg_read_frame(const uint8_t* data, size_t n, JpegFrame* frame, std::string* err);
// host Huffman decode of the single interleaved scan into coefs[total_coefs] (natural order)
int jpeg_decode_coefs(const uint8_t* data, size_t n, const JpegFrame& frame, int16_t* coefs,
                      std::string* err);
// dequantise + IDCT into d_planes[total_plane_bytes], then upsample + colour-convert to HWC RGB
hipError_t launch_jpeg_reconstruct(const JpegFrame& frame, const int16_t* d_coefs, uint8_t* d_planes,
                                   uint8_t* d_out_hwc, hipStream_t s);

hipError_t launch_tr_read_probe(const uint16_t* in, uint16_t* out, hipStream_t s);
// one block per CU (large dynamic LDS), each records (XCC id, HW_ID) and holds its CU for ~hold_us: which compute
// units a stream's blocks land on (CU-masked streams)
hipError_t launch_cu_census(unsigned* out, int nblocks, int hold_us, hipStream_t s);
// register-only MFMA stream on every SIMD (gemm.hip): d_frags = 9 x 64 x 8 halves, *flop = work of the launch
hipError_t launch_mfma_probe(const void* d_frags, float* d_sink, int iters, double* flop, hipStream_t s);
hipError_t launch_mfma_probe_order(const void* d_frags, float* d_sink, int iters, int order, double* flop, hipStream_t s);
hipError_t launch_mfma_probe32(const void* d_frags, float* d_sink, int iters, double* flop, hipStream_t s);  // 32x32x16 form

}  // namespace oake
