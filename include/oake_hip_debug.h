/*
 * oake_hip_debug.h — kernel-level test / measurement entry points of liboake_hip.so.
 *
 * NOT part of the reference-facing ABI (include/oake_hip.h is: the functions a maintainer of
 * LutingWang/OADP binds).  These are what tests/ and tools/ use to check each kernel
 * against the oracle and to measure it in isolation: raw GEMM / LayerNorm / attention launches,
 * thread-local kernel-variant switches for those handle-less launches, cycle-stamp buffers, the MFMA
 * power probe and a weight read-back.  They are exported by the same library; nothing in the product path
 * (oadp_amd/oake, oadp_amd/clip forward calls) uses them.
 */
#ifndef OAKE_HIP_DEBUG_H
#define OAKE_HIP_DEBUG_H

#include "oake_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Kernel-level debug/test entry points (used by tests/ to check each kernel against the
 * oracle; not needed by a reference-side integration).  All pointers are device pointers.
 * dtype16 = OAKE_F16 | OAKE_BF16 selects the 16-bit operand type.
 */
/* C[m,n] = A[m,k] * W[n,k]^T + bias[n] (fp32 out).  A, W are 16-bit row-major. */
OAKE_API int oake_debug_gemm(const void* d_a, const void* d_w, const float* d_bias, float* d_c,
                    int m, int n, int k, int dtype16, void* stream);
/* C16[m,n] = (quick_gelu?)(A * W^T + bias) stored in the 16-bit operand type (n % 8 == 0). */
OAKE_API int oake_debug_gemm16(const void* d_a, const void* d_w, const float* d_bias, void* d_c,
                      int m, int n, int k, int dtype16, int gelu, void* stream);
/* C16[m,n] = (quick_gelu?)(LayerNorm(x)[m,k] * W^T + bias) with the LayerNorm folded into the GEMM
 * (gamma into W, beta into bias, per-row statistics applied in the epilogue); x is 16-bit [m,k],
 * W fp32 [n,k].  Synchronous (allocates temporaries). */
OAKE_API int oake_debug_ln_gemm16(const void* d_x, const float* d_w32, const float* d_gamma,
                         const float* d_beta, const float* d_bias, void* d_c, int m, int n, int k,
                         int dtype16, int gelu, void* stream);
/* out16[n_img*l, heads*64] = softmax(q k^T) v per (image, head) with (q | k | v) = LayerNorm(x) W^T + bias rounded to 16
 * bits, as ONE kernel (csrc/qkv_attn.hip: the q / k / v values never reach memory); x 16-bit [n_img*l, heads*64], W fp32
 * [3*heads*64, heads*64] (q rows pre-scaled by the caller), l <= 53 else OAKE_ERR_UNSUPPORTED.  Synchronous.
 * d_trace (or NULL): device buffer of 64 x 3 x 6 x 8 uint64 receiving the cycle stamps of the first blocks' tile phases
 * (tools/qkv_attn_trace.py); repeats: launches of the kernel (timing loops). */
OAKE_API int oake_debug_ln_qkv_attn(const void* d_x, const float* d_w32, const float* d_gamma, const float* d_beta,
                           const float* d_bias, void* d_out, int n_img, int l, int heads, int dtype16, void* d_trace,
                           int repeats, void* stream);
/* The same contract as oake_debug_ln_qkv_attn through the 208-row tile kernel's QUAD form (csrc/qkv_attn_obj.hip: four images
 * of l <= 50 tokens per tile).  d_trace (or NULL): 64 x 3 x 6 x 8 uint64 cycle stamps. */
OAKE_API int oake_debug_ln_qkv_attn_quad(const void* d_x, const float* d_w32, const float* d_gamma, const float* d_beta,
                                         const float* d_bias, void* d_out, int n_img, int l, int heads, int dtype16,
                                         void* d_trace, int repeats, void* stream);
/* Objects mode (csrc/qkv_attn_obj.hip): x 16-bit [n_img*l + n_img, heads*64] = the crops' token rows, then one object-token
 * row per crop; mask [n_img, l-1] (OAKE_F16 | OAKE_F32, 1 = background).  out16 rows 0 .. n_img*l-1 = self-attention of the
 * token rows over their crop's tokens, rows n_img*l .. = the object tokens over their crop's patch rows (-100 * mask) and
 * themselves — with (q | k | v) = LayerNorm(x) W^T + bias rounded to 16 bits, as ONE kernel.  192 < l + 1 <= 200 else
 * OAKE_ERR_UNSUPPORTED.  Synchronous.  d_trace (or NULL): 64 x 3 x 6 x 8 uint64 cycle stamps of the first blocks' tile phases. */
OAKE_API int oake_debug_ln_qkv_attn_obj(const void* d_x, const float* d_w32, const float* d_gamma, const float* d_beta,
                               const float* d_bias, const void* d_mask, int mask_dtype, void* d_out, int n_img, int l,
                               int heads, int dtype16, void* d_trace, int repeats, void* stream);
/* y = LayerNorm(x) over last dim `c` (eps 1e-5), x [rows,c] of x_dtype (OAKE_F32 or dtype16)
 * -> y 16-bit [rows,c]. */
OAKE_API int oake_debug_layernorm(const void* d_x, int x_dtype, const float* d_gamma, const float* d_beta,
                         void* d_y, int rows, int c, int dtype16, void* stream);
/* Multi-head self-attention on packed qkv [n*l, 3*heads*64] (q pre-scaled), -> [n*l, heads*64]. */
OAKE_API int oake_debug_attention(const void* d_qkv, void* d_out, int n, int l, int heads,
                         int dtype16, void* stream);
/* The same with the objects-mode object token fused in (oadp/oake/objects.py:232-247): qkv_y [n, 3*heads*64] holds
 * one extra query / key / value per sequence; its keys are the sequence's rows 1..l-1 with the additive bias
 * -100 * mask[n, l-1] (mask_dtype OAKE_F32 or OAKE_F16) plus its own key; d_out_y [n, heads*64].
 * OAKE_ERR_UNSUPPORTED when the selected kernel form has no place for the token at this l. */
OAKE_API int oake_debug_attention_objects(const void* d_qkv, const void* d_qkv_y, const void* d_mask, int mask_dtype,
                                 void* d_out, void* d_out_y, int n, int l, int heads, int dtype16, void* stream);
/* The fused form for sequences of at most 64 tokens (csrc/attn_out.hip; heads = 12 only, else
 * OAKE_ERR_UNSUPPORTED): x[n*l, heads*64] (16-bit, in place) += attention(qkv) * W^T + bias with W [heads*64,
 * heads*64] row-major 16-bit, and d_rowpart [n*l, 16, 2] fp32 receives (sum, sum of squares) of every 64-column
 * slice of the new rows.  Synchronous (permutes W into a temporary). */
OAKE_API int oake_debug_attn_out(const void* d_qkv, const void* d_w, const float* d_bias, void* d_x,
                        float* d_rowpart, int n, int l, int heads, int dtype16, void* stream);
/* The same, launched `repeats` times back to back (x keeps accumulating); with d_trace != NULL (f16 only) the
 * measurement build of the kernel runs and d_trace [4 workgroups][12 waves][64] uint64 receives s_memtime stamps at
 * the kernel's phase boundaries (tools/attn_out_trace.py). */
OAKE_API int oake_debug_attn_out_trace(const void* d_qkv, const void* d_w, const float* d_bias, void* d_x,
                              float* d_rowpart, int n, int l, int heads, int dtype16, void* d_trace, int repeats,
                              void* stream);
/* Raw ds_read_b64_tr_b16 semantics probe: in = 256 uint16, out = 64 lanes x 4 uint16. */
OAKE_API int oake_debug_tr_read(const uint16_t* d_in, uint16_t* d_out, void* stream);
/* Which compute units do a stream's blocks land on (CU-masked streams)?  nblocks blocks of one wave, one per CU
 * (96 KiB of LDS each), each holding its CU for hold_us microseconds; d_out[2 b] = XCC id, d_out[2 b + 1] = the
 * HW_ID register (CU / SH / SE fields) of block b. */
OAKE_API int oake_debug_cu_census(uint32_t* d_out, int nblocks, int hold_us, void* stream);
/* Measurement: a register-only MFMA stream (two waves per SIMD on every CU, `iters` x 20
 * v_mfma_f32_16x16x32_f16 per wave, no LDS or memory traffic) on the 9 x 64 x 8 f16 operand fragments at
 * d_frags16; *flop (host, may be NULL) receives the FLOPs of the launch.  Timed by the caller, it gives the
 * matrix rate the board sustains under its power cap for that operand data (bench.py `roofline.sustained`). */
OAKE_API int oake_debug_mfma_probe(const void* d_frags16, float* d_sink, int iters, double* flop, void* stream);
/* The same with v_mfma_f32_32x32x16_f16 (`iters` x 10 per wave on the first 7 x 64 x 8 fragments: half the A / B operand
 * reads per FLOP, twice the accumulator traffic) — tools/mfma_shape_probe.py: which shape the board sustains more of. */
/* ... and the 16x16x32 stream over a 10 x 4 wave tile (160 accumulator registers, `iters` x 40 MFMAs per wave) in three
 * orders: 0 = row by row, 1 = serpentine, 2 = column by column, 3 = column by column serpentine (csrc/gemm.hip
 * mfma_probe_order_kernel). */
OAKE_API int oake_debug_mfma_probe_order(const void* d_frags16, float* d_sink, int iters, int order, double* flop,
                                         void* stream);
OAKE_API int oake_debug_mfma_probe_32x32(const void* d_frags16, float* d_sink, int iters, double* flop, void* stream);
/* Attention variant bits: 1 = ds_read_b64_tr_b16 V fragments (else 16-bit LDS gathers),
 * 2 = 32 queries per wave (else 64), 4 = sequences longer than 64 keys share K / V through LDS between
 * the four waves of a block, 8 = objects mode: the object token's attention rides on an idle wave
 * of that kernel, 16 = sequences of at most 64 keys without a causal mask: the two waves of a (crop,
 * head) share its K / V in LDS, staged with LDS-DMA, 32 = sequences longer than 128 keys: one block of
 * eight waves per (crop, head) reads K / V once (else two blocks of four read them twice; measured slower,
 * so not in the default), 64 = sequences of 65..208 keys: the whole K / V of a (crop, head) is brought into
 * LDS up front with LDS-DMA and the key loop runs without barriers or global accesses (measured equal, so not
 * in the default either), 128 = sequences of 193..208 keys without a causal mask (objects mode, 197): one block per
 * (crop, head), the whole score matrix of a 32-query unit in registers, one-pass softmax (attention_head.inc).
 * Default 159; the production library also accepts 31 (the cooperative kernel at 197 keys, for A/B runs). */
OAKE_API int oake_debug_set_attention_variant(int variant);
/* 1 in liboake_hip_lab.so (built with -DOAKE_LAB=1: the production kernels plus every tile configuration, kernel form
 * and measurement epilogue that lost its A/B), 0 in the production library, whose oake_debug_set_* / oake_set_option
 * refuse the lab-only values (OAKE_ERR_UNSUPPORTED / OAKE_ERR_INVALID). */
OAKE_API int oake_debug_lab_build(void);
/* The pass size oake_encode_image / oake_encode_objects cut a call of n crops into (csrc/api.hip plan_pass_size: the
 * cheapest of "passes at the cap + a shorter last one" and k, k + 1, k + 2 equal passes by tile rounds on `compute_units`
 * CUs; cap = the handle's crops per pass, tokens = rows per crop, images_per_attention_tile = 4 for sequences of <= 50
 * tokens, else 1).  Host arithmetic only: no device is touched. */
OAKE_API int oake_debug_plan_pass(int cap, int n, int tokens, int images_per_attention_tile, int width, int mlp_dim,
                                  int heads, int compute_units);
/* GEMM configuration: -1 = automatic per shape (-2: without the 320-row tile), 0..13 = forced (see csrc/gemm.hip; production build: -1, 0, 4, 5, 13).
 * All oake_debug_set_* switches are THREAD-LOCAL and affect only the handle-less oake_debug_* kernel
 * entry points of the calling thread; a handle's own switches are set with oake_set_option. */
OAKE_API int oake_debug_set_gemm_variant(int variant);
/* x[m,n] (16-bit, in place) += A * W^T + bias — the residual epilogue of out_proj / c_proj.  On the
 * persistent kernel (large m) d_rowpart [m, 16, 2] fp32 (or NULL) receives (sum, sum of squares) of
 * every 64-column slice of each output row (the LayerNorm statistics handed to the next GEMM). */
OAKE_API int oake_debug_gemm_resid16(const void* d_a, const void* d_w, const float* d_bias, void* d_x,
                            float* d_rowpart, int m, int n, int k, int dtype16, void* stream);
/* GEMM tile order: 0 = default, n > 0 = N panels of n tiles (row-major inside), n < 0 = M slabs of
 * -n tiles (column-major inside). */
OAKE_API int oake_debug_set_gemm_panel(int panel);
/* Tile walk of the fused qkv + attention kernel for the calling thread's oake_debug_ln_qkv_attn_* calls: heads per head
 * block (OAKE_OPT_QKV_WALK's meaning; 0 = group-major).  Placement only: the results do not depend on it. */
OAKE_API int oake_debug_set_qkv_walk(int heads_per_block);
/* Debug: device buffer of 4608 uint64 receiving per-tile cycle stamps of the production GEMM
 * (entry, tile start, epilogue start, epilogue end; then per-block wall-clock entry/exit), or NULL. */
OAKE_API int oake_debug_set_gemm_trace(void* d_trace);

/* Test hook: copy a 16-bit matmul weight back from the device, as uploaded (f32 -> 16 bit; q rows of
 * in_proj scaled by 1/8; visual.proj / text_projection transposed to [embed, width]).  `name` is the
 * state-dict key ("visual.conv1.weight", "...attn.in_proj_weight", "...attn.out_proj.weight",
 * "...mlp.c_fc.weight", "...mlp.c_proj.weight", "visual.proj"); "<key>#folded" reads the gamma-folded
 * copy of in_proj / c_fc.  numel must match the tensor. */
OAKE_API int oake_debug_read_weight16(oake_handle* h, const char* name, uint16_t* h_out, size_t numel);

#ifdef __cplusplus
}
#endif

#endif /* OAKE_HIP_DEBUG_H */
