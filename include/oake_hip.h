/*
 * oake_hip.h — C ABI of liboake_hip.so: the MI355X (gfx950) implementation of OADP's
 * OAKE CLIP image-encoder hot path.
 *
 * The reference (LutingWang/OADP) has no native FFI for this path: it sits behind two
 * Python-level boundaries.  This header is the C boundary that the
 * Python host layer (oadp_amd/clip/model.py) binds with ctypes; each entry point names
 * the reference interface it replaces.
 *
 *   reference call site                                   -> entry point here
 *   ---------------------------------------------------------------------------
 *   clip.load_default(...)            oadp/oake/globals.py:47, blocks.py:123,
 *                                     objects.py:290       -> oake_create + oake_load_tensor*
 *   model.encode_image(image)         oadp/oake/globals.py:57, blocks.py:129
 *                                                          -> oake_encode_image
 *   model.visual(objects, masks)      oadp/oake/objects.py:330 (+ Hooks, objects.py:198-266,
 *     after Validator._build_model      surgery objects.py:285-314)
 *                                                          -> oake_encode_objects
 *   F.normalize(embedding).half()     oadp/oake/globals.py:58-59, blocks.py:130-132,
 *                                     objects.py:331,334   -> `normalize` / `out_dtype` args
 *   preprocess(image.crop(box))       oadp/oake/blocks.py:79-81, objects.py:116-127
 *                                                          -> oake_crop_normalize (exact-size crops),
 *                                                             oake_crop_resize_normalize (resampled)
 *   image.resize((w/1.5, h/1.5))      oadp/oake/blocks.py:72-76 -> oake_resize_u8
 *   Dataset._preprocess (blocks)      oadp/oake/blocks.py:54-109 -> oake_blocks_batch (pyramid + every block
 *                                                             crop of a whole flush of images)
 *
 * Conventions: plain C, int status (0 = OAKE_OK), no exception crosses the ABI.  All data
 * pointers named d_* are DEVICE pointers owned by the caller (e.g. a torch tensor's
 * data_ptr()); pointers named h_* are host pointers.  Work is enqueued on the hipStream_t
 * passed as `stream` (an opaque void* here so the header needs no HIP include; 0 = the
 * null stream).  A handle is bound to one device, owns weights + workspace, and is not
 * thread-safe (the reference runs one process per GPU and calls the model only from the
 * main thread: oadp/oake/base.py:122-126).  Distinct handles may be driven from distinct host
 * threads (tests/test_encoder_gpu.py::test_two_handles_on_two_host_threads).
 */
#ifndef OAKE_HIP_H_
#define OAKE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OAKE_ABI_VERSION 4

#if defined(__GNUC__)
#define OAKE_API __attribute__((visibility("default")))
#else
#define OAKE_API
#endif

/* status codes */
enum {
  OAKE_OK = 0,
  OAKE_ERR_INVALID = 1,   /* bad argument / shape / dtype */
  OAKE_ERR_HIP = 2,       /* a HIP runtime call failed; see oake_last_error */
  OAKE_ERR_STATE = 3,     /* e.g. encode before all weights were loaded */
  OAKE_ERR_UNKNOWN_TENSOR = 4,
  OAKE_ERR_UNSUPPORTED = 5  /* valid input outside the implemented subset (e.g. progressive JPEG) */
};

/* element types for image inputs / embedding outputs */
enum {
  OAKE_F32 = 0,
  OAKE_F16 = 1,
  OAKE_BF16 = 2,
  OAKE_U8 = 3
};
/* Layout flag, OR-ed into the `out_dtype` of oake_crop_resize_normalize(_batch) and the `in_dtype` of oake_encode_objects on
 * a handle whose conv1 pads (objects mode: stride 16, padding 15 [REF oadp/oake/objects.py:298-301]): the crops are written
 * into / read from the ZERO-PADDED 16-bit batch [n, 3, rows, row_stride] conv1 gathers its patches from (oake_padded_layout:
 * the image at (padding, padding) of each plane) instead of a dense [n,3,S,S] tensor, and the library's own pad pass over
 * the batch is not run.  The caller creates the buffer zero-filled once; the writer keeps every border element zero. */
#define OAKE_LAYOUT_PADDED 0x100

typedef struct oake_handle oake_handle;

/*
 * Architecture + patch-embedding geometry.  ViT-B/32 defaults: image 224,
 * patch 32, width 768, layers 12, heads 12 (head_dim must be 64), mlp 3072, embed 512.
 * `stride`/`padding` describe conv1: 32/0 for encode_image; objects mode uses the
 * reference's surgery (oadp/oake/objects.py:298-301): stride = 32 // upsample = 16,
 * padding = (32 - 1) // 2 = 15, which makes grid = 14 and 197 tokens.
 * `compute_dtype`: OAKE_F16 (reference GPU dtype, default) or OAKE_BF16 — the 16-bit type fed
 * to the MFMA units; accumulation, LayerNorm statistics and softmax are fp32.  The residual stream
 * x has its own element type, `residual_dtype` below.
 */
typedef struct oake_config {
  int32_t image_size;
  int32_t patch_size;
  int32_t stride;
  int32_t padding;
  int32_t width;
  int32_t layers;
  int32_t heads;
  int32_t mlp_dim;
  int32_t embed_dim;
  int32_t compute_dtype;
  int32_t max_batch;      /* workspace is sized for this many crops per internal pass at most (the vision tower caps
                             it further by `pass_rows` below; a call with more crops is cut into EQUAL passes;
                             csrc/api.hip) */
  int32_t residual_dtype; /* element type of the residual stream x: == compute_dtype (default; the
                             reference's GPU model keeps x in fp16 too) or OAKE_F32 */
  int32_t pass_rows;      /* vision tower: token rows per internal pass the crops-per-pass cap is derived from —
                             min(max_batch, pass_rows / tokens rounded down to 32, at least 32).  0 = the library's
                             default, 25 600 rows (a pass's activations then turn over inside the 256-MB Infinity
                             Cache); < 0 = no row cap (max_batch alone).  The library reads NO environment variable:
                             the Python host maps OAKE_PASS_ROWS / OAKE_PASS_CROPS onto this field and max_batch
                             (oadp_amd/clip/model.py); the pass size changes the tile shapes of the small last-layer
                             GEMMs and therefore the rounding of the results (<= 3e-4 on the unit-norm output) */
} oake_config;

OAKE_API uint32_t oake_abi_version(void);

/* Fill *cfg with ViT-B/32 defaults (stride 32, padding 0, f16 compute + f16 residual, max_batch 256, pass_rows 0). */
OAKE_API void oake_default_config(oake_config* cfg);

/* Create a handle on HIP device `device`.  Allocates weights + workspace. */
OAKE_API int oake_create(const oake_config* cfg, int device, oake_handle** out);
OAKE_API void oake_destroy(oake_handle* h);

/* Last error text for this handle (or for a failed oake_create when h == NULL). */
OAKE_API const char* oake_last_error(const oake_handle* h);

/* Derived geometry: grid = (image + 2*padding - patch)/stride + 1, tokens = grid*grid + 1. */
OAKE_API int oake_grid(const oake_handle* h);
OAKE_API int oake_tokens(const oake_handle* h);
/* Geometry of the zero-padded 16-bit batch this handle's conv1 gathers its patches from (OAKE_LAYOUT_PADDED): planes of
 * `rows` x `row_stride` pixels, the image at (padding, padding).  OAKE_ERR_UNSUPPORTED for handles whose conv1 does not
 * pad / cut patches (encode_image at stride == patch), for fp32 residual streams and text handles. */
OAKE_API int oake_padded_layout(const oake_handle* h, int* padding, int* rows, int* row_stride);

/*
 * Upload one tensor of the OpenAI-CLIP state_dict by its key (the same keys the reference's
 * `clip` package loads, SURVEY.md §7 hard part 1), e.g. "visual.conv1.weight",
 * "visual.class_embedding", "visual.positional_embedding" (must have oake_tokens() rows —
 * the host interpolates it for objects mode as objects.py:292-296 does), "visual.ln_pre.weight",
 * "visual.transformer.resblocks.3.attn.in_proj_weight", ..., "visual.ln_post.bias",
 * "visual.proj".  `h_data` is a HOST pointer to `numel` contiguous fp32 values in the
 * state_dict's own layout.  Synchronous.  Keys outside the vision tower are rejected with
 * OAKE_ERR_UNKNOWN_TENSOR.
 */
OAKE_API int oake_load_tensor(oake_handle* h, const char* name, const float* h_data, size_t numel);

/* Number of tensors still missing before encode may be called (0 = ready). */
OAKE_API int oake_missing_tensors(const oake_handle* h);

/*
 * model.encode_image(images): d_images is [n,3,image,image] NCHW contiguous of `in_dtype`
 * (OAKE_F32 | OAKE_F16 | OAKE_BF16), already CLIP-normalised.  d_out is [n,embed_dim] of
 * `out_dtype` (OAKE_F32 | OAKE_F16).  normalize != 0 fuses the callers'
 * F.normalize(embedding) (L2 over dim 1, eps 1e-12) before the output cast.
 * n may exceed max_batch (processed in passes).  n == 0 is a no-op.
 */
OAKE_API int oake_encode_image(oake_handle* h, const void* d_images, int in_dtype, int n,
                      void* d_out, int out_dtype, int normalize, void* stream);

/*
 * model.visual(objects, masks) after the reference's objects-mode surgery + Hooks: the
 * object-token stream y is returned (ln_post + proj applied).  Requires stride/padding such
 * that grid*grid == mask elements per crop.  d_masks is [n,1,grid,grid] of `mask_dtype`
 * (OAKE_F32 | OAKE_F16), 1 = background, 0 = object (objects.py:129-155); the additive
 * attention bias is -100*mask for patch keys and 0 for the object token itself
 * (objects.py:209-213).
 */
OAKE_API int oake_encode_objects(oake_handle* h, const void* d_objects, int in_dtype,
                        const void* d_masks, int mask_dtype, int n,
                        void* d_out, int out_dtype, int normalize, void* stream);

/*
 * GPU half of `preprocess(image.crop(box))` for crops that need NO resampling (the 224x224 blocks
 * of oadp/oake/blocks.py:79-81) of a uint8 HWC (interleaved RGB) device image: each of the k boxes
 * (x1,y1,x2,y2 int32, PIL crop semantics: zero fill outside the image) must be exactly
 * out_size x out_size; the result is a [k,3,out,out] NCHW tensor of `out_dtype`, scaled by 1/255 and
 * normalised with mean/std (3 floats each, host pointers), bit-exact w.r.t. ToTensor + Normalize in
 * fp32.  Crops that need resampling go through oake_crop_resize_normalize.
 */
OAKE_API int oake_crop_normalize(oake_handle* h, const uint8_t* d_image_hwc, int height, int width,
                        const int32_t* d_boxes_xyxy, int k, int out_size,
                        const float* h_mean3, const float* h_std3,
                        void* d_out, int out_dtype, void* stream);

/*
 * `preprocess(image.crop(box))` of the reference's DataLoader workers, on the device and bit-exact
 * with Pillow/torchvision (oadp/oake/objects.py:116-127, globals.py:26-33): for each of k boxes
 * (x1,y1,x2,y2 HOST floats) PIL Image.crop (round-half-even coordinates, zero fill outside the
 * image), Resize(out_size, BICUBIC) — Pillow's antialiased fixed-point two-pass resampler — CenterCrop,
 * ToTensor and Normalize(mean, std).  squash != 0 resizes straight to out_size x out_size instead
 * (our reading of clip.load_default(True)).  d_out is [k,3,out,out] NCHW of out_dtype (F32|F16).
 */
OAKE_API int oake_crop_resize_normalize(oake_handle* h, const uint8_t* d_image_hwc, int height, int width,
                               const float* h_boxes_xyxy, int k, int out_size, int squash,
                               const float* h_mean3, const float* h_std3,
                               void* d_out, int out_dtype, void* stream);
/* The same for the images of one flush in ONE call (a sweep otherwise pays an interpreter round trip
 * per image): image i (d_images[i], heights[i] x widths[i]) contributes counts[i] boxes, taken in order
 * from h_boxes_xyxy; d_out is [sum counts, 3, out, out], image after image — or, with out_dtype = OAKE_F16 |
 * OAKE_LAYOUT_PADDED, the zero-padded batch [sum counts, 3, rows, row_stride] of oake_padded_layout (out == the handle's
 * image size; bit-identical pixels, csrc/resample.hip resample_v4p_kernel). */
OAKE_API int oake_crop_resize_normalize_batch(oake_handle* h, int n_images, const uint8_t* const* d_images,
                                     const int* heights, const int* widths, const float* h_boxes_xyxy,
                                     const int* counts, int out_size, int squash, const float* h_mean3,
                                     const float* h_std3, void* d_out, int out_dtype, void* stream);

/* The same for the images of one flush: three launches in all (every box of every image is one job of
 * the same coefficient / horizontal / vertical kernels). */

/*
 * Blocks mode in one call: for each of the n_images uint8 HWC RGB device images, what
 * oadp/oake/blocks.py:89-109 (Dataset._preprocess) hands to the encoder —
 *   row 0      preprocess(image)                      Resize(block_size, BICUBIC) + CenterCrop + Normalize
 *   then       for every pyramid level (level 0 = the image, level k+1 = level k resized with Pillow's
 *              bicubic filter to (int(w / rescale), int(h / rescale)), until a side is shorter than
 *              block_size): the block_size x block_size crops at itertools.product(_partition(w),
 *              _partition(h)) (x outer, y inner), each ToTensor + Normalize
 * written image after image into d_out [sum of counts, 3, block_size, block_size] of out_dtype (F32|F16).
 * counts_out[i] (optional) receives the number of rows of image i (= oake_blocks_count).  Pixels are
 * bit-exact with the PIL / torchvision path; the index arithmetic is integer-exact with the reference's
 * _partition / _partitions.  Work is batched by level across images: a flush costs about 4 launches per
 * pyramid level, whatever the number of images.
 */
OAKE_API int oake_blocks_batch(oake_handle* h, int n_images, const uint8_t* const* d_images, const int* heights,
                      const int* widths, int block_size, int max_stride, double rescale,
                      const float* h_mean3, const float* h_std3, void* d_out, int out_dtype,
                      int* counts_out, void* stream);
/* Rows oake_blocks_batch produces for a width x height image (1 + tiles of every level); host only,
 * -1 for invalid arguments. */
OAKE_API int oake_blocks_count(int width, int height, int block_size, int max_stride, double rescale);

/* PIL Image.resize((dw, dh)) (default BICUBIC) of a uint8 HWC RGB device image — the pyramid step of
 * oadp/oake/blocks.py:72-76 — bit-exact with Pillow. */
OAKE_API int oake_resize_u8(oake_handle* h, const uint8_t* d_src_hwc, int sh, int sw,
                   uint8_t* d_dst_hwc, int dh, int dw, void* stream);

/*
 * Text tower: model.encode_text(tokens) of oadp/prompts/vild.py:62-66 — the
 * same transformer blocks with a causal attention mask.  Tensor names as in the CLIP state dict:
 * token_embedding.weight [vocab,width], positional_embedding [context,width],
 * transformer.resblocks.<l>.*, ln_final.{weight,bias}, text_projection [width,embed].
 * oake_encode_text: d_tokens int32 [n, length] (length <= context: the fork's adaptively_tokenize
 * trims the context, which a causal model allows); the feature is taken at the position of the
 * highest token id of each row (the EOT token), as text.argmax(dim=-1) does.
 */
typedef struct oake_text_config {
  int32_t context, vocab, width, layers, heads, mlp_dim, embed_dim, compute_dtype, max_batch;
  int32_t reserved[3];
} oake_text_config;
OAKE_API void oake_text_default_config(oake_text_config* cfg); /* CLIP ViT-B/32 text: 77, 49408, 512, 12, 8, 2048, 512 */
OAKE_API int oake_text_create(const oake_text_config* cfg, int device, oake_handle** out);
OAKE_API int oake_encode_text(oake_handle* h, const int32_t* d_tokens, int n, int length, void* d_out,
                     int out_dtype, int normalize, void* stream);

/*
 * Baseline JPEG -> uint8 HWC RGB on the device, bit-identical to
 * PIL.Image.open(...).convert('RGB') (libjpeg-turbo defaults: islow IDCT, fancy upsampling) — the
 * decode of torchvision CocoDetection._load_image behind oadp/oake/base.py:53.  The Huffman pass runs
 * on the calling host thread, IDCT / upsampling / colour conversion on `stream`.  h_data is a HOST
  * buffer holding the whole file.  Sequential and progressive Huffman-coded frames are covered;
 * arithmetic-coded / CMYK / 12-bit files return OAKE_ERR_UNSUPPORTED (decode those with PIL and upload
 * the pixels instead).
 * oake_jpeg_info needs no handle and no GPU.
 */
OAKE_API int oake_jpeg_info(const uint8_t* h_data, size_t nbytes, int* height, int* width, int* components);
/* oake_jpeg_info for n files: heights[i] / widths[i] and status[i] (OAKE_OK, OAKE_ERR_UNSUPPORTED or
 * OAKE_ERR_INVALID) per file; returns OAKE_OK unless an argument is NULL. */
OAKE_API int oake_jpeg_info_batch(int n, const uint8_t* const* h_datas, const size_t* nbytes, int* heights,
                         int* widths, int* status);
OAKE_API int oake_decode_jpeg(oake_handle* h, const uint8_t* h_data, size_t nbytes, uint8_t* d_out_hwc,
                     size_t out_capacity, int* height, int* width, void* stream);
/* A batch of files in one call: the Huffman passes run concurrently on `threads` host threads inside
 * the library (one image per task), then one upload and the GPU half per image on `stream`.
 * status[i] = OAKE_OK / OAKE_ERR_UNSUPPORTED / OAKE_ERR_INVALID per image (the call itself fails only
 * for HIP errors); heights / widths may be NULL. */
OAKE_API int oake_decode_jpeg_batch(oake_handle* h, int n, const uint8_t* const* h_datas, const size_t* nbytes,
                           uint8_t* const* d_outs, const size_t* capacities, int* heights, int* widths,
                           int* status, int threads, void* stream);
/* The two halves separately, so that the serial half can run in many worker processes:
 * oake_jpeg_entropy_decode — host only, no handle, no GPU: the quantised DCT coefficients of every
 *   component, [component][block row][block column][64] in natural order, MCU-padded (h_coefs NULL:
 *   size query through *total);
 * oake_jpeg_reconstruct — uploads such coefficients (h_data is needed again for the frame header and
 *   quantisation tables) and runs IDCT / upsampling / colour conversion on `stream`. */
OAKE_API int oake_jpeg_entropy_decode(const uint8_t* h_data, size_t nbytes, int16_t* h_coefs, size_t capacity,
                             size_t* total);
OAKE_API int oake_jpeg_reconstruct(oake_handle* h, const uint8_t* h_data, size_t nbytes, const int16_t* h_coefs,
                          size_t ncoefs, uint8_t* d_out_hwc, size_t out_capacity, int* height,
                          int* width, void* stream);

/*
 * Per-kernel timing with HIP events on the launch stream (bench.py's `roofline` object).
 * enable=1 brackets every kernel launch with events; oake_profile_read synchronises and
 * returns, for up to `cap` kernel slots, name / total milliseconds / launch count / flops.
 * enable=S > 1 stamps only every S-th launch (counted over all kernels since the call).  `launches` counts the
 * stamped launches (what total_ms, flops and bytes cover), `seen` all launches since the reset.  Caveat, measured:
 * the begin stamp of a kernel whose predecessor carries no stamp is taken at dispatch, before that predecessor has
 * drained — sparse durations can add up to MORE than the wall clock; bench.py stamps every launch.
 * Profiling perturbs launches — never leave it on inside a throughput measurement.
 */
typedef struct oake_profile_entry {
  char name[48];
  double total_ms;
  double flops;      /* algorithmic FLOPs summed over the launches (2 per MAC), 0 for non-GEMM */
  double bytes;      /* algorithmic bytes summed over the launches (HBM-bound kernels) */
  int64_t launches;  /* stamped launches */
  int64_t seen;      /* all launches since oake_profile_reset */
} oake_profile_entry;

OAKE_API int oake_profile_enable(oake_handle* h, int enable);
OAKE_API int oake_profile_read(oake_handle* h, oake_profile_entry* entries, int cap, int* count);
OAKE_API int oake_profile_reset(oake_handle* h);

/*
 * Per-handle switches (nothing process-wide: two handles / lanes never see each other's settings).
 *   OAKE_OPT_CLS_LAST           oake_encode_image computes the last block for the CLS rows only (the only
 *                               rows ln_post reads): K / V projections of all tokens, everything else of
 *                               that block for one row per image.  0 = run the block for every token as
 *                               the reference does (A/B runs, tests).  Default 1.
 *   OAKE_OPT_GEMM_VARIANT       -1 = automatic per shape (default; -2 = the same, 160-row tiles only: A/B runs); forced: 0, 4, 5, 13 in the production library, 0..13 in
 *                               the lab build liboake_hip_lab.so (csrc/gemm.hip).  Other values: OAKE_ERR_INVALID.
 *   OAKE_OPT_GEMM_PANEL         GEMM tile order: 0 = default, n > 0 = N panels of n tiles (row-major inside),
 *                               n < 0 = M slabs of -n tiles (column-major inside).  Default 0.
 *   OAKE_OPT_ATTENTION_VARIANT  bit set of attention kernel forms (documented with
 *                               oake_debug_set_attention_variant, oake_hip_debug.h).  Default 159; the production
 *                               library accepts 159 and 31 (OAKE_ERR_INVALID otherwise); the lab build takes 0..255.
 *   OAKE_OPT_PATCH_DIRECT       conv1 gathers its patch rows straight from a 16-bit NCHW input batch (no
 *                               im2col pass) where the geometry allows it.  0 = always im2col.  2 = also from an
 *                               FP32 batch (patch 32): the GEMM's DMA waves load, round and write the LDS image
 *                               themselves (bit-identical; measured slower than im2col with two lanes: lab build only,
 *                               the production library answers OAKE_ERR_INVALID).  Default 1.
 *   OAKE_OPT_CU_COUNT           compute units the caller's stream may use: a handle driven on a CU-masked stream
 *                               (hipExtStreamCreateWithCUMask: two lanes on disjoint halves of the chip) sizes the
 *                               grids of its persistent kernels to that.  0 = every CU of the device.  Default 0.
 *   OAKE_OPT_FUSE_ATTN_OUT      sequences of at most 64 tokens on a 16-bit residual stream (encode_image at 224^2,
 *                               blocks mode) in the ViT-B geometry (12 heads x 64): attention + out_proj + residual +
 *                               the next LayerNorm's row statistics run as ONE kernel, one workgroup per image, and the
 *                               attention output never reaches memory (csrc/attn_out.hip).  Parity-tested; measured
 *                               slower than the two separate launches (39 vs 36 us per layer at batch 256: the
 *                               out_proj weights stream through each CU's 64 B/clk vector-memory path once per
 *                               image, docs/history/round4.md): the kernel is in the lab build liboake_hip_lab.so only; the
 *                               production library answers OAKE_ERR_INVALID to a non-zero value.  Default 0.
 *   OAKE_OPT_FUSE_QKV_ATTN      on a 16-bit residual stream, ln_1 + attn.in_proj + softmax(q k^T) v of a layer run as
 *                               ONE persistent kernel: the q | k | v values go from the MFMA accumulators through LDS
 *                               into the attention and never reach memory.  Same 16-bit q / k / v values as the
 *                               two-launch form.  0: off.  1 (default): sequences of at most 50 tokens (encode_image at
 *                               224^2 / patch 32, blocks mode) as tiles of (four images, one head) of 208 rows, 51 .. 53
 *                               tokens as tiles of (three images, one head) of 160 rows; objects mode (192 .. 199 tokens
 *                               per crop + its object token, oake_encode_objects) as tiles of (crop, head) with the
 *                               object token's masked attention in the same kernel (csrc/qkv_attn_obj.hip,
 *                               csrc/qkv_attn.hip).  2: as 1, the three-image form for every sequence of at most 53
 *                               tokens (measurement).  Other values: OAKE_ERR_INVALID.
 *                               [REF oadp/oake/globals.py:57, oadp/oake/objects.py:223-247]
 *   OAKE_OPT_PASS_CROPS         crops per internal encoder pass (vision handles).  get: the cap in force — what
 *                               oake_create derived from cfg.max_batch and cfg.pass_rows (default: ~25 600 token
 *                               rows).  set: a bound >= 1; the cap becomes min(value, the cap the handle
 *                               was created with: the workspace is sized for that) — what the reference's
 *                               `mini_batch_size` is: a memory bound on one pass [REF oadp/oake/objects.py:321-331].  A call's crops are cut into
 *                               passes under the cap: full passes and a shorter last one, or equal passes — whichever
 *                               fills whole rounds of tiles on the device (oake_debug_plan_pass, oake_hip_debug.h).
 *   OAKE_OPT_QKV_WALK           the fused qkv + attention kernel's tile walk inside an XCD (csrc/qkv_attn_obj.hip,
 *                               walk_decode): n > 0 = head blocks of n heads x group blocks of 32 / n groups (a group =
 *                               a crop, or four images), so that one round of an XCD's 32 blocks streams n x 295 KB of
 *                               the in-projection past 32 / n groups' rows instead of all 3.5 MB of it; 0 = group-major
 *                               (every head of 2.67 groups per round: the order of round 5).  Placement only: results
 *                               are identical.  Head counts that n does not divide fall back to 0.  Default 4.
 */
enum {
  OAKE_OPT_CLS_LAST = 1,
  OAKE_OPT_GEMM_VARIANT = 2,
  OAKE_OPT_GEMM_PANEL = 3,
  OAKE_OPT_ATTENTION_VARIANT = 4,
  OAKE_OPT_PATCH_DIRECT = 5,
  OAKE_OPT_CU_COUNT = 6,
  OAKE_OPT_FUSE_ATTN_OUT = 7,
  OAKE_OPT_PASS_CROPS = 8,
  OAKE_OPT_FUSE_QKV_ATTN = 9,
  OAKE_OPT_QKV_WALK = 10
};
OAKE_API int oake_set_option(oake_handle* h, int option, int value);
OAKE_API int oake_get_option(const oake_handle* h, int option, int* value);


#ifdef __cplusplus
}
#endif
#endif /* OAKE_HIP_H_ */
