/* clip_amd.h — MI355X-specific extensions of the clip.h C ABI.
 *
 * Everything here is ADDITIVE to the reference API (include/clip.h).  Plain C
 * ABI: pointers + sizes only, no torch / HIP types in signatures (a HIP stream
 * is passed as void*).  Device pointers are raw HBM addresses on the ctx's
 * device (e.g. torch.Tensor.data_ptr()).
 */
#ifndef CLIP_AMD_H
#define CLIP_AMD_H

#include "clip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of visible HIP devices (0 when there is no GPU / driver). */
int clip_amd_device_count(void);

/* Like clip_model_load (reference clip.cpp:334) but on an explicit device ordinal.
 * clip_model_load uses $CLIP_AMD_DEVICE, else $LOCAL_RANK, else device 0. */
struct clip_ctx * clip_amd_model_load(const char * fname, int verbosity, int device);

/* Multi-GPU form of clip_model_load (the reference has no multi-device path): ONE process, the weights
 * replicated on the first n_devices HIP devices (n_devices <= 0: all visible), one context + stream + host thread per
 * device behind the returned handle.  clip_image_batch_encode (reference clip.cpp:1247) on such a handle shards a batch
 * of B >= 2 * n_devices images into contiguous runs of ceil(B / n_devices), copies to each device only its shard,
 * runs the identical kernels, and collects the final embeddings with ONE ncclAllGather (RCCL over xGMI) of
 * [ceil(B / n_devices)][projection_dim] f32 rows per device into [n_devices * ceil(B / n_devices)][projection_dim] on
 * every device, then one device-to-host copy from device 0 into `vec`.  Everything else (text encode, small batches)
 * runs on device 0.  RCCL (librccl.so) is bound with dlopen by this call only.  NULL on failure.  clip_free() releases
 * every replica. */
struct clip_ctx * clip_amd_model_load_multi(const char * fname, int verbosity, int n_devices);
/* Number of devices behind a handle (1 for clip_model_load / clip_amd_model_load handles). */
int clip_amd_ctx_device_count(const struct clip_ctx * ctx);

/* Repacked-weight cache (replaces the per-load tensor read + repack of reference clip.cpp:435-461).  Opt-in through the
 * environment: CLIP_AMD_WEIGHT_CACHE=<directory>.  clip_model_load then keeps "<directory>/<file name>.<content key>.hbm" — the exact
 * HBM image of the model — and later loads of the same file content read that image instead of repacking the GGUF tensors.
 * Returns 1 when this context's weights came from the cache, 0 when they were repacked from the GGUF. */
int clip_amd_weights_from_cache(const struct clip_ctx * ctx);
/* The shard clip_image_batch_encode gives device `device_index` of `n_devices` for a batch of `total` images: rows
 * [*lo, *hi), and the padded per-device row count of the all-gather (pure arithmetic, no device needed). */
void clip_amd_shard_bounds(int total, int n_devices, int device_index, int * lo, int * hi, int * rows_per_device);
/* Device address (on device `device_index`) of the gathered [n_devices * rows_per_device][projection_dim] embeddings of
 * the last sharded clip_image_batch_encode: the canonical device-resident result (valid until the next call). */
const float * clip_amd_gathered_embeddings(const struct clip_ctx * ctx, int device_index);

/* Device-resident sharded encodes on a clip_amd_model_load_multi handle (the form bench.py --single-process measures): shard g of the
 * batch — items [lo, hi) of clip_amd_shard_bounds(total, n_devices, g) — already sits ON device g: d_imgs[g] = [hi - lo][S][S][3] f32
 * preprocessed images; d_ids[g] = the token ids of the shard's texts back to back, h_offsets = total + 1 prefix offsets of ALL texts
 * (host).  Each replica runs its tower on its own stream and host thread, then ONE grouped ncclAllGather leaves [n_devices *
 * rows_per_device][projection_dim] on every device (clip_amd_gathered_embeddings); vec (host, [total][projection_dim]) may be NULL.
 * Synchronous: returns when every replica stream has finished.  clip_amd_image_batch_encode_u8 and clip_text_batch_encode shard the
 * same way on such a handle when total >= 2 * n_devices. */
bool clip_amd_image_batch_encode_device_multi(struct clip_ctx * ctx, const float * const * d_imgs, int total, bool normalize, float * vec);
bool clip_amd_text_batch_encode_device_multi(struct clip_ctx * ctx, const int32_t * const * d_ids, const int32_t * h_offsets, int total, bool normalize,
                                             float * vec);
/* Both towers of a step in ONE call: on every device the vision tower (reference clip.cpp:1247) runs on the replica's stream and the text
 * tower (clip.cpp:1016) on the stream of a sibling context of that device (created on the first such call: own activation workspace,
 * the replica's weight image), forked and joined with events, then ONE grouped ncclAllGather of [rows_per_device(n_images) + rows_per_device(n_texts)]
 * rows per device; clip_amd_gathered_embeddings then holds n_devices such blocks (image rows first).  Shards as above; vec_img
 * [n_images][projection_dim] and vec_txt [n_texts][projection_dim] (host) may each be NULL.  Synchronous. */
bool clip_amd_encode_pair_device_multi(struct clip_ctx * ctx, const float * const * d_imgs, int n_images, const int32_t * const * d_ids,
                                       const int32_t * h_offsets, int n_texts, bool normalize, float * vec_img, float * vec_txt);

/* Which device a ctx lives on (-1: host-only ctx, see CLIP_AMD_ALLOW_NO_DEVICE in DESIGN.md). */
int clip_amd_ctx_device(const struct clip_ctx * ctx);

/* Bind all subsequent launches of this ctx to an existing HIP stream (e.g. torch's current
 * stream, as an integer handle).  NULL restores the ctx's own stream.  A context is single-stream and
 * single-thread at any one time (it owns one activation workspace): the switch orders the new stream behind
 * everything still queued on the previous one — by recording an event on the PREVIOUS stream, so that stream must still
 * exist when clip_amd_set_stream (and clip_free, which synchronises the stream in use) is called: switch the context back
 * (or free it) BEFORE destroying a stream it was bound to. */
void clip_amd_set_stream(struct clip_ctx * ctx, void * hip_stream);

/* Tell the context that its device carries other work at the same time (shared != 0) — typically the other tower of a two-tower step
 * on a second context / stream.  The GEMM heuristic then keeps to kernels that share a compute unit (two workgroups per CU); with
 * shared == 0 (the default: one encoder call at a time, as every reference caller does) the wide q/k/v and FFN-up GEMMs of a large batch
 * run on kernels that take a whole CU each (k_gemm32.hip).  Results agree within fp16 output rounding either way.  The two-tower
 * multi-GPU entry point (clip_amd_encode_pair_device_multi) knows by itself. */
void clip_amd_set_device_shared(struct clip_ctx * ctx, int shared);

/* Device-resident form of clip_image_batch_encode (reference clip.cpp:1247-1523):
 * d_imgs  : B x [S,S,3] float32 interleaved RGB, already preprocessed, in HBM
 * d_out   : B x projection_dim float32 in HBM
 * Asynchronous on the ctx stream; no host<->device copies. */
bool clip_amd_image_batch_encode_device(struct clip_ctx * ctx, const float * d_imgs, int batch, float * d_out,
                                        bool normalize);

/* GPU image preprocessing ("next" row §8f-1; reference clip_image_preprocess / clip_image_batch_preprocess,
 * clip.cpp:728-1008): n raw RGB u8 images of any size (HOST pointers, clip_image_u8 as filled by
 * clip_image_load_from_file) -> d_out [n][S][S][3] float32 in HBM, resized (antialiased bicubic, shorter side -> S),
 * centre-cropped and normalised exactly like clip_image_preprocess (bit-identical).  Asynchronous on the ctx stream. */
bool clip_amd_image_batch_preprocess_device(struct clip_ctx * ctx, const struct clip_image_u8 * imgs, int n, float * d_out);

/* clip_image_batch_preprocess + clip_image_batch_encode in one call with the preprocessing on the GPU.
 * vec: [n][projection_dim] on the host.  Same results as the two-step host path. */
bool clip_amd_image_batch_encode_u8(struct clip_ctx * ctx, const struct clip_image_u8 * imgs, int n, float * vec, bool normalize);

/* Batched text encoding ("next" row §8f-2; per-text semantics identical to
 * clip_text_encode, reference clip.cpp:1016-1233).  Texts are ragged:
 * tokens[i].data holds tokens[i].size ids incl. BOS/EOS.  vec: [n_texts][projection_dim]. */
bool clip_text_batch_encode(const struct clip_ctx * ctx, const int n_threads, const struct clip_tokens * tokens,
                            size_t n_texts, float * vec, const bool normalize);

/* Device-resident batched text encode: d_ids = concatenated ids (int32, HBM),
 * h_offsets = n_texts+1 prefix offsets (HOST memory), d_out [n_texts][proj] in HBM. */
bool clip_amd_text_batch_encode_device(struct clip_ctx * ctx, const int32_t * d_ids, const int32_t * h_offsets,
                                       int n_texts, float * d_out, bool normalize);

/* GPU-side zero-shot scoring ("next" row §8f-2; reference clip_similarity_score + softmax_with_sorting as composed by
 * clip_zero_shot_label_image, clip.cpp:1624-1659, and tests/benchmark.cpp:114-160).  For each of n_images embeddings
 * d_img [n_images][dim]: similarities with d_txt [n_labels][dim] (sequential fp32 dot, same order as the host),
 * exp(x)+1e-9 normalised by the double-precision sum, sorted descending -> d_scores / d_indices [n_images][n_labels].
 * All pointers are HBM addresses; asynchronous on the ctx stream.  n_labels <= 8192, dim <= 4096. */
bool clip_amd_zero_shot_score_device(struct clip_ctx * ctx, const float * d_img, int n_images, const float * d_txt, int n_labels, int dim,
                                     float * d_scores, int * d_indices);

/* Batched clip_zero_shot_label_image: raw u8 images (host) x labels -> scores / indices [n_images][n_labels] (host), per
 * image exactly what clip_zero_shot_label_image returns.  Labels are encoded once (one ragged text batch), images are
 * preprocessed, encoded and scored on the GPU. */
bool clip_amd_zero_shot_label_images(struct clip_ctx * ctx, const struct clip_image_u8 * imgs, int n_images, const char ** labels,
                                     size_t n_labels, float * scores, int * indices);

/* Block until everything queued on the ctx stream has finished. */
void clip_amd_synchronize(struct clip_ctx * ctx);

/* Per-kernel-family accumulated device time (ms) since the last reset, measured with HIP events
 * on the ctx stream when profiling is enabled.  families: 0 gemm, 1 attention, 2 layernorm,
 * 3 other.  launches[] receives launch counts.  Returns number of families written. */
void clip_amd_profile_enable(struct clip_ctx * ctx, bool on);
int clip_amd_profile_read(struct clip_ctx * ctx, float * ms, int64_t * launches, int cap, bool reset);
/* Text report, one line per (kernel family : MxNxK) tag: "tag launches total_ms flops bytes".
 * Returns the number of bytes needed (call with buf == NULL to size the buffer). */
int clip_amd_profile_report(struct clip_ctx * ctx, char * buf, int cap, bool reset);

/* ---- kernel-level test hooks (used by tests/ only; host pointers, synchronous) ----
 * Y[M,N] = X[M,K] . W[N,K]^T (+bias) through the production dequant-GEMM kernel.
 * w_raw is the tensor in its GGUF/ggml block layout (type: ggml type id 0,1,2,3,6,7,8).
 * epilogue: 0 plain f32 out, 1 f16 out (returned widened to f32), 2 gelu->f16, 3 quick-gelu->f16,
 *           4 residual: Y = resid + X.W^T + bias (f32).  tile: 0 auto, else BM*1000+BN. */
int clip_amd_test_gemm(int type, const void * w_raw, int64_t N, int64_t K, const float * x, int64_t M,
                       const float * bias, const float * resid, float * y, int epilogue, int tile);
/* Same, plus the two epilogue features the plain hook cannot reach: epilogue 1 with qcols / qscale (columns n < qcols
 * are multiplied by qscale AFTER the bias: the 1/sqrt(d_head) Q scale of the fused q/k/v projection, reference
 * clip.cpp:1363) and epilogue 5 = patch embedding (reference clip.cpp:1309-1331): GEMM row m -> output row
 * (m / Np) * T + 1 + m % Np with pos[1 + m % Np] added, no bias; y is [(M / Np) * T][N] and is uploaded first so that
 * untouched rows (the class-token rows) keep the caller's fill; pos is [T][N]. */
int clip_amd_test_gemm_ex(int type, const void * w_raw, int64_t N, int64_t K, const float * x, int64_t M,
                          const float * bias, const float * resid, float * y, int epilogue, int tile,
                          int qcols, float qscale, int Np, int T, const float * pos);
/* LayerNorm fold A/B (the LayerNorm launches of reference clip.cpp:1350-1355,1400-1405 folded into the GEMM epilogues around them):
 *   x1 = resid + a . W1^T + b1                       [M][h]   W1: [h][K1]  (residual epilogue)
 *   y  = epi2( LN(x1; gamma, beta, eps) . W2^T + b2 )  [M][N2]  W2: [N2][h]  (epi2: 1 f16 (+ qcols / qscale), 2 gelu, 3 quick-gelu)
 * fold = 0: three launches (GEMM, LayerNorm kernel, GEMM); fold = 1: two launches — the residual epilogue also writes fp16(x1 gamma) and
 * partial row statistics, the second GEMM's epilogue applies rstd (acc - mean c) + b'.  tile1 / tile2 as `tile` of clip_amd_test_gemm.
 * x1_out [M][h] f32, y_out [M][N2] (fp16 widened).  h % 64 == 0.  tile1 < 0: both GEMMs on the small-M kernels (k_skinny.hip, M <= 128; fold = 0
 * is then the LayerNorm fused on the operand, fold = 1 the folded form).  fold = 2: the folded form with the operand CENTRED (the default of
 * the layer chain since round 4): fp16((x1 - mu) gamma) with mu_m = the mean of row m of `resid` (what the previous LayerNorm's consumer
 * leaves), and rstd (acc - (mean - mu) c) + b' in the consumer. */
int clip_amd_test_lnfold(int type, const void * w1_raw, int64_t h, int64_t K1, const void * w2_raw, int64_t N2, const float * a, int64_t M,
                         const float * b1, const float * resid, const float * gamma, const float * beta, float eps, const float * b2,
                         int epi2, int tile1, int tile2, int fold, int qcols, float qscale, float * x1_out, float * y_out);
/* The tile the heuristic of launch_gemm picks for an [M][K] x [N][K]^T problem (pure host arithmetic, no device needed):
 * BM * 1000 + BN; BM = 65: the mid-M ring kernel on 64-row tiles (k_gemm_ring.hip), BN = 256 / 258-260: the large-M panel kernels. */
int clip_amd_test_gemm_tile(int64_t M, int64_t N, int64_t K, int quantised);
int clip_amd_test_gemm_tile_ex(int64_t M, int64_t N, int64_t K, int quantised, int shared_device);   /* ... with clip_amd_set_device_shared's flag */
/* Average device time (microseconds, HIP events) of one GEMM shape through the production kernel on random
 * weights of ggml type `type`; < 0 on error.  Used by scripts/gemm_bench.py for kernel A/B work. */
float clip_amd_bench_gemm(int type, int64_t N, int64_t K, int64_t M, int epilogue, int tile, int iters);
/* Small-M kernel (one image / one text: the layers use it up to 64 rows, the hook up to 512; k_skinny.hip): y[M][N] = epilogue(A . W^T + bias) with A = fp16(x), or — ln_w
 * != NULL — A = LayerNorm(x) fused into the kernel.  epilogue 0 f32, 1 f16 (+ qcols / qscale), 2 gelu, 3 quick-gelu, 4 residual
 * (stats_out, if not NULL, receives the [128 rows][128 slots][2] partial row statistics the epilogue leaves for the next LayerNorm:
 * slot j of a row = its (sum, sum of squares) over output columns 16 j .. 16 j + 15).
 * Returns -5 when the combination is not covered by this path. */
int clip_amd_test_skinny(int type, const void * w_raw, int64_t N, int64_t K, const float * x, int64_t M, const float * bias,
                         const float * resid, const float * ln_w, const float * ln_b, float eps, float * y, int epilogue,
                         int qcols, float qscale, float * stats_out);
/* y = LayerNorm(x)*w + b, rows x h. out_f16 != 0 rounds the result through fp16. */
int clip_amd_test_layernorm(const float * x, const float * w, const float * b, float eps, int64_t rows, int64_t h,
                            float * y, int out_f16);
/* Multi-head attention over nseq sequences of length T each: qkv [nseq*T][3h] (q pre-scaled), out [nseq*T][h]. */
int clip_amd_test_attention(const float * qkv, int nseq, int T, int h, int n_head, int causal, float * out);

#ifdef __cplusplus
}
#endif

#endif /* CLIP_AMD_H */
