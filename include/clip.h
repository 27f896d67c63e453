/* clip.h — C API of the MI355X-native CLIP encoder (libclip.so).
 *
 * Drop-in boundary: every struct layout and every exported function below has
 * the same name, argument order/meaning and return convention as the public
 * header of monatis/clip.cpp (reference clip.h:14-109), so the reference's
 * callers (examples/main.cpp, zsl.cpp, extract.cpp, simple.c,
 * tests/benchmark.cpp and the ctypes binding clip_cpp/clip.py) compile and
 * link against this library unchanged.  What differs is everything behind it:
 * the forward pass runs as hand-written HIP kernels on gfx950 and there is no
 * ggml.  `n_threads` arguments are accepted for compatibility; the encoders
 * ignore them (host-side batch preprocessing still uses them).
 *
 * Error convention: bool true = success; loaders return NULL on
 * any failure (the reference throws / exit(1)s on malformed files — this
 * library never throws across the C ABI).  Every encoder fails LOUDLY
 * (stderr + false) when no HIP device / kernel image is available: there is no
 * CPU fallback.
 */
#ifndef CLIP_H
#define CLIP_H

#include "ggml/ggml.h" /* timing shim only: ggml_time_init / ggml_time_us (used by the reference's examples) */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct clip_ctx; /* opaque (reference clip.h:8) */

/* reference clip.h:14-23 — 8 x 4 bytes */
struct clip_text_hparams {
    int32_t n_vocab;
    int32_t num_positions;
    int32_t hidden_size;
    int32_t n_intermediate;
    int32_t projection_dim;
    int32_t n_head;
    int32_t n_layer;
    float eps;
};

/* reference clip.h:25-34 */
struct clip_vision_hparams {
    int32_t image_size;
    int32_t patch_size;
    int32_t hidden_size;
    int32_t n_intermediate;
    int32_t projection_dim;
    int32_t n_head;
    int32_t n_layer;
    float eps;
};

/* reference clip.h:36-40 — `data` is allocated by clip_tokenize with new[] and owned by the caller */
typedef int32_t clip_vocab_id;
struct clip_tokens {
    clip_vocab_id * data;
    size_t size;
};

/* reference clip.h:50-55 — interleaved RGB, 8 bit */
struct clip_image_u8 {
    int nx;
    int ny;
    uint8_t * data;
    size_t size;
};

/* reference clip.h:57-64 — interleaved RGBRGB... (HWC) float32, already normalised */
struct clip_image_f32 {
    int nx;
    int ny;
    float * data;
    size_t size;
};

/* reference clip.h:66-74 */
struct clip_image_u8_batch {
    struct clip_image_u8 * data;
    size_t size;
};
struct clip_image_f32_batch {
    struct clip_image_f32 * data;
    size_t size;
};

/* ---- model lifetime (reference clip.h:42-47, clip.cpp:334-596, :1010-1014, :1846-1847) ---- */
struct clip_ctx * clip_model_load(const char * fname, const int verbosity);
void clip_free(struct clip_ctx * ctx);
struct clip_text_hparams * clip_get_text_hparams(struct clip_ctx * ctx);
struct clip_vision_hparams * clip_get_vision_hparams(struct clip_ctx * ctx);

/* ---- tokenizer (reference clip.h:76, clip.cpp:598-679): bit-exact token ids ---- */
bool clip_tokenize(const struct clip_ctx * ctx, const char * text, struct clip_tokens * tokens);

/* ---- image containers + host preprocessing (reference clip.h:78-88,93-94; clip.cpp:681-1008) ---- */
struct clip_image_u8 * clip_image_u8_make();
struct clip_image_f32 * clip_image_f32_make();
void clip_image_u8_clean(struct clip_image_u8 * img);
void clip_image_f32_clean(struct clip_image_f32 * res);
void clip_image_u8_free(struct clip_image_u8 * img);
void clip_image_f32_free(struct clip_image_f32 * res);
bool clip_image_load_from_file(const char * fname, struct clip_image_u8 * img);
bool clip_image_preprocess(const struct clip_ctx * ctx, const struct clip_image_u8 * img, struct clip_image_f32 * res);
void clip_image_batch_preprocess(const struct clip_ctx * ctx, const int n_threads,
                                 const struct clip_image_u8_batch * img_inputs, struct clip_image_f32_batch * imgs_resized);

/* ---- THE HOT PATH (reference clip.h:90-97; clip.cpp:1016-1233, :1235-1245, :1247-1523) ----
 * vec must hold projection_dim floats (x batch size for the batch call, row-major [B][proj]). */
bool clip_text_encode(const struct clip_ctx * ctx, const int n_threads, const struct clip_tokens * tokens, float * vec,
                      const bool normalize);
bool clip_image_encode(const struct clip_ctx * ctx, const int n_threads, struct clip_image_f32 * img, float * vec,
                       const bool normalize);
bool clip_image_batch_encode(const struct clip_ctx * ctx, const int n_threads, const struct clip_image_f32_batch * imgs,
                             float * vec, const bool normalize);

/* ---- scoring helpers (reference clip.h:101-106; clip.cpp:1525-1659) ---- */
bool clip_compare_text_and_image(const struct clip_ctx * ctx, const int n_threads, const char * text,
                                 const struct clip_image_u8 * image, float * score);
float clip_similarity_score(const float * vec1, const float * vec2, const int vec_dim);
bool softmax_with_sorting(float * arr, const int length, float * sorted_scores, int * indices);
bool clip_zero_shot_label_image(struct clip_ctx * ctx, const int n_threads, const struct clip_image_u8 * input_img,
                                const char ** labels, const size_t n_labels, float * scores, int * indices);

/* ---- quantizer (reference clip.h:108; clip.cpp:1661-1844). itype: 2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0 ---- */
bool clip_model_quantize(const char * fname_inp, const char * fname_out, const int itype);

#ifdef __cplusplus
}
#endif

#endif /* CLIP_H */
