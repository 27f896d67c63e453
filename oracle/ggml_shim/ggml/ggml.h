// oracle/ggml_shim/ggml/ggml.h — TEST INFRASTRUCTURE (checker only; never on the product path).
//
// The subset of the ggml / gguf C API that the reference's clip.cpp calls (census: SURVEY.md Appendix B), declared from the published
// API of ggml @ dd1d575 (the un-vendored submodule of /root/reference) so that the reference's OWN clip.cpp compiles where it lies and runs
// its OWN loader, tokenizer, preprocessing, scoring and — op by op — its OWN vision / text graphs.  The implementation behind these
// declarations (ggml_shim.cpp) is NOT ggml: tensors are plain host arrays, every op is evaluated eagerly when it is called (clip.cpp fills its
// inputs before it builds the ops that read them, so program order is a valid schedule), and the arithmetic of each op is the oracle's
// restatement of the ggml arithmetic (clip_oracle.cpp, same functions).  What this pins: the oracle's WIRING of the two graphs against the
// reference's source.  What it cannot pin: the op arithmetic itself (still "parity unpinned" against ggml; oracle/GGML_ASSUMPTIONS.md).
#pragma once

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MAX_DIMS 4
#define GGML_MAX_NAME 64
#define GGML_MEM_ALIGN 16
#define GGML_QNT_VERSION 2
#define GGML_OBJECT_SIZE 32
#define GGML_PAD(x, n) (((x) + (n) - 1) & ~((n) - 1))
#define GGML_ASSERT(x)                                                                \
    do {                                                                              \
        if (!(x)) {                                                                   \
            fprintf(stderr, "GGML_ASSERT: %s:%d: %s\n", __FILE__, __LINE__, #x);     \
            abort();                                                                  \
        }                                                                             \
    } while (0)

typedef uint16_t ggml_fp16_t;

enum ggml_type {
    GGML_TYPE_F32 = 0,
    GGML_TYPE_F16 = 1,
    GGML_TYPE_Q4_0 = 2,
    GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q5_0 = 6,
    GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,
    GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_I8 = 16,
    GGML_TYPE_I16 = 17,
    GGML_TYPE_I32 = 18,
    GGML_TYPE_COUNT,
};

struct ggml_context;

struct ggml_tensor {
    enum ggml_type type;
    int n_dims;
    int64_t ne[GGML_MAX_DIMS];   // elements per dimension, ne[0] contiguous
    size_t nb[GGML_MAX_DIMS];    // strides in bytes
    void * data;
    char name[GGML_MAX_NAME];
    // shim-private
    struct ggml_context * ctx_;
    bool owns_data_;
};

struct ggml_init_params {
    size_t mem_size;
    void * mem_buffer;
    bool no_alloc;
};

struct ggml_scratch {
    size_t offs;
    size_t size;
    void * data;
};

struct ggml_cgraph {
    int n_nodes;
    struct ggml_tensor * last;
};

struct ggml_cplan {
    size_t work_size;
    uint8_t * work_data;
    int n_threads;
};

// timing (used by the reference's example programs)
void ggml_time_init(void);
int64_t ggml_time_ms(void);
int64_t ggml_time_us(void);

// contexts
struct ggml_context * ggml_init(struct ggml_init_params params);
void ggml_free(struct ggml_context * ctx);
size_t ggml_used_mem(const struct ggml_context * ctx);
size_t ggml_get_mem_size(const struct ggml_context * ctx);
size_t ggml_set_scratch(struct ggml_context * ctx, struct ggml_scratch scratch);

// tensors
int64_t ggml_nelements(const struct ggml_tensor * t);
size_t ggml_nbytes(const struct ggml_tensor * t);
size_t ggml_nbytes_pad(const struct ggml_tensor * t);
size_t ggml_element_size(const struct ggml_tensor * t);
float ggml_fp16_to_fp32(ggml_fp16_t x);
ggml_fp16_t ggml_fp32_to_fp16(float x);

struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0);
struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1);
struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2);
struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
struct ggml_tensor * ggml_new_i32(struct ggml_context * ctx, int32_t value);
struct ggml_tensor * ggml_new_f32(struct ggml_context * ctx, float value);
struct ggml_tensor * ggml_dup_tensor(struct ggml_context * ctx, const struct ggml_tensor * src);
struct ggml_tensor * ggml_get_tensor(struct ggml_context * ctx, const char * name);
struct ggml_tensor * ggml_set_name(struct ggml_tensor * t, const char * name);
struct ggml_tensor * ggml_set_zero(struct ggml_tensor * t);
void ggml_set_i32_1d(const struct ggml_tensor * t, int i, int32_t value);
void * ggml_get_data(const struct ggml_tensor * t);
float * ggml_get_data_f32(const struct ggml_tensor * t);

// ops
struct ggml_tensor * ggml_add(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_acc(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, size_t nb1, size_t nb2, size_t nb3, size_t offset);
struct ggml_tensor * ggml_mul(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_div(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_sqr(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_sqrt(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_sum(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_repeat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps);
struct ggml_tensor * ggml_mul_mat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_scale_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_gelu_inplace(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_gelu_quick_inplace(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_soft_max_inplace(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_diag_mask_inf_inplace(struct ggml_context * ctx, struct ggml_tensor * a, int n_past);
struct ggml_tensor * ggml_cpy(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_cont(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1);
struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2);
struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
struct ggml_tensor * ggml_permute(struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3);
struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_conv_2d(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, int s0, int s1, int p0, int p1, int d0, int d1);

// graph (everything has been evaluated by the time these are called)
void ggml_build_forward_expand(struct ggml_cgraph * cgraph, struct ggml_tensor * tensor);
struct ggml_cplan ggml_graph_plan(struct ggml_cgraph * cgraph, int n_threads);
int ggml_graph_compute(struct ggml_cgraph * cgraph, struct ggml_cplan * cplan);

// quantisation entry points of the quantize tool (quantize_row_*_reference over rows of k; the histogram argument is ignored)
size_t ggml_quantize_q4_0(const float * src, void * dst, int n, int k, int64_t * hist);
size_t ggml_quantize_q4_1(const float * src, void * dst, int n, int k, int64_t * hist);
size_t ggml_quantize_q5_0(const float * src, void * dst, int n, int k, int64_t * hist);
size_t ggml_quantize_q5_1(const float * src, void * dst, int n, int k, int64_t * hist);
size_t ggml_quantize_q8_0(const float * src, void * dst, int n, int k, int64_t * hist);

// gguf
struct gguf_context;

struct gguf_init_params {
    bool no_alloc;
    struct ggml_context ** ctx;
};

struct gguf_context * gguf_init_empty(void);
struct gguf_context * gguf_init_from_file(const char * fname, struct gguf_init_params params);
void gguf_free(struct gguf_context * ctx);
int gguf_get_version(const struct gguf_context * ctx);
size_t gguf_get_alignment(const struct gguf_context * ctx);
size_t gguf_get_data_offset(const struct gguf_context * ctx);
int gguf_get_n_kv(const struct gguf_context * ctx);
int gguf_find_key(const struct gguf_context * ctx, const char * key);
const char * gguf_get_key(const struct gguf_context * ctx, int i);
uint32_t gguf_get_val_u32(const struct gguf_context * ctx, int i);
float gguf_get_val_f32(const struct gguf_context * ctx, int i);
bool gguf_get_val_bool(const struct gguf_context * ctx, int i);
const char * gguf_get_val_str(const struct gguf_context * ctx, int i);
int gguf_get_arr_n(const struct gguf_context * ctx, int i);
const void * gguf_get_arr_data(const struct gguf_context * ctx, int i);
const char * gguf_get_arr_str(const struct gguf_context * ctx, int key_id, int i);
int gguf_get_n_tensors(const struct gguf_context * ctx);
size_t gguf_get_tensor_offset(const struct gguf_context * ctx, int i);
char * gguf_get_tensor_name(const struct gguf_context * ctx, int i);
// writer side of the quantize tool
void gguf_set_kv(struct gguf_context * ctx, struct gguf_context * src);
void gguf_set_val_u32(struct gguf_context * ctx, const char * key, uint32_t val);
void gguf_add_tensor(struct gguf_context * ctx, const struct ggml_tensor * tensor);
void gguf_set_tensor_type(struct gguf_context * ctx, const char * name, enum ggml_type type);
void gguf_set_tensor_data(struct gguf_context * ctx, const char * name, const void * data, size_t size);
size_t gguf_get_meta_size(const struct gguf_context * ctx);
void gguf_get_meta_data(const struct gguf_context * ctx, void * data);

#ifdef __cplusplus
}
#endif
