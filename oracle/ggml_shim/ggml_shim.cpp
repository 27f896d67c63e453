// oracle/ggml_shim/ggml_shim.cpp — TEST INFRASTRUCTURE (checker only; never on the product path).
//
// An eager stand-in for the slice of ggml / gguf that /root/reference/clip.cpp calls, so that the reference's own source runs here:
// its loader on our GGUF files, its tokenizer / preprocessing / scoring, and its two graph builders op by op.  See ggml/ggml.h for what
// this does and does not pin.  Tensors are host arrays owned by their context; every op computes its result when it is called; the
// arithmetic of an op is the oracle's restatement of ggml's (the orc_* functions of clip_oracle.cpp, compiled into the same library):
//   mul_mat   quantised / f16 weights: activations converted to the weight type's vec_dot type, f32 x f32: double accumulation
//   norm      double sums, biased variance, (x - mean) * 1 / sqrtf(var + eps)
//   soft_max  fp16 exp table, -inf -> 0;  gelu / gelu_quick: fp16 tables
//   conv_2d   fp16 im2col x fp16 kernel, k = (c, ky, kx)
//   get_rows  dequantize_row
// Data movement ops (repeat, permute, cont, reshape, cpy, acc) and the f32 elementwise ops are exact.
#include "ggml/ggml.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

// the oracle's kernels (clip_oracle.cpp; mode 0 = ggml-faithful numerics)
extern "C" {
size_t orc_row_bytes(int type, int64_t k);
void orc_dequantize(int type, const void * src, float * dst, int64_t nrows, int64_t k);
int orc_mul_mat(int type, const void * w, int64_t N, int64_t K, const float * X, int64_t M, float * Y, int mode, int n_threads);
void orc_activation(float * x, int64_t n, int use_gelu, int mode);
void orc_softmax_rows(float * s, int64_t rows, int64_t n, int mode);
uint16_t orc_f2h(float x);
float orc_h2f(uint16_t x);
}

struct ggml_context {
    std::vector<ggml_tensor *> tensors;
    bool no_alloc = false;
    size_t bytes = 0;
};

namespace {

[[noreturn]] void die(const char * what) {
    fprintf(stderr, "ggml shim: %s\n", what);
    abort();
}

size_t type_size(ggml_type t) {          // bytes per block
    switch (t) {
    case GGML_TYPE_F32: case GGML_TYPE_I32: return 4;
    case GGML_TYPE_F16: case GGML_TYPE_I16: return 2;
    case GGML_TYPE_I8: return 1;
    case GGML_TYPE_Q4_0: return 18;
    case GGML_TYPE_Q4_1: return 20;
    case GGML_TYPE_Q5_0: return 22;
    case GGML_TYPE_Q5_1: return 24;
    case GGML_TYPE_Q8_0: return 34;
    default: die("unsupported tensor type");
    }
}
int blck(ggml_type t) {
    switch (t) {
    case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_0: case GGML_TYPE_Q5_1: case GGML_TYPE_Q8_0: return 32;
    default: return 1;
    }
}

void set_strides(ggml_tensor * t) {
    t->nb[0] = type_size(t->type);
    t->nb[1] = t->nb[0] * (size_t)(t->ne[0] / blck(t->type));
    for (int i = 2; i < GGML_MAX_DIMS; i++) t->nb[i] = t->nb[i - 1] * (size_t)t->ne[i - 1];
}

bool contiguous(const ggml_tensor * t) {
    if (t->nb[0] != type_size(t->type)) return false;
    if (t->nb[1] != t->nb[0] * (size_t)(t->ne[0] / blck(t->type))) return false;
    for (int i = 2; i < GGML_MAX_DIMS; i++)
        if (t->nb[i] != t->nb[i - 1] * (size_t)t->ne[i - 1]) return false;
    return true;
}

ggml_tensor * new_tensor(ggml_context * ctx, ggml_type type, int n_dims, const int64_t * ne, bool alloc) {
    ggml_tensor * t = (ggml_tensor *)calloc(1, sizeof(ggml_tensor));
    t->type = type;
    t->n_dims = n_dims;
    for (int i = 0; i < GGML_MAX_DIMS; i++) t->ne[i] = i < n_dims ? ne[i] : 1;
    set_strides(t);
    t->ctx_ = ctx;
    if (alloc && !ctx->no_alloc) {
        const size_t nbytes = ggml_nbytes(t);
        t->data = calloc(1, nbytes ? nbytes : 1);      // (ggml leaves new tensors uninitialised; zero is the deterministic choice)
        if (!t->data) die("out of memory");
        t->owns_data_ = true;
        ctx->bytes += nbytes;
    }
    ctx->tensors.push_back(t);
    return t;
}

ggml_tensor * like(ggml_context * ctx, const ggml_tensor * a, ggml_type type = GGML_TYPE_F32) { return new_tensor(ctx, type, a->n_dims, a->ne, true); }

ggml_tensor * view_of(ggml_context * ctx, ggml_tensor * a) {      // same data, same shape / strides
    ggml_tensor * t = new_tensor(ctx, a->type, a->n_dims, a->ne, false);
    memcpy(t->nb, a->nb, sizeof t->nb);
    t->data = a->data;
    return t;
}

inline float * f32_at(const ggml_tensor * t, int64_t i0, int64_t i1, int64_t i2, int64_t i3) {
    return (float *)((char *)t->data + i0 * t->nb[0] + i1 * t->nb[1] + i2 * t->nb[2] + i3 * t->nb[3]);
}

void need_f32(const ggml_tensor * t, const char * op) {
    if (t->type != GGML_TYPE_F32) { fprintf(stderr, "ggml shim: %s expects f32\n", op); abort(); }
}
bool same_shape(const ggml_tensor * a, const ggml_tensor * b) {
    for (int i = 0; i < GGML_MAX_DIMS; i++) if (a->ne[i] != b->ne[i]) return false;
    return true;
}

// c = f(a, b) elementwise on equal shapes (any strides), c contiguous
template <typename F> ggml_tensor * binary(ggml_context * ctx, ggml_tensor * a, ggml_tensor * b, const char * op, F f) {
    need_f32(a, op); need_f32(b, op);
    if (!same_shape(a, b)) { fprintf(stderr, "ggml shim: %s: shapes differ\n", op); abort(); }
    ggml_tensor * c = like(ctx, a);
    for (int64_t i3 = 0; i3 < a->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < a->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < a->ne[1]; i1++)
                for (int64_t i0 = 0; i0 < a->ne[0]; i0++) *f32_at(c, i0, i1, i2, i3) = f(*f32_at(a, i0, i1, i2, i3), *f32_at(b, i0, i1, i2, i3));
    return c;
}

// ---- gguf ----
struct Kv {
    std::string key;
    uint32_t type = 0, elem_type = 0;
    uint64_t n = 0;                       // array length
    uint64_t scalar = 0;                  // raw bits of a scalar
    std::string str;
    std::vector<std::string> strs;
    std::vector<uint8_t> raw;             // array of scalars
};
struct TInfo {
    std::string name;
    uint64_t offset = 0;
};

size_t scalar_size(uint32_t t) {
    switch (t) {
    case 0: case 1: case 7: return 1;
    case 2: case 3: return 2;
    case 4: case 5: case 6: return 4;
    case 10: case 11: case 12: return 8;
    default: return 0;
    }
}

}  // namespace

struct gguf_context {
    uint32_t version = 0;
    size_t alignment = 32, data_offset = 0;
    std::vector<Kv> kv;
    std::vector<TInfo> infos;
    // writer side (gguf_init_empty): name / type / shape / byte size per tensor, offsets recomputed from the sizes
    struct Out { std::string name; uint32_t type; uint32_t n_dims; int64_t ne[4]; size_t size; };
    std::vector<Out> out;
};

extern "C" {

// ---- timing ----
void ggml_time_init(void) {}
int64_t ggml_time_us(void) { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int64_t ggml_time_ms(void) { return ggml_time_us() / 1000; }

// ---- contexts ----
struct ggml_context * ggml_init(struct ggml_init_params params) {
    ggml_context * c = new ggml_context();
    c->no_alloc = params.no_alloc;
    return c;
}
void ggml_free(struct ggml_context * ctx) {
    if (!ctx) return;
    for (ggml_tensor * t : ctx->tensors) {
        if (t->owns_data_) free(t->data);
        free(t);
    }
    delete ctx;
}
size_t ggml_used_mem(const struct ggml_context * ctx) { return ctx->bytes; }
size_t ggml_get_mem_size(const struct ggml_context * ctx) { return ctx->bytes + ctx->tensors.size() * (sizeof(ggml_tensor) + GGML_OBJECT_SIZE); }
size_t ggml_set_scratch(struct ggml_context *, struct ggml_scratch) { return 0; }

// ---- tensors ----
int64_t ggml_nelements(const struct ggml_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
size_t ggml_nbytes(const struct ggml_tensor * t) { return type_size(t->type) * (size_t)(t->ne[0] / blck(t->type)) * (size_t)(t->ne[1] * t->ne[2] * t->ne[3]); }
size_t ggml_nbytes_pad(const struct ggml_tensor * t) { return GGML_PAD(ggml_nbytes(t), (size_t)GGML_MEM_ALIGN); }
size_t ggml_element_size(const struct ggml_tensor * t) { return type_size(t->type); }
float ggml_fp16_to_fp32(ggml_fp16_t x) { return orc_h2f(x); }
ggml_fp16_t ggml_fp32_to_fp16(float x) { return orc_f2h(x); }

struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0) { const int64_t ne[1] = {ne0}; return new_tensor(ctx, type, 1, ne, true); }
struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1) { const int64_t ne[2] = {ne0, ne1}; return new_tensor(ctx, type, 2, ne, true); }
struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2) { const int64_t ne[3] = {ne0, ne1, ne2}; return new_tensor(ctx, type, 3, ne, true); }
struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) { const int64_t ne[4] = {ne0, ne1, ne2, ne3}; return new_tensor(ctx, type, 4, ne, true); }
struct ggml_tensor * ggml_new_i32(struct ggml_context * ctx, int32_t value) { ggml_tensor * t = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, 1); *(int32_t *)t->data = value; return t; }
struct ggml_tensor * ggml_new_f32(struct ggml_context * ctx, float value) { ggml_tensor * t = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, 1); *(float *)t->data = value; return t; }
struct ggml_tensor * ggml_dup_tensor(struct ggml_context * ctx, const struct ggml_tensor * src) { return new_tensor(ctx, src->type, src->n_dims, src->ne, true); }
struct ggml_tensor * ggml_get_tensor(struct ggml_context * ctx, const char * name) {
    for (ggml_tensor * t : ctx->tensors) if (!strcmp(t->name, name)) return t;
    return nullptr;
}
struct ggml_tensor * ggml_set_name(struct ggml_tensor * t, const char * name) { snprintf(t->name, sizeof t->name, "%s", name); return t; }
struct ggml_tensor * ggml_set_zero(struct ggml_tensor * t) { memset(t->data, 0, ggml_nbytes(t)); return t; }
void ggml_set_i32_1d(const struct ggml_tensor * t, int i, int32_t value) {
    if (t->type != GGML_TYPE_I32) die("ggml_set_i32_1d on a non-i32 tensor");
    ((int32_t *)t->data)[i] = value;
}
void * ggml_get_data(const struct ggml_tensor * t) { return t->data; }
float * ggml_get_data_f32(const struct ggml_tensor * t) { need_f32(t, "ggml_get_data_f32"); return (float *)t->data; }

// ---- elementwise ----
struct ggml_tensor * ggml_add(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, a, b, "add", [](float x, float y) { return x + y; }); }
struct ggml_tensor * ggml_mul(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, a, b, "mul", [](float x, float y) { return x * y; }); }
struct ggml_tensor * ggml_div(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, a, b, "div", [](float x, float y) { return x / y; }); }
struct ggml_tensor * ggml_sqr(struct ggml_context * ctx, struct ggml_tensor * a) { return binary(ctx, a, a, "sqr", [](float x, float) { return x * x; }); }
struct ggml_tensor * ggml_sqrt(struct ggml_context * ctx, struct ggml_tensor * a) { return binary(ctx, a, a, "sqrt", [](float x, float) { return sqrtf(x); }); }
struct ggml_tensor * ggml_sum(struct ggml_context * ctx, struct ggml_tensor * a) {     // ggml_vec_sum_ggf: double accumulation
    need_f32(a, "sum");
    double s = 0.0;
    for (int64_t i3 = 0; i3 < a->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < a->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < a->ne[1]; i1++)
                for (int64_t i0 = 0; i0 < a->ne[0]; i0++) s += (double)*f32_at(a, i0, i1, i2, i3);
    return ggml_new_f32(ctx, (float)s);
}

// dst = a; dst[view at `offset` with strides nb1..nb3] += b   (b's shape)
struct ggml_tensor * ggml_acc(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, size_t nb1, size_t nb2, size_t nb3, size_t offset) {
    need_f32(a, "acc"); need_f32(b, "acc");
    if (!contiguous(a)) die("acc: a must be contiguous");
    ggml_tensor * c = like(ctx, a);
    memcpy(c->data, a->data, ggml_nbytes(a));
    for (int64_t i3 = 0; i3 < b->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < b->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < b->ne[1]; i1++) {
                const size_t o = offset + i1 * nb1 + i2 * nb2 + i3 * nb3;
                if (o + (size_t)b->ne[0] * 4 > ggml_nbytes(c)) die("acc: view outside the destination");
                float * d = (float *)((char *)c->data + o);
                for (int64_t i0 = 0; i0 < b->ne[0]; i0++) d[i0] += *f32_at(b, i0, i1, i2, i3);
            }
    return c;
}

struct ggml_tensor * ggml_repeat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    need_f32(a, "repeat");
    for (int i = 0; i < GGML_MAX_DIMS; i++) if (b->ne[i] % a->ne[i]) die("repeat: shape is not a multiple");
    ggml_tensor * c = new_tensor(ctx, GGML_TYPE_F32, b->n_dims, b->ne, true);
    for (int64_t i3 = 0; i3 < c->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < c->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < c->ne[1]; i1++)
                for (int64_t i0 = 0; i0 < c->ne[0]; i0++)
                    *f32_at(c, i0, i1, i2, i3) = *f32_at(a, i0 % a->ne[0], i1 % a->ne[1], i2 % a->ne[2], i3 % a->ne[3]);
    return c;
}

struct ggml_tensor * ggml_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps) {
    need_f32(a, "norm");
    if (!contiguous(a)) die("norm: contiguous input expected");
    ggml_tensor * c = like(ctx, a);
    const int64_t h = a->ne[0], rows = ggml_nelements(a) / h;
    for (int64_t r = 0; r < rows; r++) {
        const float * x = (const float *)a->data + r * h;
        float * y = (float *)c->data + r * h;
        double sum = 0.0;
        for (int64_t i = 0; i < h; i++) sum += (double)x[i];
        const float mean = (float)(sum / h);
        double sum2 = 0.0;
        for (int64_t i = 0; i < h; i++) { const float v = x[i] - mean; y[i] = v; sum2 += (double)(v * v); }
        const float variance = (float)(sum2 / h);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int64_t i = 0; i < h; i++) y[i] = y[i] * scale;
    }
    return c;
}

// a [K, N, a2, a3] x b [K, M, b2, b3] -> [N, M, b2, b3]; a is broadcast over b's batch dimensions
struct ggml_tensor * ggml_mul_mat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    need_f32(b, "mul_mat");
    if (a->ne[0] != b->ne[0]) die("mul_mat: inner dimensions differ");
    if (!contiguous(a) || !contiguous(b)) die("mul_mat: contiguous operands expected");
    if (b->ne[2] % a->ne[2] || b->ne[3] % a->ne[3]) die("mul_mat: batch dimensions do not broadcast");
    const int64_t K = a->ne[0], N = a->ne[1], M = b->ne[1];
    const int64_t ne[4] = {N, M, b->ne[2], b->ne[3]};
    ggml_tensor * c = new_tensor(ctx, GGML_TYPE_F32, b->n_dims < 2 ? 2 : b->n_dims, ne, true);
    if (a->ne[2] == 1 && a->ne[3] == 1) {
        // one weight matrix for every slice: all rows of b at once
        if (orc_mul_mat((int)a->type, a->data, N, K, (const float *)b->data, M * b->ne[2] * b->ne[3], (float *)c->data, 0, 0) != 1) die("mul_mat: unsupported weight type");
        return c;
    }
    const int64_t r2 = b->ne[2] / a->ne[2], r3 = b->ne[3] / a->ne[3];
    for (int64_t i3 = 0; i3 < b->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < b->ne[2]; i2++) {
            const char * as = (const char *)a->data + (i2 / r2) * a->nb[2] + (i3 / r3) * a->nb[3];
            const float * bs = (const float *)((const char *)b->data + i2 * b->nb[2] + i3 * b->nb[3]);
            float * cs = (float *)((char *)c->data + i2 * c->nb[2] + i3 * c->nb[3]);
            if (orc_mul_mat((int)a->type, as, N, K, bs, M, cs, 0, 0) != 1) die("mul_mat: unsupported type");
        }
    return c;
}

struct ggml_tensor * ggml_scale_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    need_f32(a, "scale"); need_f32(b, "scale");
    if (!contiguous(a)) die("scale: contiguous input expected");
    const float v = *(const float *)b->data;
    float * x = (float *)a->data;
    for (int64_t i = 0, n = ggml_nelements(a); i < n; i++) x[i] *= v;
    return view_of(ctx, a);
}
struct ggml_tensor * ggml_gelu_inplace(struct ggml_context * ctx, struct ggml_tensor * a) {
    need_f32(a, "gelu");
    if (!contiguous(a)) die("gelu: contiguous input expected");
    orc_activation((float *)a->data, ggml_nelements(a), 1, 0);
    return view_of(ctx, a);
}
struct ggml_tensor * ggml_gelu_quick_inplace(struct ggml_context * ctx, struct ggml_tensor * a) {
    need_f32(a, "gelu_quick");
    if (!contiguous(a)) die("gelu_quick: contiguous input expected");
    orc_activation((float *)a->data, ggml_nelements(a), 0, 0);
    return view_of(ctx, a);
}
struct ggml_tensor * ggml_soft_max_inplace(struct ggml_context * ctx, struct ggml_tensor * a) {
    need_f32(a, "soft_max");
    if (!contiguous(a)) die("soft_max: contiguous input expected");
    orc_softmax_rows((float *)a->data, ggml_nelements(a) / a->ne[0], a->ne[0], 0);
    return view_of(ctx, a);
}
struct ggml_tensor * ggml_diag_mask_inf_inplace(struct ggml_context * ctx, struct ggml_tensor * a, int n_past) {
    need_f32(a, "diag_mask_inf");
    for (int64_t i3 = 0; i3 < a->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < a->ne[2]; i2++)
            for (int64_t j = 0; j < a->ne[1]; j++)
                for (int64_t i = n_past + j + 1; i < a->ne[0]; i++) *f32_at(a, i, j, i2, i3) = -INFINITY;
    return view_of(ctx, a);
}

// ---- data movement ----
struct ggml_tensor * ggml_cont(struct ggml_context * ctx, struct ggml_tensor * a) {
    need_f32(a, "cont");
    ggml_tensor * c = like(ctx, a);
    float * d = (float *)c->data;
    for (int64_t i3 = 0; i3 < a->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < a->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < a->ne[1]; i1++)
                for (int64_t i0 = 0; i0 < a->ne[0]; i0++) *d++ = *f32_at(a, i0, i1, i2, i3);
    return c;
}
struct ggml_tensor * ggml_cpy(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    need_f32(a, "cpy"); need_f32(b, "cpy");
    if (ggml_nelements(a) != ggml_nelements(b)) die("cpy: element counts differ");
    if (!contiguous(b)) die("cpy: contiguous destination expected");
    float * d = (float *)b->data;               // logical element order of a into the flat destination
    for (int64_t i3 = 0; i3 < a->ne[3]; i3++)
        for (int64_t i2 = 0; i2 < a->ne[2]; i2++)
            for (int64_t i1 = 0; i1 < a->ne[1]; i1++)
                for (int64_t i0 = 0; i0 < a->ne[0]; i0++) *d++ = *f32_at(a, i0, i1, i2, i3);
    return view_of(ctx, b);
}
static struct ggml_tensor * reshape(struct ggml_context * ctx, struct ggml_tensor * a, int n_dims, const int64_t * ne) {
    if (!contiguous(a)) die("reshape: contiguous input expected");
    int64_t n = 1;
    for (int i = 0; i < n_dims; i++) n *= ne[i];
    if (n != ggml_nelements(a)) die("reshape: element counts differ");
    ggml_tensor * t = new_tensor(ctx, a->type, n_dims, ne, false);
    t->data = a->data;
    return t;
}
struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1) { const int64_t ne[2] = {ne0, ne1}; return reshape(ctx, a, 2, ne); }
struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2) { const int64_t ne[3] = {ne0, ne1, ne2}; return reshape(ctx, a, 3, ne); }
struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) { const int64_t ne[4] = {ne0, ne1, ne2, ne3}; return reshape(ctx, a, 4, ne); }
struct ggml_tensor * ggml_permute(struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3) {
    const int ax[4] = {axis0, axis1, axis2, axis3};
    bool seen[4] = {false, false, false, false};
    for (int i = 0; i < 4; i++) { if (ax[i] < 0 || ax[i] > 3 || seen[ax[i]]) die("permute: not a permutation"); seen[ax[i]] = true; }
    ggml_tensor * t = view_of(ctx, a);
    t->n_dims = 4;
    for (int i = 0; i < 4; i++) { t->ne[ax[i]] = a->ne[i]; t->nb[ax[i]] = a->nb[i]; }     // dimension i of a becomes dimension ax[i]
    return t;
}
struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    if (b->type != GGML_TYPE_I32) die("get_rows: i32 indices expected");
    if (!contiguous(a)) die("get_rows: contiguous source expected");
    const int64_t k = a->ne[0], nrows = a->ne[1] * a->ne[2] * a->ne[3], n = ggml_nelements(b);
    const int64_t ne[2] = {k, n};
    ggml_tensor * c = new_tensor(ctx, GGML_TYPE_F32, 2, ne, true);
    for (int64_t i = 0; i < n; i++) {
        const int32_t r = ((const int32_t *)b->data)[i];
        if (r < 0 || r >= nrows) die("get_rows: index out of range");
        if (a->type == GGML_TYPE_F32) memcpy((float *)c->data + i * k, (const char *)a->data + (size_t)r * a->nb[1], (size_t)k * 4);
        else orc_dequantize((int)a->type, (const char *)a->data + (size_t)r * a->nb[1], (float *)c->data + i * k, 1, k);
    }
    return c;
}

// a: kernel [KW, KH, IC, OC] f16;  b: input [W, H, IC, N] f32  ->  [OW, OH, OC, N] f32; fp16 im2col, k = (c, ky, kx)
struct ggml_tensor * ggml_conv_2d(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, int s0, int s1, int p0, int p1, int d0, int d1) {
    need_f32(b, "conv_2d");
    if (a->type != GGML_TYPE_F16 && a->type != GGML_TYPE_F32) die("conv_2d: f16 / f32 kernel expected");
    if (p0 || p1 || d0 != 1 || d1 != 1) die("conv_2d: only the unpadded, undilated form the reference uses");
    if (a->ne[2] != b->ne[2]) die("conv_2d: channel counts differ");
    if (!contiguous(a) || !contiguous(b)) die("conv_2d: contiguous operands expected");
    const int64_t KW = a->ne[0], KH = a->ne[1], IC = a->ne[2], OC = a->ne[3], W = b->ne[0], H = b->ne[1], N = b->ne[3];
    const int64_t OW = (W - KW) / s0 + 1, OH = (H - KH) / s1 + 1, Kp = IC * KH * KW;
    const int64_t ne[4] = {OW, OH, OC, N};
    ggml_tensor * c = new_tensor(ctx, GGML_TYPE_F32, 4, ne, true);
    std::vector<float> w((size_t)OC * Kp);
    if (a->type == GGML_TYPE_F16) orc_dequantize(GGML_TYPE_F16, a->data, w.data(), 1, OC * Kp);
    else memcpy(w.data(), a->data, w.size() * 4);
    const float * in = (const float *)b->data;
    float * out = (float *)c->data;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t n = 0; n < N; n++)
        for (int64_t oy = 0; oy < OH; oy++) {
            std::vector<float> col((size_t)Kp);
            for (int64_t ox = 0; ox < OW; ox++) {
                for (int64_t ch = 0; ch < IC; ch++)
                    for (int64_t ky = 0; ky < KH; ky++)
                        for (int64_t kx = 0; kx < KW; kx++) {
                            const float v = in[((n * IC + ch) * H + (oy * s1 + ky)) * W + (ox * s0 + kx)];
                            col[(ch * KH + ky) * KW + kx] = orc_h2f(orc_f2h(v));
                        }
                for (int64_t oc = 0; oc < OC; oc++) {
                    const float * wr = w.data() + oc * Kp;
                    double s = 0.0;
                    for (int64_t k = 0; k < Kp; k++) s += (double)(wr[k] * col[k]);
                    out[((n * OC + oc) * OH + oy) * OW + ox] = (float)s;
                }
            }
        }
    return c;
}

// ---- graph: nothing left to do ----
void ggml_build_forward_expand(struct ggml_cgraph * g, struct ggml_tensor * t) { g->n_nodes++; g->last = t; }
struct ggml_cplan ggml_graph_plan(struct ggml_cgraph *, int n_threads) { ggml_cplan p; p.work_size = 0; p.work_data = nullptr; p.n_threads = n_threads; return p; }
int ggml_graph_compute(struct ggml_cgraph *, struct ggml_cplan *) { return 0; }

// ---- the quantize tool's side: ggml_quantize_* = quantize_row_*_reference over rows of k (the oracle's codecs; the histogram is not kept) ----
size_t orc_quantize(int type, const float * src, void * dst, int64_t nrows, int64_t k);
static size_t quantize_rows_(int type, const float * src, void * dst, int n, int k) { return orc_quantize(type, src, dst, n / k, k); }
size_t ggml_quantize_q4_0(const float * src, void * dst, int n, int k, int64_t *) { return quantize_rows_(GGML_TYPE_Q4_0, src, dst, n, k); }
size_t ggml_quantize_q4_1(const float * src, void * dst, int n, int k, int64_t *) { return quantize_rows_(GGML_TYPE_Q4_1, src, dst, n, k); }
size_t ggml_quantize_q5_0(const float * src, void * dst, int n, int k, int64_t *) { return quantize_rows_(GGML_TYPE_Q5_0, src, dst, n, k); }
size_t ggml_quantize_q5_1(const float * src, void * dst, int n, int k, int64_t *) { return quantize_rows_(GGML_TYPE_Q5_1, src, dst, n, k); }
size_t ggml_quantize_q8_0(const float * src, void * dst, int n, int k, int64_t *) { return quantize_rows_(GGML_TYPE_Q8_0, src, dst, n, k); }

// ---- gguf writer: header + key/values + tensor infos, padded to the alignment (the caller writes the tensor data itself) ----
struct gguf_context * gguf_init_empty(void) { gguf_context * g = new gguf_context(); g->version = 2; return g; }
void gguf_set_kv(struct gguf_context * ctx, struct gguf_context * src) {
    for (const Kv & kv : src->kv) {
        bool found = false;
        for (Kv & mine : ctx->kv) if (mine.key == kv.key) { mine = kv; found = true; break; }
        if (!found) ctx->kv.push_back(kv);
    }
    ctx->version = src->version;
    ctx->alignment = src->alignment;
}
void gguf_set_val_u32(struct gguf_context * ctx, const char * key, uint32_t val) {
    Kv kv;
    kv.key = key; kv.type = 4; kv.scalar = val;
    for (Kv & mine : ctx->kv) if (mine.key == kv.key) { mine = kv; return; }
    ctx->kv.push_back(kv);
}
void gguf_add_tensor(struct gguf_context * ctx, const struct ggml_tensor * t) {
    gguf_context::Out o;
    o.name = t->name; o.type = (uint32_t)t->type; o.n_dims = (uint32_t)t->n_dims; o.size = ggml_nbytes(t);
    for (int i = 0; i < 4; i++) o.ne[i] = t->ne[i];
    ctx->out.push_back(o);
}
static gguf_context::Out * out_by_name(struct gguf_context * ctx, const char * name) {
    for (auto & o : ctx->out) if (o.name == name) return &o;
    die("gguf writer: unknown tensor name");
}
void gguf_set_tensor_type(struct gguf_context * ctx, const char * name, enum ggml_type type) {
    gguf_context::Out * o = out_by_name(ctx, name);
    o->type = (uint32_t)type;
    o->size = type_size(type) * (size_t)(o->ne[0] / blck(type)) * (size_t)(o->ne[1] * o->ne[2] * o->ne[3]);
}
void gguf_set_tensor_data(struct gguf_context * ctx, const char * name, const void *, size_t size) {
    if (out_by_name(ctx, name)->size != size) die("gguf writer: data size does not match the tensor type");
}
static void encode_meta(const struct gguf_context * ctx, std::vector<uint8_t> & b) {
    auto put = [&](const void * p, size_t n) { b.insert(b.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    auto pstr = [&](const std::string & v) { const uint64_t n = v.size(); put(&n, 8); put(v.data(), v.size()); };
    put("GGUF", 4);
    put(&ctx->version, 4);
    const uint64_t nt = ctx->out.size(), nkv = ctx->kv.size();
    put(&nt, 8); put(&nkv, 8);
    for (const Kv & kv : ctx->kv) {
        pstr(kv.key);
        put(&kv.type, 4);
        if (kv.type == 8) pstr(kv.str);
        else if (kv.type == 9) {
            put(&kv.elem_type, 4); put(&kv.n, 8);
            if (kv.elem_type == 8) for (const std::string & e : kv.strs) pstr(e);
            else put(kv.raw.data(), kv.raw.size());
        } else put(&kv.scalar, scalar_size(kv.type));
    }
    uint64_t off = 0;
    for (const auto & o : ctx->out) {
        pstr(o.name);
        put(&o.n_dims, 4);
        for (uint32_t k = 0; k < o.n_dims; k++) { const uint64_t v = (uint64_t)o.ne[k]; put(&v, 8); }
        put(&o.type, 4);
        put(&off, 8);
        off += (o.size + ctx->alignment - 1) / ctx->alignment * ctx->alignment;
    }
    b.resize((b.size() + ctx->alignment - 1) / ctx->alignment * ctx->alignment, 0);
}
size_t gguf_get_meta_size(const struct gguf_context * ctx) { std::vector<uint8_t> b; encode_meta(ctx, b); return b.size(); }
void gguf_get_meta_data(const struct gguf_context * ctx, void * data) { std::vector<uint8_t> b; encode_meta(ctx, b); memcpy(data, b.data(), b.size()); }

// ---- gguf reader (container layout: SURVEY.md Appendix A) ----
struct gguf_context * gguf_init_from_file(const char * fname, struct gguf_init_params params) {
    FILE * f = fopen(fname, "rb");
    if (!f) return nullptr;
    std::vector<uint8_t> d;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    // the header region only: tensor data is read by the caller through its own stream
    const size_t head = (size_t)std::min<long>(sz, 64l << 20);
    d.resize(head);
    if (fread(d.data(), 1, head, f) != head) { fclose(f); return nullptr; }
    fclose(f);
    size_t p = 0;
    bool bad = false;
    auto rd = [&](void * out, size_t n) { if (p + n > d.size()) { bad = true; memset(out, 0, n); return; } memcpy(out, d.data() + p, n); p += n; };
    auto rstr = [&]() { uint64_t n = 0; rd(&n, 8); std::string s; if (bad || p + n > d.size()) { bad = true; return s; } s.assign((const char *)d.data() + p, n); p += n; return s; };
    char magic[4]; rd(magic, 4);
    if (memcmp(magic, "GGUF", 4) != 0) return nullptr;
    gguf_context * g = new gguf_context();
    rd(&g->version, 4);
    uint64_t n_tensors = 0, n_kv = 0;
    rd(&n_tensors, 8); rd(&n_kv, 8);
    for (uint64_t i = 0; i < n_kv && !bad; i++) {
        Kv kv;
        kv.key = rstr();
        rd(&kv.type, 4);
        if (kv.type == 8) kv.str = rstr();
        else if (kv.type == 9) {
            rd(&kv.elem_type, 4); rd(&kv.n, 8);
            if (kv.elem_type == 8) for (uint64_t k = 0; k < kv.n && !bad; k++) kv.strs.push_back(rstr());
            else { const size_t es = scalar_size(kv.elem_type); if (!es) bad = true; kv.raw.resize(es * kv.n); rd(kv.raw.data(), kv.raw.size()); }
        } else { const size_t es = scalar_size(kv.type); if (!es) bad = true; rd(&kv.scalar, es); }
        g->kv.push_back(std::move(kv));
    }
    ggml_context * meta = nullptr;
    if (params.ctx) { ggml_init_params ip = {0, nullptr, true}; meta = ggml_init(ip); *params.ctx = meta; }
    for (uint64_t i = 0; i < n_tensors && !bad; i++) {
        TInfo ti;
        ti.name = rstr();
        uint32_t nd = 0, type = 0;
        rd(&nd, 4);
        int64_t ne[4] = {1, 1, 1, 1};
        if (nd > 4) { bad = true; break; }
        for (uint32_t k = 0; k < nd; k++) { uint64_t v = 0; rd(&v, 8); ne[k] = (int64_t)v; }
        rd(&type, 4); rd(&ti.offset, 8);
        if (meta && !bad) { ggml_tensor * t = new_tensor(meta, (ggml_type)type, (int)nd, ne, false); ggml_set_name(t, ti.name.c_str()); }
        g->infos.push_back(std::move(ti));
    }
    if (bad) { fprintf(stderr, "ggml shim: malformed or oversized GGUF header in %s\n", fname); delete g; return nullptr; }
    const int ia = gguf_find_key(g, "general.alignment");
    if (ia >= 0) g->alignment = (size_t)gguf_get_val_u32(g, ia);
    g->data_offset = (p + g->alignment - 1) / g->alignment * g->alignment;
    return g;
}
void gguf_free(struct gguf_context * ctx) { delete ctx; }
int gguf_get_version(const struct gguf_context * ctx) { return (int)ctx->version; }
size_t gguf_get_alignment(const struct gguf_context * ctx) { return ctx->alignment; }
size_t gguf_get_data_offset(const struct gguf_context * ctx) { return ctx->data_offset; }
int gguf_get_n_kv(const struct gguf_context * ctx) { return (int)ctx->kv.size(); }
int gguf_find_key(const struct gguf_context * ctx, const char * key) {
    for (size_t i = 0; i < ctx->kv.size(); i++) if (ctx->kv[i].key == key) return (int)i;
    return -1;
}
const char * gguf_get_key(const struct gguf_context * ctx, int i) { return ctx->kv[i].key.c_str(); }
uint32_t gguf_get_val_u32(const struct gguf_context * ctx, int i) { return (uint32_t)ctx->kv[i].scalar; }
float gguf_get_val_f32(const struct gguf_context * ctx, int i) { float v; const uint32_t b = (uint32_t)ctx->kv[i].scalar; memcpy(&v, &b, 4); return v; }
bool gguf_get_val_bool(const struct gguf_context * ctx, int i) { return (ctx->kv[i].scalar & 0xFF) != 0; }
const char * gguf_get_val_str(const struct gguf_context * ctx, int i) { return ctx->kv[i].str.c_str(); }
int gguf_get_arr_n(const struct gguf_context * ctx, int i) { return (int)ctx->kv[i].n; }
const void * gguf_get_arr_data(const struct gguf_context * ctx, int i) { return ctx->kv[i].raw.data(); }
const char * gguf_get_arr_str(const struct gguf_context * ctx, int key_id, int i) { return ctx->kv[key_id].strs[i].c_str(); }
int gguf_get_n_tensors(const struct gguf_context * ctx) { return (int)ctx->infos.size(); }
size_t gguf_get_tensor_offset(const struct gguf_context * ctx, int i) { return (size_t)ctx->infos[i].offset; }
char * gguf_get_tensor_name(const struct gguf_context * ctx, int i) { return (char *)ctx->infos[i].name.c_str(); }

}  // extern "C"
