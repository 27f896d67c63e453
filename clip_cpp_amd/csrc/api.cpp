// api.cpp — the extern "C" boundary: every symbol of include/clip.h (= reference clip.h:42-109) plus
// the MI355X extensions of include/clip_amd.h.  Thin shim: argument checking, host<->HBM staging,
// then forward.cpp.  Nothing here throws across the ABI: every entry point that parses a file or allocates in
// proportion to caller input is a function-try-block that turns an exception (bad_alloc, length_error, ...) into NULL / false.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/clip_amd.h"
#include "model.h"

using namespace clipamd;

namespace {

int default_device() {
    const char * e = getenv("CLIP_AMD_DEVICE");
    if (!e || !*e) e = getenv("LOCAL_RANK");
    if (e && *e) {
        int n = 0;
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0) return atoi(e) % n;
        (void)hipGetLastError();
    }
    return 0;
}

// RAII device buffer for the test hooks
struct DBuf {
    void * p = nullptr;
    explicit DBuf(size_t n) { if (hipMalloc(&p, n ? n : 16) != hipSuccess) p = nullptr; }
    ~DBuf() { if (p) (void)hipFree(p); }
};

std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();

}  // namespace

extern "C" {

// ---- ggml timing shim (include/ggml/ggml.h) ----
void ggml_time_init(void) { g_t0 = std::chrono::steady_clock::now(); }
int64_t ggml_time_us(void) {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - g_t0).count();
}
int64_t ggml_time_ms(void) { return ggml_time_us() / 1000; }

// ---- model lifetime ----
struct clip_ctx * clip_model_load(const char * fname, const int verbosity) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!fname) return nullptr;
    return load_model(fname, verbosity, default_device());
} catch (const std::exception & e) { fprintf(stderr, "clip_model_load: %s\n", e.what()); return nullptr; } catch (...) { fprintf(stderr, "clip_model_load: unknown exception\n"); return nullptr; }
struct clip_ctx * clip_amd_model_load(const char * fname, int verbosity, int device) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!fname) return nullptr;
    return load_model(fname, verbosity, device);
} catch (const std::exception & e) { fprintf(stderr, "clip_amd_model_load: %s\n", e.what()); return nullptr; } catch (...) { fprintf(stderr, "clip_amd_model_load: unknown exception\n"); return nullptr; }
void clip_free(struct clip_ctx * ctx) { RelaxCapture relax_capture; free_model(ctx); }
struct clip_text_hparams * clip_get_text_hparams(struct clip_ctx * ctx) { return &ctx->text_hparams; }
struct clip_vision_hparams * clip_get_vision_hparams(struct clip_ctx * ctx) { return &ctx->vision_hparams; }

int clip_amd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int clip_amd_ctx_device(const struct clip_ctx * ctx) { return ctx ? ctx->device : -1; }
void clip_amd_set_device_shared(struct clip_ctx * ctx, int shared) {
    if (ctx) ctx->device_shared = shared != 0;
}

void clip_amd_set_stream(struct clip_ctx * ctx, void * hip_stream) {
    if (!ctx || ctx->device < 0) return;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next == ctx->stream) return;
    // A context owns ONE workspace (activations, staging buffers, split-K tickets): work queued on the new stream must not
    // start before what is still queued on the previous one has finished.
    (void)hipSetDevice(ctx->device);
    if (!ctx->ev_stream_switch) (void)hipEventCreateWithFlags(&ctx->ev_stream_switch, hipEventDisableTiming);
    if (ctx->ev_stream_switch && hipEventRecord(ctx->ev_stream_switch, ctx->stream) == hipSuccess &&
        hipStreamWaitEvent(next, ctx->ev_stream_switch, 0) == hipSuccess) {
    } else {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
    }
    ctx->stream = next;
}
struct clip_ctx * clip_amd_model_load_multi(const char * fname, int verbosity, int n_devices) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!fname) return nullptr;
    return multi_load(fname, verbosity, n_devices);
} catch (const std::exception & e) { fprintf(stderr, "clip_amd_model_load_multi: %s\n", e.what()); return nullptr; } catch (...) { fprintf(stderr, "clip_amd_model_load_multi: unknown exception\n"); return nullptr; }
int clip_amd_ctx_device_count(const struct clip_ctx * ctx) { return ctx ? multi_device_count(ctx) : 0; }
int clip_amd_weights_from_cache(const struct clip_ctx * ctx) { return ctx && ctx->weights_from_cache ? 1 : 0; }
void clip_amd_shard_bounds(int total, int n_devices, int device_index, int * lo, int * hi, int * rows_per_device) {
    int l = 0, h = 0, p = 0;
    if (n_devices > 0 && device_index >= 0 && device_index < n_devices && total >= 0) multi_shard(total, n_devices, device_index, &l, &h, &p);
    if (lo) *lo = l;
    if (hi) *hi = h;
    if (rows_per_device) *rows_per_device = p;
}
const float * clip_amd_gathered_embeddings(const struct clip_ctx * ctx, int device_index) { return ctx ? multi_gathered(ctx, device_index) : nullptr; }
void clip_amd_synchronize(struct clip_ctx * ctx) {
    if (ctx && ctx->device >= 0) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
}

// ---- tokenizer ----
bool clip_tokenize(const struct clip_ctx * ctx, const char * text, struct clip_tokens * tokens) try {
    if (!ctx->has_text_encoder) {
        printf("This GGUF file seems to have no text encoder\n");
        return false;
    }
    std::vector<int32_t> v;
    if (!tokenize_text(ctx, text, v)) return false;
    tokens->size = v.size();
    tokens->data = new clip_vocab_id[v.size()];   // caller-owned, as in the reference (clip.cpp:675)
    std::copy(v.begin(), v.end(), tokens->data);
    return true;
} catch (const std::exception & e) { fprintf(stderr, "clip_tokenize: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_tokenize: unknown exception\n"); return false; }

// ---- image containers ----
struct clip_image_u8 * clip_image_u8_make() { return new clip_image_u8(); }
struct clip_image_f32 * clip_image_f32_make() { return new clip_image_f32(); }
void clip_image_u8_clean(struct clip_image_u8 * img) {
    if (img && img->data) { delete[] img->data; img->data = nullptr; }
}
void clip_image_f32_clean(struct clip_image_f32 * res) {
    if (res && res->data) { delete[] res->data; res->data = nullptr; }
}
void clip_image_u8_free(struct clip_image_u8 * img) { if (img) { clip_image_u8_clean(img); delete img; } }
void clip_image_f32_free(struct clip_image_f32 * res) { if (res) { clip_image_f32_clean(res); delete res; } }

bool clip_image_load_from_file(const char * fname, struct clip_image_u8 * img) try { return load_image_file(fname, img); } catch (const std::exception & e) { fprintf(stderr, "clip_image_load_from_file: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_image_load_from_file: unknown exception\n"); return false; }
bool clip_image_preprocess(const struct clip_ctx * ctx, const struct clip_image_u8 * img, struct clip_image_f32 * res) {
    return preprocess_image(ctx, img, res);
}
void clip_image_batch_preprocess(const struct clip_ctx * ctx, const int n_threads, const struct clip_image_u8_batch * img_inputs,
                                 struct clip_image_f32_batch * imgs_resized) {
    imgs_resized->size = img_inputs->size;
    const size_t n = img_inputs->size;
    const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_threads), n));
    if (nt == 1) {
        for (size_t i = 0; i < n; i++) preprocess_image(ctx, &img_inputs->data[i], &imgs_resized->data[i]);
        return;
    }
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nt; t++)
        pool.emplace_back([=]() {
            for (size_t i = t; i < n; i += nt) preprocess_image(ctx, &img_inputs->data[i], &imgs_resized->data[i]);
        });
    for (auto & th : pool) th.join();
}

// ---- encoders: host-pointer forms (the reference API) ----
bool clip_image_batch_encode(const struct clip_ctx * cctx, const int n_threads, const struct clip_image_f32_batch * imgs, float * vec,
                             const bool normalize) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    (void)n_threads;
    clip_ctx * ctx = const_cast<clip_ctx *>(cctx);  // const handle, mutable workspace — as in the reference
    if (!ctx->has_vision_encoder) {
        printf("This gguf file seems to have no vision encoder\n");
        return false;
    }
    if (ctx->device < 0) {
        fprintf(stderr, "clip_image_batch_encode: no HIP device bound to this context — the encoders have no CPU fallback\n");
        return false;
    }
    const int B = (int)imgs->size;
    if (B <= 0) return true;
    const int S = ctx->vision_hparams.image_size, proj = ctx->vision_hparams.projection_dim;
    for (int b = 0; b < B; b++) {
        if (imgs->data[b].nx != S || imgs->data[b].ny != S || !imgs->data[b].data) {
            fprintf(stderr, "clip_image_batch_encode: image %d is %dx%d, expected %dx%d (run clip_image_preprocess first)\n", b,
                    imgs->data[b].nx, imgs->data[b].ny, S, S);
            return false;   // the reference GGML_ASSERTs here (clip.cpp:1293)
        }
    }
    (void)hipSetDevice(ctx->device);
    const int nt = std::max(1, n_threads);
    if (ctx->multi && B >= 2 * multi_device_count(ctx)) {
        // sharded over the devices of a clip_amd_model_load_multi context: ceil(B/G) images per device, one RCCL all-gather
        const bool mok = multi_image_batch_encode(ctx, imgs->data, B, vec, normalize, nt);
        if (!mok) fprintf(stderr, "clip_image_batch_encode: multi-device encode failed: %s\n", hipGetErrorString(hipGetLastError()));
        return mok;
    }
    if (!ensure_io(ctx, 16, (size_t)proj * 4 * B)) {
        fprintf(stderr, "clip_image_batch_encode: out of device memory\n");
        return false;
    }
    // pinned, threaded, double-buffered staging (host_pipeline.cpp): pack(k+1) || H2D(k) || forward(k-1)
    bool ok = encode_images_from_host(ctx, imgs->data, B, (float *)ctx->io_out, normalize, nt);
    ok = ok && hipMemcpyAsync(vec, ctx->io_out, (size_t)proj * 4 * B, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    ok = hipStreamSynchronize(ctx->stream) == hipSuccess && ok;
    if (!ok) fprintf(stderr, "clip_image_batch_encode: HIP error: %s\n", hipGetErrorString(hipGetLastError()));
    if (ctx->profiling) prof_collect(ctx);
    return ok;
} catch (const std::exception & e) { fprintf(stderr, "clip_image_batch_encode: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_image_batch_encode: unknown exception\n"); return false; }

bool clip_image_encode(const struct clip_ctx * ctx, const int n_threads, struct clip_image_f32 * img, float * vec, const bool normalize) {
    if (!ctx->has_vision_encoder) {
        printf("This gguf file seems to have no vision encoder\n");
        return false;
    }
    clip_image_f32_batch b{};
    b.size = 1;
    b.data = img;
    return clip_image_batch_encode(ctx, n_threads, &b, vec, normalize);
}

bool clip_amd_image_batch_encode_device(struct clip_ctx * ctx, const float * d_imgs, int batch, float * d_out, bool normalize) {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    return vision_forward_device(ctx, d_imgs, batch, d_out, normalize);
}

bool clip_amd_image_batch_preprocess_device(struct clip_ctx * ctx, const struct clip_image_u8 * imgs, int n, float * d_out) {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    return preprocess_batch_device(ctx, imgs, n, d_out);
}

// raw u8 images -> embeddings with the resize/crop/normalise on the GPU (bit-identical to clip_image_preprocess):
// ships <= 3 B/pixel of the ORIGINAL image instead of 12 B/pixel of the resized one and takes the double-precision
// resampling (the dominant host cost of benchmark.cpp / zsl.cpp style callers) off the CPU.
// n raw images -> d_out [n][proj] on ctx's device, queued on ctx->stream.
static bool encode_u8_to_device(clip_ctx * ctx, const clip_image_u8 * imgs, int n, float * d_out, bool normalize) {
    const int S = ctx->vision_hparams.image_size, proj = ctx->vision_hparams.projection_dim;
    const size_t per = (size_t)S * S * 3;
    (void)hipSetDevice(ctx->device);
    bool ok = true;
    int b0 = 0, piece_idx = 0;
    // Two granularities (as the f32 host pipeline, host_pipeline.cpp).  FORWARD chunks of <= 256 images / ~512 MB of raw pixels: one vision
    // forward each.  STAGING pieces of <= 128 images (CLIP_AMD_U8_PIECE) inside a chunk: the host fills pinned slot (piece % 4) while the previous pieces' H2D
    // (copy stream) and preprocessing kernels run, so only the first piece's host copy and the last piece's H2D are exposed before the
    // forward — and the pieces of chunk c + 1 are staged under chunk c's forward.  The preprocessed f32 images of a chunk land in ONE
    // buffer (io_in): the kernels that write it for chunk c + 1 are stream-ordered behind the forward that reads it for chunk c.
    static const int piece_max = [] { const char * e = getenv("CLIP_AMD_U8_PIECE"); const int v = e ? atoi(e) : 128; return v < 8 ? 8 : v > 256 ? 256 : v; }();   // tuning
    const bool pipelined = n > piece_max;
    while (b0 < n && ok) {
        int bc = 0;
        size_t bytes = 0;
        while (b0 + bc < n && bc < 256 && (bc == 0 || bytes < ((size_t)512 << 20))) {
            bytes += (size_t)3 * (size_t)std::max(0, imgs[b0 + bc].nx) * (size_t)std::max(0, imgs[b0 + bc].ny);
            bc++;
        }
        if (!ensure_io(ctx, per * 4 * std::min(n, 256), 16)) {
            fprintf(stderr, "clip_amd_image_batch_encode_u8: out of device memory\n");
            return false;
        }
        for (int p0 = 0; p0 < bc && ok;) {
            int pn = 0;
            size_t pbytes = 0;
            while (p0 + pn < bc && pn < (pipelined ? piece_max : bc) && (pn == 0 || pbytes < ((size_t)128 << 20))) {
                pbytes += (size_t)3 * (size_t)std::max(0, imgs[b0 + p0 + pn].nx) * (size_t)std::max(0, imgs[b0 + p0 + pn].ny);
                pn++;
            }
            ok = ok && preprocess_batch_device(ctx, imgs + b0 + p0, pn, (float *)ctx->io_in + (size_t)p0 * per, pipelined ? piece_idx : -1);
            p0 += pn;
            piece_idx++;
        }
        ok = ok && vision_forward_device(ctx, (const float *)ctx->io_in, bc, d_out + (size_t)b0 * proj, normalize);
        b0 += bc;
    }
    return ok;
}

bool clip_amd_image_batch_encode_u8(struct clip_ctx * ctx, const struct clip_image_u8 * imgs, int n, float * vec, bool normalize) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!ctx->has_vision_encoder) {
        printf("This gguf file seems to have no vision encoder\n");
        return false;
    }
    if (ctx->device < 0) {
        fprintf(stderr, "clip_amd_image_batch_encode_u8: no HIP device bound to this context — the encoders have no CPU fallback\n");
        return false;
    }
    if (n <= 0) return true;
    const int proj = ctx->vision_hparams.projection_dim;
    bool ok;
    if (ctx->multi && n >= 2 * multi_device_count(ctx)) {
        // sharded over the devices of a clip_amd_model_load_multi context (the ~150 KB / image entry point is the one
        // that can scale; the f32 host path is PCIe- and packer-bound on one GPU already): each replica preprocesses and encodes its own
        // contiguous shard, ONE all-gather of the embeddings
        ok = multi_run(ctx, n, proj, vec, "clip_amd_image_batch_encode_u8",
                       [&](int, clip_ctx * c, int l, int h, float * d_send) { return encode_u8_to_device(c, imgs + l, h - l, d_send, normalize); });
    } else {
        (void)hipSetDevice(ctx->device);
        ok = ensure_io(ctx, 16, (size_t)proj * 4 * n);
        ok = ok && encode_u8_to_device(ctx, imgs, n, (float *)ctx->io_out, normalize);
        ok = ok && hipMemcpyAsync(vec, ctx->io_out, (size_t)proj * 4 * n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!ok) fprintf(stderr, "clip_amd_image_batch_encode_u8: failed (%s)\n", hipGetErrorString(hipGetLastError()));
    if (ctx->profiling) prof_collect(ctx);
    return ok;
} catch (const std::exception & e) { fprintf(stderr, "clip_amd_image_batch_encode_u8: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_amd_image_batch_encode_u8: unknown exception\n"); return false; }

// Device-resident sharded image encode on a clip_amd_model_load_multi context (the measured form of SURVEY 8e: bench.py --single-process).
// d_imgs[g]: the preprocessed f32 images of shard g ([hi - lo][S][S][3], clip_amd_shard_bounds(total, G, g)) ON device g.
bool clip_amd_image_batch_encode_device_multi(struct clip_ctx * ctx, const float * const * d_imgs, int total, bool normalize, float * vec) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!ctx || !ctx->multi || !ctx->has_vision_encoder || !d_imgs) { fprintf(stderr, "clip_amd_image_batch_encode_device_multi: needs a clip_amd_model_load_multi context with a vision encoder\n"); return false; }
    if (total <= 0) return true;
    return multi_run(ctx, total, ctx->vision_hparams.projection_dim, vec, "clip_amd_image_batch_encode_device_multi",
                     [&](int g, clip_ctx * c, int l, int h, float * d_send) { return vision_forward_device(c, d_imgs[g], h - l, d_send, normalize); });
} catch (...) { fprintf(stderr, "clip_amd_image_batch_encode_device_multi: exception\n"); return false; }

// ... and of ragged texts: d_ids[g] = the ids of shard g's texts back to back ON device g, h_offsets = total + 1 prefix offsets of ALL texts (host).
bool clip_amd_text_batch_encode_device_multi(struct clip_ctx * ctx, const int32_t * const * d_ids, const int32_t * h_offsets, int total, bool normalize,
                                             float * vec) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!ctx || !ctx->multi || !ctx->has_text_encoder || !d_ids || !h_offsets) { fprintf(stderr, "clip_amd_text_batch_encode_device_multi: needs a clip_amd_model_load_multi context with a text encoder\n"); return false; }
    if (total <= 0) return true;
    const int G = multi_device_count(ctx);
    std::vector<std::vector<int32_t>> off(G);          // per-shard offsets rebased to 0 (kept alive until multi_run has synchronised)
    return multi_run(ctx, total, ctx->text_hparams.projection_dim, vec, "clip_amd_text_batch_encode_device_multi",
                     [&](int g, clip_ctx * c, int l, int h, float * d_send) {
                         off[g].resize((size_t)(h - l) + 1);
                         for (int i = l; i <= h; i++) off[g][(size_t)(i - l)] = h_offsets[i] - h_offsets[l];
                         return text_forward_device(c, d_ids[g], off[g].data(), h - l, d_send, normalize);
                     });
} catch (...) { fprintf(stderr, "clip_amd_text_batch_encode_device_multi: exception\n"); return false; }

// Both towers of a step on a clip_amd_model_load_multi context, device-resident shards (bench.py --single-process): on every device the
// vision tower (reference clip.cpp:1247) and the text tower (:1016) run on two streams (replica context + a twin context of the same
// device), ONE grouped all-gather carries both towers' rows.  vec_img [n_images][proj] / vec_txt [n_texts][proj] on the host, or NULL.
bool clip_amd_encode_pair_device_multi(struct clip_ctx * ctx, const float * const * d_imgs, int n_images, const int32_t * const * d_ids,
                                       const int32_t * h_offsets, int n_texts, bool normalize, float * vec_img, float * vec_txt) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!ctx || !ctx->multi || !ctx->has_vision_encoder || !ctx->has_text_encoder || !d_imgs || !d_ids || !h_offsets) {
        fprintf(stderr, "clip_amd_encode_pair_device_multi: needs a two-tower clip_amd_model_load_multi context\n");
        return false;
    }
    if (n_images <= 0 || n_texts <= 0) { fprintf(stderr, "clip_amd_encode_pair_device_multi: needs images and texts\n"); return false; }
    if (ctx->vision_hparams.projection_dim != ctx->text_hparams.projection_dim) { fprintf(stderr, "clip_amd_encode_pair_device_multi: the towers project to different widths\n"); return false; }
    const int G = multi_device_count(ctx);
    std::vector<std::vector<int32_t>> off(G);          // per-shard offsets rebased to 0 (alive until multi_run_pair has synchronised)
    return multi_run_pair(ctx, n_images, n_texts, ctx->vision_hparams.projection_dim, vec_img, vec_txt, "clip_amd_encode_pair_device_multi",
                          [&](int g, clip_ctx * c, int l, int h, float * d_send) { return vision_forward_device(c, d_imgs[g], h - l, d_send, normalize); },
                          [&](int g, clip_ctx * c, int l, int h, float * d_send) {
                              off[g].resize((size_t)(h - l) + 1);
                              for (int i = l; i <= h; i++) off[g][(size_t)(i - l)] = h_offsets[i] - h_offsets[l];
                              return text_forward_device(c, d_ids[g], off[g].data(), h - l, d_send, normalize);
                          });
} catch (...) { fprintf(stderr, "clip_amd_encode_pair_device_multi: exception\n"); return false; }

// n ragged texts (host token lists) -> d_out [n][proj] on ctx's device, queued on ctx->stream.  ids / off: caller-owned staging that must
// stay alive until the stream is synchronised (the upload of pageable memory may still be reading it).
static bool texts_to_device(clip_ctx * ctx, const clip_tokens * tokens, size_t n_texts, std::vector<int32_t> & ids, std::vector<int32_t> & off,
                            float * d_out, bool normalize) {
    ids.clear();
    off.assign(n_texts + 1, 0);
    for (size_t i = 0; i < n_texts; i++) {
        if (!tokens[i].data || tokens[i].size == 0) { fprintf(stderr, "clip_text_encode: empty token list\n"); return false; }
        for (size_t j = 0; j < tokens[i].size; j++) {
            const int32_t id = tokens[i].data[j];
            if (id < 0 || id >= ctx->text_hparams.n_vocab) { fprintf(stderr, "clip_text_encode: token id %d out of range\n", id); return false; }
            ids.push_back(id);
        }
        off[i + 1] = (int32_t)ids.size();
    }
    (void)hipSetDevice(ctx->device);
    // persistent device staging (ids in io_in): no hipMalloc / hipFree on the per-call path
    if (!ensure_io(ctx, ids.size() * 4, 16)) { fprintf(stderr, "clip_text_encode: out of device memory\n"); return false; }
    bool ok = hipMemcpyAsync(ctx->io_in, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    return ok && text_forward_device(ctx, (const int32_t *)ctx->io_in, off.data(), (int)n_texts, d_out, normalize);
}

bool clip_text_batch_encode(const struct clip_ctx * cctx, const int n_threads, const struct clip_tokens * tokens, size_t n_texts, float * vec,
                            const bool normalize) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    (void)n_threads;
    clip_ctx * ctx = const_cast<clip_ctx *>(cctx);
    if (!ctx->has_text_encoder) {
        printf("This GGUF file seems to have no text encoder\n");
        return false;
    }
    if (ctx->device < 0) {
        fprintf(stderr, "clip_text_encode: no HIP device bound to this context — the encoders have no CPU fallback\n");
        return false;
    }
    if (n_texts == 0) return true;
    const int proj = ctx->text_hparams.projection_dim;
    bool ok;
    if (ctx->multi && n_texts >= 2 * (size_t)multi_device_count(ctx)) {
        // sharded over the devices of a clip_amd_model_load_multi context: contiguous runs of texts per replica, ONE all-gather
        const int G = multi_device_count(ctx);
        std::vector<std::vector<int32_t>> ids(G), off(G);
        ok = multi_run(ctx, (int)n_texts, proj, vec, "clip_text_batch_encode",
                       [&](int g, clip_ctx * c, int l, int h, float * d_send) { return texts_to_device(c, tokens + l, (size_t)(h - l), ids[g], off[g], d_send, normalize); });
    } else {
        std::vector<int32_t> ids, off;
        (void)hipSetDevice(ctx->device);
        ok = ensure_io(ctx, 16, n_texts * (size_t)proj * 4);
        ok = ok && texts_to_device(ctx, tokens, n_texts, ids, off, (float *)ctx->io_out, normalize);
        ok = ok && hipMemcpyAsync(vec, ctx->io_out, n_texts * (size_t)proj * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (ctx->profiling) prof_collect(ctx);
    return ok;
} catch (const std::exception & e) { fprintf(stderr, "clip_text_batch_encode: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_text_batch_encode: unknown exception\n"); return false; }

bool clip_text_encode(const struct clip_ctx * ctx, const int n_threads, const struct clip_tokens * tokens, float * vec, const bool normalize) {
    if (!ctx->has_text_encoder) {
        printf("This GGUF file seems to have no text encoder\n");
        return false;
    }
    return clip_text_batch_encode(ctx, n_threads, tokens, 1, vec, normalize);
}

bool clip_amd_text_batch_encode_device(struct clip_ctx * ctx, const int32_t * d_ids, const int32_t * h_offsets, int n_texts, float * d_out,
                                       bool normalize) {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    return text_forward_device(ctx, d_ids, h_offsets, n_texts, d_out, normalize);
}

// ---- scoring (host, exact reference semantics) ----
float clip_similarity_score(const float * vec1, const float * vec2, const int vec_dim) {
    float dot = 0.0f;   // sequential f32 accumulation, reference clip.cpp:1525-1532
    for (int i = 0; i < vec_dim; i++) dot += vec1[i] * vec2[i];
    return dot;
}

bool clip_compare_text_and_image(const struct clip_ctx * ctx, const int n_threads, const char * text, const struct clip_image_u8 * image,
                                 float * score) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!(ctx->has_text_encoder && ctx->has_vision_encoder)) {
        printf("clip_compare_text_and_image function can only be used with two-tower models\n");
        return false;
    }
    const int dim = ctx->vision_hparams.projection_dim;
    std::vector<float> img_vec(dim), txt_vec(dim);
    std::vector<int32_t> ids;
    if (!tokenize_text(ctx, text, ids)) return false;
    clip_tokens tk{ids.data(), ids.size()};
    if (!clip_text_encode(ctx, n_threads, &tk, txt_vec.data(), true)) return false;
    clip_image_f32 res{};
    if (!preprocess_image(ctx, image, &res)) return false;
    const bool ok = clip_image_encode(ctx, n_threads, &res, img_vec.data(), true);
    clip_image_f32_clean(&res);   // the reference leaks this temporary
    if (!ok) return false;
    *score = clip_similarity_score(img_vec.data(), txt_vec.data(), dim);
    return true;
} catch (const std::exception & e) { fprintf(stderr, "clip_compare_text_and_image: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_compare_text_and_image: unknown exception\n"); return false; }

bool softmax_with_sorting(float * arr, const int length, float * sorted_scores, int * indices) {
    if (length < 0) return false;
    // exp(x) + 1e-9, no max-subtraction, sum in double (reference clip.cpp:1591-1622)
    double sum = 0.0;
    for (int i = 0; i < length; i++) {
        arr[i] = (float)(exp(arr[i]) + 1e-9);
        sum += arr[i];
    }
    std::vector<int> order(length);
    for (int i = 0; i < length; i++) {
        arr[i] = (float)(arr[i] / sum);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return arr[a] > arr[b]; });
    for (int i = 0; i < length; i++) {
        sorted_scores[i] = arr[order[i]];
        indices[i] = order[i];
    }
    return true;
}

bool clip_zero_shot_label_image(struct clip_ctx * ctx, const int n_threads, const struct clip_image_u8 * input_img, const char ** labels,
                                const size_t n_labels, float * scores, int * indices) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!(ctx->has_text_encoder && ctx->has_vision_encoder)) {
        printf("clip_zero_shot_label_image function can only be used with two-tower models\n");
        return false;
    }
    const int dim = ctx->vision_hparams.projection_dim;
    clip_image_f32 res{};
    if (!preprocess_image(ctx, input_img, &res)) return false;
    std::vector<float> img_vec(dim);
    const bool iok = clip_image_encode(ctx, n_threads, &res, img_vec.data(), false);   // un-normalised, as the reference (:1639)
    clip_image_f32_clean(&res);
    if (!iok) return false;
    // all labels in ONE batched text pass (the reference encodes them one by one, :1647-1653)
    std::vector<std::vector<int32_t>> ids(n_labels);
    std::vector<clip_tokens> toks(n_labels);
    for (size_t i = 0; i < n_labels; i++) {
        if (!tokenize_text(ctx, labels[i], ids[i])) return false;
        toks[i] = clip_tokens{ids[i].data(), ids[i].size()};
    }
    std::vector<float> txt((size_t)n_labels * dim), sims(n_labels);
    if (!clip_text_batch_encode(ctx, n_threads, toks.data(), n_labels, txt.data(), false)) return false;
    for (size_t i = 0; i < n_labels; i++) sims[i] = clip_similarity_score(img_vec.data(), txt.data() + i * dim, dim);
    return softmax_with_sorting(sims.data(), (int)n_labels, scores, indices);
} catch (const std::exception & e) { fprintf(stderr, "clip_zero_shot_label_image: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_zero_shot_label_image: unknown exception\n"); return false; }

// ---- batched zero-shot on the GPU: preprocessing, both towers and the scoring stay on the device ----
bool clip_amd_zero_shot_score_device(struct clip_ctx * ctx, const float * d_img, int n_images, const float * d_txt, int n_labels, int dim,
                                     float * d_scores, int * d_indices) {
    if (!ctx || ctx->device < 0) {
        fprintf(stderr, "clip_amd_zero_shot_score_device: no HIP device bound to this context\n");
        return false;
    }
    if (!launch_zero_shot(d_img, n_images, d_txt, n_labels, dim, d_scores, d_indices, ctx->stream)) {
        fprintf(stderr, "clip_amd_zero_shot_score_device: unsupported size (n_labels %d > 8192 or dim %d > 4096)\n", n_labels, dim);
        return false;
    }
    return hipGetLastError() == hipSuccess;
}

bool clip_amd_zero_shot_label_images(struct clip_ctx * ctx, const struct clip_image_u8 * imgs, int n_images, const char ** labels,
                                     size_t n_labels, float * scores, int * indices) try {
    RelaxCapture relax_capture;     // (model.h: this thread may allocate while another thread of the process captures)
    if (!(ctx->has_text_encoder && ctx->has_vision_encoder)) {
        printf("clip_zero_shot_label_image function can only be used with two-tower models\n");
        return false;
    }
    if (ctx->device < 0) {
        fprintf(stderr, "clip_amd_zero_shot_label_images: no HIP device bound to this context — no CPU fallback\n");
        return false;
    }
    if (n_images <= 0 || n_labels == 0) return true;
    const int dim = ctx->vision_hparams.projection_dim, S = ctx->vision_hparams.image_size;
    if (ctx->text_hparams.projection_dim != dim) return false;
    std::vector<int32_t> ids, off(n_labels + 1, 0);
    std::vector<int32_t> one;
    for (size_t i = 0; i < n_labels; i++) {
        if (!tokenize_text(ctx, labels[i], one)) return false;
        ids.insert(ids.end(), one.begin(), one.end());
        off[i + 1] = (int32_t)ids.size();
    }
    (void)hipSetDevice(ctx->device);
    const int chunk = std::min(n_images, 256);
    DBuf d_ids(ids.size() * 4), d_txt(n_labels * (size_t)dim * 4), d_sc((size_t)chunk * n_labels * 4), d_ix((size_t)chunk * n_labels * 4);
    if (!d_ids.p || !d_txt.p || !d_sc.p || !d_ix.p || !ensure_io(ctx, (size_t)S * S * 12 * chunk, (size_t)dim * 4 * chunk)) {
        fprintf(stderr, "clip_amd_zero_shot_label_images: out of device memory\n");
        return false;
    }
    // labels once, un-normalised like the reference (clip.cpp:1639-1653) ...
    bool ok = hipMemcpyAsync(d_ids.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && text_forward_device(ctx, (const int32_t *)d_ids.p, off.data(), (int)n_labels, (float *)d_txt.p, false);
    // ... then the images in chunks: u8 -> preprocess -> vision tower -> scores, all on the device
    for (int b0 = 0; b0 < n_images && ok; b0 += chunk) {
        const int bc = std::min(chunk, n_images - b0);
        ok = ok && preprocess_batch_device(ctx, imgs + b0, bc, (float *)ctx->io_in);
        ok = ok && vision_forward_device(ctx, (const float *)ctx->io_in, bc, (float *)ctx->io_out, false);
        ok = ok && clip_amd_zero_shot_score_device(ctx, (const float *)ctx->io_out, bc, (const float *)d_txt.p, (int)n_labels, dim, (float *)d_sc.p, (int *)d_ix.p);
        ok = ok && hipMemcpyAsync(scores + (size_t)b0 * n_labels, d_sc.p, (size_t)bc * n_labels * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(indices + (size_t)b0 * n_labels, d_ix.p, (size_t)bc * n_labels * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (ctx->profiling) prof_collect(ctx);
    return ok;
} catch (const std::exception & e) { fprintf(stderr, "clip_amd_zero_shot_label_images: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_amd_zero_shot_label_images: unknown exception\n"); return false; }

// ---- quantizer (reference clip.cpp:1661-1844): re-emit the GGUF with 2-D "*weight" tensors quantised ----
bool clip_model_quantize(const char * fname_inp, const char * fname_out, const int itype) try {
    switch (itype) {
    case 2: case 3: case 6: case 7: case 8: break;
    default:
        fprintf(stderr, "%s: invalid quantization type %d\n", "clip_model_quantize", itype);
        return false;
    }
    GgufFile g;
    std::string err;
    if (!g.open(fname_inp, err)) { fprintf(stderr, "clip_model_quantize: %s\n", err.c_str()); return false; }
    std::vector<std::pair<std::string, GgufValue>> kv;
    bool have_qv = false;
    for (auto & e : g.kv) {
        if (e.first == "general.file_type") { kv.emplace_back(e.first, gguf_make_u32((uint32_t)itype)); continue; }
        if (e.first == "general.quantization_version") { kv.emplace_back(e.first, gguf_make_u32(2)); have_qv = true; continue; }
        kv.push_back(e);
    }
    if (!have_qv) kv.emplace_back("general.quantization_version", gguf_make_u32(2));   // GGML_QNT_VERSION
    std::vector<std::vector<uint8_t>> store(g.tensors.size());
    std::vector<GgufOutTensor> outs;
    size_t total_org = 0, total_new = 0;
    std::vector<float> f32buf;
    for (size_t i = 0; i < g.tensors.size(); i++) {
        const GgufTensorInfo & t = g.tensors[i];
        const std::string & name = t.name;
        // regex ".*weight" (full match) and n_dims == 2   (reference clip.cpp:1711-1739)
        bool quantize = name.size() >= 6 && name.compare(name.size() - 6, 6, "weight") == 0 && t.n_dims == 2;
        GgufOutTensor o;
        o.name = name;
        o.n_dims = t.n_dims;
        for (int d = 0; d < 4; d++) o.ne[d] = t.ne[d];
        if (quantize) {
            if (t.type != GT_F32 && t.type != GT_F16) { printf("Please use an input file in f32 or f16\n"); return false; }
            if (t.ne[0] % 32) quantize = false;
        }
        if (quantize) {
            const int64_t n = t.ne[0] * t.nrows();
            f32buf.resize((size_t)n);
            dequantize_row(t.type, t.data, f32buf.data(), n);
            store[i].resize(ggml_row_bytes(itype, t.ne[0]) * (size_t)t.nrows());
            quantize_rows(itype, f32buf.data(), store[i].data(), t.nrows(), t.ne[0]);
            o.type = itype;
            o.data = store[i].data();
            o.nbytes = store[i].size();
        } else {
            o.type = t.type;
            o.data = t.data;
            o.nbytes = t.nbytes;
        }
        total_org += t.nbytes;
        total_new += o.nbytes;
        printf("%s: n_dims = %d | quantize=%d | size = %f MB -> %f MB\n", name.c_str(), t.n_dims, (int)quantize, t.nbytes / 1024.0 / 1024.0,
               o.nbytes / 1024.0 / 1024.0);
        outs.push_back(o);
    }
    if (!gguf_write(fname_out, g.version, g.alignment, kv, outs, err)) { fprintf(stderr, "clip_model_quantize: %s\n", err.c_str()); return false; }
    printf("%s: original size  = %8.2f MB\n", "clip_model_quantize", total_org / 1024.0 / 1024.0);
    printf("%s: quantized size  = %8.2f MB\n", "clip_model_quantize", total_new / 1024.0 / 1024.0);
    return true;
} catch (const std::exception & e) { fprintf(stderr, "clip_model_quantize: %s\n", e.what()); return false; } catch (...) { fprintf(stderr, "clip_model_quantize: unknown exception\n"); return false; }

// ---- profiling ----
void clip_amd_profile_enable(struct clip_ctx * ctx, bool on) {
    if (!ctx || ctx->device < 0) return;
    if (!on) prof_collect(ctx);
    ctx->profiling = on;
}

// Copies a textual report "tag launches total_ms flops bytes\n" into buf; returns bytes needed.
int clip_amd_profile_report(struct clip_ctx * ctx, char * buf, int cap, bool reset) {
    if (!ctx) return 0;
    if (ctx->device >= 0) prof_collect(ctx);
    std::string s;
    char line[256];
    for (auto & e : ctx->prof) {
        snprintf(line, sizeof line, "%s %lld %.6f %.6e %.6e\n", e.first.c_str(), (long long)e.second.launches, e.second.ms, e.second.flops,
                 e.second.bytes);
        s += line;
    }
    if (buf && cap > 0) {
        const size_t n = std::min((size_t)cap - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    if (reset) ctx->prof.clear();
    return (int)s.size() + 1;
}

int clip_amd_profile_read(struct clip_ctx * ctx, float * ms, int64_t * launches, int cap, bool reset) {
    if (!ctx) return 0;
    if (ctx->device >= 0) prof_collect(ctx);
    static const char * fam[4] = {"gemm", "attention", "layernorm", ""};
    float m[4] = {0, 0, 0, 0};
    int64_t l[4] = {0, 0, 0, 0};
    for (auto & e : ctx->prof) {
        int f = 3;
        for (int i = 0; i < 3; i++)
            if (e.first.compare(0, strlen(fam[i]), fam[i]) == 0) { f = i; break; }
        m[f] += (float)e.second.ms;
        l[f] += e.second.launches;
    }
    const int n = std::min(cap, 4);
    for (int i = 0; i < n; i++) { ms[i] = m[i]; launches[i] = l[i]; }
    if (reset) ctx->prof.clear();
    return n;
}

// ---- kernel-level test hooks (compiled out with -DCLIPAMD_TEST_HOOKS=0: `make hooks=0`) ----
#ifndef CLIPAMD_TEST_HOOKS
#define CLIPAMD_TEST_HOOKS 1
#endif
#if CLIPAMD_TEST_HOOKS
int clip_amd_test_gemm_tile(int64_t M, int64_t N, int64_t K, int quantised) {
    const int Kpad = (int)((K + 63) / 64 * 64);
    return gemm_tile_for((int)M, (int)N, Kpad, quantised != 0);
}
int clip_amd_test_gemm_tile_ex(int64_t M, int64_t N, int64_t K, int quantised, int shared_device) {
    const int Kpad = (int)((K + 63) / 64 * 64);
    return gemm_tile_for((int)M, (int)N, Kpad, quantised != 0, shared_device != 0);
}

// Extended form: qcols / qscale (EPI_F16 Q-scale path, clip.cpp:1363) and epilogue 5 = EPI_PATCH_F32 (patch embedding:
// row m of the GEMM lands in output row (m / Np) * T + 1 + m % Np and gets pos[1 + m % Np] added; no bias; the class-token
// rows (b * T) are left untouched).  For epilogue 5, y is [(M / Np) * T][N] and pos is [T][N].
int clip_amd_test_gemm_ex(int type, const void * w_raw, int64_t N, int64_t K, const float * x, int64_t M, const float * bias, const float * resid,
                          float * y, int epilogue, int tile, int qcols, float qscale, int Np, int T, const float * pos) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); fprintf(stderr, "clip_amd_test_gemm: no HIP device\n"); return -1; }
    if (epilogue == 5 && (Np <= 0 || T < Np + 1 || M % Np || !pos)) return -3;
    // the weight goes through the production repack path (load.cpp)
    DevWeight W;
    void * wbase = nullptr;
    if (!repack_for_test(type, w_raw, N, K, W, &wbase)) return -2;
    const int Kpad = W.Kpad;
    const size_t out_rows = epilogue == 5 ? (size_t)(M / Np) * T : (size_t)M;
    DBuf dx32((size_t)M * K * 4), dx16((size_t)(M + 1) * Kpad * 2), dbias((size_t)N * 4), dres((size_t)M * N * 4), dout(out_rows * N * 4), dout32(out_rows * N * 4),
        dpos(epilogue == 5 ? (size_t)T * N * 4 : 16);
    int rc = 0;
    hipStream_t s = nullptr;
    (void)hipMemcpy(dx32.p, x, (size_t)M * K * 4, hipMemcpyHostToDevice);
    if (bias) (void)hipMemcpy(dbias.p, bias, (size_t)N * 4, hipMemcpyHostToDevice);
    launch_f32_to_f16((const float *)dx32.p, (int)K, (half_t *)dx16.p, Kpad, (int)M, (int)K, Kpad, s);
    GemmParams p;
    p.A = (const half_t *)dx16.p; p.lda = Kpad; p.M = (int)M; p.W = W; p.bias = bias ? (const float *)dbias.p : nullptr; p.ldc = (int)N;
    DBuf skw((size_t)64 << 20), skc(4096 * 4);   // split-K workspace + ticket counters (as load.cpp gives the ctx)
    (void)hipMemset(skc.p, 0, 4096 * 4);
    p.sk_ws = (float *)skw.p; p.sk_ws_floats = (size_t)16 << 20; p.sk_cnt = (unsigned *)skc.p; p.sk_cnt_n = 4096;
    DBuf panel((size_t)W.Npad * W.Kpad * 2);         // fp16 panel for the 8-wave large-M kernel (quantised weights)
    p.w16_scratch = (half_t *)panel.p; p.w16_scratch_halfs = panel.p ? (size_t)W.Npad * W.Kpad : 0;
    int epi = EPI_F32;
    switch (epilogue) {
    case 0: epi = EPI_F32; p.out = dout32.p; break;
    case 1: epi = EPI_F16; p.out = dout.p; p.qscale = qscale; p.qcols = qcols; break;
    case 2: epi = EPI_GELU_F16; p.out = dout.p; break;
    case 3: epi = EPI_QGELU_F16; p.out = dout.p; break;
    case 4:
        epi = EPI_RESID_F32;
        (void)hipMemcpy(dout32.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice);
        p.out = dout32.p;
        p.resid = (const float *)dout32.p;
        break;
    case 5:
        epi = EPI_PATCH_F32;
        (void)hipMemcpy(dpos.p, pos, (size_t)T * N * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dout32.p, y, out_rows * N * 4, hipMemcpyHostToDevice);   // caller's fill: rows the epilogue must not touch
        p.out = dout32.p; p.Np = Np; p.T = T; p.pos = (const float *)dpos.p; p.bias = nullptr;
        break;
    default: rc = -3;
    }
    if (rc == 0) {
        launch_gemm(p, epi, tile, s);
        if (epi == EPI_F16 || epi == EPI_GELU_F16 || epi == EPI_QGELU_F16)
            launch_f16_to_f32((const half_t *)dout.p, (int)N, (float *)dout32.p, (int)N, (int)M, (int)N, s);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = -4;
        else (void)hipMemcpy(y, dout32.p, out_rows * N * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(wbase);
    return rc;
}

// LayerNorm fold A/B (gemm_common.h): residual GEMM -> LayerNorm -> consumer GEMM, as three launches (fold = 0) or as two with the
// LayerNorm folded into the epilogues (fold = 1).  See include/clip_amd.h.
int clip_amd_test_lnfold(int type, const void * w1_raw, int64_t h, int64_t K1, const void * w2_raw, int64_t N2, const float * a, int64_t M,
                         const float * b1, const float * resid, const float * gamma, const float * beta, float eps, const float * b2,
                         int epi2, int tile1, int tile2, int fold, int qcols, float qscale, float * x1_out, float * y_out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); fprintf(stderr, "clip_amd_test_lnfold: no HIP device\n"); return -1; }
    if (h % 64 || epi2 < 1 || epi2 > 3) return -3;
    DevWeight W1, W2;
    void * w1base = nullptr, * w2base = nullptr;
    if (!repack_for_test(type, w1_raw, h, K1, W1, &w1base)) return -2;
    if (!repack_for_test(type, w2_raw, N2, h, W2, &w2base)) { (void)hipFree(w1base); return -2; }
    const int st_stride = (int)((M + 63) & ~(int64_t)63);
    DBuf da32((size_t)M * K1 * 4), da16((size_t)(M + 1) * W1.Kpad * 2), dx((size_t)M * h * 4), dxn((size_t)(M + 1) * W2.Kpad * 2), dy16((size_t)M * N2 * 2),
        dy32((size_t)M * N2 * 4), db1((size_t)h * 4), db2((size_t)N2 * 4), dg((size_t)h * 4), dbeta((size_t)h * 4), dc((size_t)N2 * 4), dbf((size_t)N2 * 4),
        dstats((size_t)(h / 16) * st_stride * 8);
    hipStream_t s = nullptr;
    (void)hipMemcpy(da32.p, a, (size_t)M * K1 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx.p, resid, (size_t)M * h * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db1.p, b1, (size_t)h * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db2.p, b2, (size_t)N2 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dg.p, gamma, (size_t)h * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dbeta.p, beta, (size_t)h * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dxn.p, 0, (size_t)(M + 1) * W2.Kpad * 2);
    // fold == 2: the centred form (GemmParams::xg_mu / ln_mu): the offsets are the row means of the INCOMING residual rows — what the
    // previous LayerNorm's consumer leaves in the layer chain — computed here in double
    DBuf dmu((size_t)M * 4), dmu2((size_t)M * 4);
    if (fold == 2) {
        std::vector<float> mu((size_t)M);
        for (int64_t m = 0; m < M; m++) {
            double sacc = 0;
            for (int64_t k = 0; k < h; k++) sacc += resid[m * h + k];
            mu[(size_t)m] = (float)(sacc / (double)h);
        }
        (void)hipMemcpy(dmu.p, mu.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    }
    launch_f32_to_f16((const float *)da32.p, (int)K1, (half_t *)da16.p, W1.Kpad, (int)M, (int)K1, W1.Kpad, s);
    DBuf skw((size_t)64 << 20), skc(4096 * 4);
    (void)hipMemset(skc.p, 0, 4096 * 4);
    DBuf panel1((size_t)W1.Npad * W1.Kpad * 2), panel2((size_t)W2.Npad * W2.Kpad * 2);
    auto common = [&](GemmParams & p, const DBuf & panel, const DevWeight & W) {
        p.sk_ws = (float *)skw.p; p.sk_ws_floats = (size_t)16 << 20; p.sk_cnt = (unsigned *)skc.p; p.sk_cnt_n = 4096;
        p.w16_scratch = (half_t *)panel.p; p.w16_scratch_halfs = panel.p ? (size_t)W.Npad * W.Kpad : 0;
    };
    GemmParams p1;
    p1.A = (const half_t *)da16.p; p1.lda = W1.Kpad; p1.M = (int)M; p1.W = W1; p1.bias = (const float *)db1.p; p1.out = dx.p; p1.ldc = (int)h;
    p1.resid = (const float *)dx.p;
    common(p1, panel1, W1);
    GemmParams p2;
    p2.A = (const half_t *)dxn.p; p2.lda = W2.Kpad; p2.M = (int)M; p2.W = W2; p2.out = dy16.p; p2.ldc = (int)N2; p2.qcols = qcols; p2.qscale = qscale;
    common(p2, panel2, W2);
    const int epi = epi2 == 1 ? EPI_F16 : epi2 == 2 ? EPI_GELU_F16 : EPI_QGELU_F16;
    if (tile1 < 0) {
        // the small-M kernels (k_skinny.hip; M <= 64 in the layers, the hook allows up to 128): folded form, or LayerNorm fused on the operand
        if (M > 128 || h > 2048) { (void)hipFree(w1base); (void)hipFree(w2base); return -5; }
        DBuf dskst((size_t)2 * SKINNY_MAX_ROWS * 128 * sizeof(float2));
        SkinnyParams a1;
        a1.A16 = (const half_t *)da16.p; a1.lda = W1.Kpad; a1.M = (int)M; a1.W = W1; a1.bias = (const float *)db1.p; a1.out = dx.p; a1.ldc = (int)h; a1.resid = (const float *)dx.p;
        SkinnyParams a2;
        a2.M = (int)M; a2.W = W2; a2.out = dy16.p; a2.ldc = (int)N2; a2.qcols = qcols; a2.qscale = qscale; a2.eps = eps;
        if (fold) {
            launch_fold_vectors(W2, (const float *)dg.p, (const float *)dbeta.p, (const float *)db2.p, (float *)dc.p, (float *)dbf.p, s);
            a1.xg_out = (half_t *)dxn.p; a1.ldxg = W2.Kpad; a1.xg_gamma = (const float *)dg.p; a1.fstats_out = (float2 *)dstats.p; a1.fstride_out = st_stride;
            a2.A16 = (const half_t *)dxn.p; a2.lda = W2.Kpad; a2.bias = (const float *)dbf.p; a2.ln_c = (const float *)dc.p;
            a2.fstats = (const float2 *)dstats.p; a2.fslots = (int)h / 16; a2.fslotw = 16; a2.fstride = st_stride;
            if (fold == 2) { a1.xg_mu = (const float *)dmu.p; a2.ln_mu = (const float *)dmu.p; a2.mu_out = (float *)dmu2.p; }
        } else {
            a1.stats_out = (float2 *)dskst.p;
            a2.x32 = (const float *)dx.p; a2.ldx = (int)h; a2.ln_w = (const float *)dg.p; a2.ln_b = (const float *)dbeta.p; a2.stats_in = (const float2 *)dskst.p;
            a2.stats_slots = (int)h / 16; a2.bias = (const float *)db2.p;
        }
        int rc2 = 0;
        if (!skinny_supported(a1, EPI_RESID_F32) || !skinny_supported(a2, epi)) rc2 = -5;
        else {
            launch_skinny(a1, EPI_RESID_F32, s);
            launch_skinny(a2, epi, s);
            launch_f16_to_f32((const half_t *)dy16.p, (int)N2, (float *)dy32.p, (int)N2, (int)M, (int)N2, s);
            if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc2 = -4;
            else {
                (void)hipMemcpy(x1_out, dx.p, (size_t)M * h * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(y_out, dy32.p, (size_t)M * N2 * 4, hipMemcpyDeviceToHost);
            }
        }
        (void)hipFree(w1base);
        (void)hipFree(w2base);
        return rc2;
    }
    if (fold) {
        launch_fold_vectors(W2, (const float *)dg.p, (const float *)dbeta.p, (const float *)db2.p, (float *)dc.p, (float *)dbf.p, s);
        p1.xg_out = (half_t *)dxn.p; p1.ldxg = W2.Kpad; p1.xg_gamma = (const float *)dg.p; p1.stats_out = (float2 *)dstats.p; p1.stats_stride = st_stride;
        int t1 = tile1 ? tile1 : gemm_tile_for((int)M, (int)h, W1.Kpad, W1.wtype != W_F16);
        const int slotw = gemm_fold_slotw(t1);
        if (fold == 2) p1.xg_mu = (const float *)dmu.p;
        launch_gemm(p1, EPI_RESID_F32, tile1, s);
        p2.bias = (const float *)dbf.p; p2.ln_c = (const float *)dc.p; p2.ln_stats = (const float2 *)dstats.p; p2.ln_slotw = slotw; p2.ln_slots = (int)h / slotw;
        p2.ln_stride = st_stride; p2.ln_eps = eps;
        if (fold == 2) { p1.xg_mu = (const float *)dmu.p; p2.ln_mu = (const float *)dmu.p; p2.mu_out = (float *)dmu2.p; }
        launch_gemm(p2, epi, tile2, s);
    } else {
        launch_gemm(p1, EPI_RESID_F32, tile1, s);
        launch_layernorm((const float *)dx.p, (int)h, nullptr, 1, (const float *)dg.p, (const float *)dbeta.p, eps, (int)M, (int)h, (half_t *)dxn.p, W2.Kpad, nullptr, 0, s);
        p2.bias = (const float *)db2.p;
        launch_gemm(p2, epi, tile2, s);
    }
    launch_f16_to_f32((const half_t *)dy16.p, (int)N2, (float *)dy32.p, (int)N2, (int)M, (int)N2, s);
    int rc = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = -4;
    else {
        (void)hipMemcpy(x1_out, dx.p, (size_t)M * h * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(y_out, dy32.p, (size_t)M * N2 * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(w1base);
    (void)hipFree(w2base);
    return rc;
}

int clip_amd_test_gemm(int type, const void * w_raw, int64_t N, int64_t K, const float * x, int64_t M, const float * bias, const float * resid,
                       float * y, int epilogue, int tile) {
    return clip_amd_test_gemm_ex(type, w_raw, N, K, x, M, bias, resid, y, epilogue, tile, 0, 1.0f, 0, 0, nullptr);
}


// Micro-benchmark of one GEMM shape through the production kernel (random weights of ggml type `type`,
// quantised with the product codecs).  Returns the average kernel time in microseconds (HIP events), < 0 on error.
float clip_amd_bench_gemm(int type, int64_t N, int64_t K, int64_t M, int epilogue, int tile, int iters) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return -1.f; }
    std::vector<float> w((size_t)N * K);
    uint32_t st = 12345u;
    for (auto & v : w) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 9) % 2001 - 1000) * 4e-5f; }
    std::vector<uint8_t> raw(ggml_row_bytes(type, K) * (size_t)N);
    if (raw.empty() || quantize_rows(type, w.data(), raw.data(), N, K) == 0) return -2.f;
    DevWeight W;
    void * wbase = nullptr;
    if (!repack_for_test(type, raw.data(), N, K, W, &wbase)) return -2.f;
    const int Kpad = W.Kpad;
    DBuf dx((size_t)(M + 1) * Kpad * 2), dbias((size_t)N * 4), dout((size_t)M * N * 4);
    std::vector<uint16_t> hx((size_t)M * Kpad);
    for (auto & v : hx) { st = st * 1664525u + 1013904223u; v = f32_to_f16_bits(((int)(st >> 9) % 2001 - 1000) * 1e-3f); }
    (void)hipMemcpy(dx.p, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dbias.p, 0, (size_t)N * 4);
    (void)hipMemset(dout.p, 0, (size_t)M * N * 4);
    GemmParams p;
    p.A = (const half_t *)dx.p; p.lda = Kpad; p.M = (int)M; p.W = W; p.bias = (const float *)dbias.p; p.ldc = (int)N; p.out = dout.p;
    p.resid = (const float *)dout.p; p.qcols = 0;
    DBuf skw((size_t)64 << 20), skc(4096 * 4);
    (void)hipMemset(skc.p, 0, 4096 * 4);
    p.sk_ws = (float *)skw.p; p.sk_ws_floats = (size_t)16 << 20; p.sk_cnt = (unsigned *)skc.p; p.sk_cnt_n = 4096;
    DBuf panel((size_t)W.Npad * W.Kpad * 2);
    const bool pre = (epilogue >> 16) & 1;             // bit 16: time the GEMM alone on an already dequantised panel (per-layer form)
    if (pre && W.wtype != W_F16 && W.wtype != W_F32 && panel.p) {
        const DevWeight * w = &W;
        half_t * o = (half_t *)panel.p;
        launch_dequant(&w, &o, 1, nullptr);
        p.w16_pre = (const half_t *)panel.p;
    } else {
        p.w16_scratch = (half_t *)panel.p; p.w16_scratch_halfs = panel.p ? (size_t)W.Npad * W.Kpad : 0;
    }
    p.debug = (epilogue >> 8) & 0xFF;   // ablation switches (tuning only)
    // bit 17: time the epilogue with the LayerNorm fold — consumer form for the fp16 epilogues (row statistics + c vector), producer form
    // (xg + statistics out) for the residual epilogue
    const bool fold = (epilogue >> 17) & 1;
    epilogue &= 0xFF;
    const int st_stride = (int)((M + 63) & ~(int64_t)63);
    DBuf dstats(fold ? (size_t)(N > K ? N : K) / 32 * st_stride * 8 + 64 : 16), dvec(fold ? (size_t)(N > K ? N : K) * 4 + 64 : 16), dxg(fold ? (size_t)M * N * 2 + 64 : 16);
    if (fold) {
        std::vector<float> hs((size_t)(N > K ? N : K) / 32 * st_stride * 2), hv((size_t)(N > K ? N : K), 1.0f);
        for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 0.5f * 64; hs[i + 1] = 60.f; }
        (void)hipMemcpy(dstats.p, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dvec.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
        if (epilogue == EPI_RESID_F32) {
            p.xg_out = (half_t *)dxg.p; p.ldxg = (int)N; p.xg_gamma = (const float *)dvec.p; p.stats_out = (float2 *)dstats.p; p.stats_stride = st_stride;
        } else if (epilogue == EPI_F16 || epilogue == EPI_GELU_F16 || epilogue == EPI_QGELU_F16) {
            p.ln_c = (const float *)dvec.p; p.ln_stats = (const float2 *)dstats.p; p.ln_slotw = 64; p.ln_slots = (int)K / 64; p.ln_stride = st_stride; p.ln_eps = 1e-5f;
        }
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    // GEMM_ROTATE=R: cycle through R separate copies of the weight (and of its panel), as the layers of a tower do — the
    // default re-multiplies one weight, which stays in L2 / Infinity Cache
    int rot = 1;
    { const char * e = getenv("GEMM_ROTATE"); if (e) rot = atoi(e); if (rot < 1) rot = 1; if (rot > 64) rot = 64; }
    std::vector<DevWeight> Ws(1, p.W);
    std::vector<const half_t *> pans(1, p.w16_pre);
    std::vector<void *> owned;
    for (int r = 1; r < rot; r++) {
        DevWeight Wr;
        void * base = nullptr;
        if (!repack_for_test(type, raw.data(), N, K, Wr, &base)) break;
        owned.push_back(base);
        const half_t * pr = nullptr;
        if (pre && Wr.wtype != W_F16 && Wr.wtype != W_F32) {
            void * pp = nullptr;
            if (hipMalloc(&pp, (size_t)Wr.Npad * Wr.Kpad * 2) != hipSuccess) break;
            owned.push_back(pp);
            const DevWeight * w = &Wr;
            half_t * o = (half_t *)pp;
            launch_dequant(&w, &o, 1, nullptr);
            pr = (const half_t *)pp;
        }
        Ws.push_back(Wr);
        pans.push_back(pr);
    }
    rot = (int)Ws.size();
    auto go = [&](int i) { p.W = Ws[i % rot]; if (pre) p.w16_pre = pans[i % rot]; launch_gemm(p, epilogue, tile, nullptr); };
    for (int i = 0; i < 3; i++) go(i);
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; i++) go(i);
    (void)hipEventRecord(b, nullptr);
    float ms = -1.f;
    if (hipEventSynchronize(b) == hipSuccess && hipGetLastError() == hipSuccess) (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    for (void * o : owned) (void)hipFree(o);
    if (getenv("CLIPAMD_G8_STAMPS")) {   // tuning builds (-DCLIPAMD_G8_TIMING): dump the per-workgroup phase stamps of the last launch
        const int nwg = 4096;
        std::vector<unsigned long long> st((size_t)nwg * 8);
        (void)hipMemcpy(st.data(), skw.p, st.size() * 8, hipMemcpyDeviceToHost);
        FILE * f = fopen(getenv("CLIPAMD_G8_STAMPS"), "a");
        if (f) {
            fprintf(f, "# N=%lld K=%lld M=%lld epi=%d tile=%d\n", (long long)N, (long long)K, (long long)M, epilogue, tile);
            for (int i = 0; i < nwg; i++) {
                const unsigned long long * s8 = &st[(size_t)i * 8];
                if (!s8[0] || !s8[3]) continue;
                fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu\n", i, s8[0], s8[1], s8[2], s8[3], s8[5], s8[4], s8[6]);
            }
            fclose(f);
        }
    }
    (void)hipFree(wbase);
    return ms < 0 ? -4.f : ms * 1000.f / iters;
}

// Small-M kernel (k_skinny.hip) on host data: y[M][N] = epilogue(A . W^T + bias), M <= SKINNY_MAX_ROWS.  ln_w != NULL: A = LayerNorm(x) fused in
// the kernel (row statistics via launch_row_stats, as forward.cpp does for the first layer); else A = fp16(x).
// epilogue: 0 f32, 1 f16 (+ qcols / qscale), 2 gelu, 3 quick-gelu, 4 residual (also returns the partial row statistics it leaves:
// stats_out [N/16][128][2], may be NULL).  Returns 0, or -5 when the combination is not covered by the skinny path.
int clip_amd_test_skinny(int type, const void * w_raw, int64_t N, int64_t K, const float * x, int64_t M, const float * bias, const float * resid,
                         const float * ln_w, const float * ln_b, float eps, float * y, int epilogue, int qcols, float qscale, float * stats_out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return -1; }
    DevWeight W;
    void * wbase = nullptr;
    if (!repack_for_test(type, w_raw, N, K, W, &wbase)) return -2;
    const int Kpad = W.Kpad;
    DBuf dx32((size_t)M * K * 4), dx16((size_t)(M + 1) * Kpad * 2), dbias((size_t)N * 4), dout((size_t)M * N * 4), dout32((size_t)M * N * 4), dlw((size_t)K * 4), dlb((size_t)K * 4),
        dst((size_t)2 * SKINNY_MAX_ROWS * 128 * 8);
    hipStream_t s = nullptr;
    (void)hipMemcpy(dx32.p, x, (size_t)M * K * 4, hipMemcpyHostToDevice);
    if (bias) (void)hipMemcpy(dbias.p, bias, (size_t)N * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dst.p, 0, (size_t)2 * SKINNY_MAX_ROWS * 128 * 8);
    SkinnyParams p;
    p.M = (int)M; p.W = W; p.bias = bias ? (const float *)dbias.p : nullptr; p.ldc = (int)N; p.qcols = qcols; p.qscale = qscale;
    if (ln_w) {
        (void)hipMemcpy(dlw.p, ln_w, (size_t)K * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dlb.p, ln_b, (size_t)K * 4, hipMemcpyHostToDevice);
        launch_row_stats((const float *)dx32.p, (int)K, (int)M, (int)K, (float2 *)dst.p, s);
        p.x32 = (const float *)dx32.p; p.ldx = (int)K; p.ln_w = (const float *)dlw.p; p.ln_b = (const float *)dlb.p; p.eps = eps;
        p.stats_in = (const float2 *)dst.p; p.stats_slots = 1;
    } else {
        launch_f32_to_f16((const float *)dx32.p, (int)K, (half_t *)dx16.p, Kpad, (int)M, (int)K, Kpad, s);
        p.A16 = (const half_t *)dx16.p; p.lda = Kpad;
    }
    int epi = EPI_F32;
    switch (epilogue) {
    case 0: epi = EPI_F32; p.out = dout32.p; break;
    case 1: epi = EPI_F16; p.out = dout.p; break;
    case 2: epi = EPI_GELU_F16; p.out = dout.p; break;
    case 3: epi = EPI_QGELU_F16; p.out = dout.p; break;
    case 4:
        epi = EPI_RESID_F32;
        (void)hipMemcpy(dout32.p, resid, (size_t)M * N * 4, hipMemcpyHostToDevice);
        p.out = dout32.p; p.resid = (const float *)dout32.p;
        p.stats_out = (float2 *)dst.p + (size_t)SKINNY_MAX_ROWS * 128;
        break;
    default: (void)hipFree(wbase); return -3;
    }
    int rc = 0;
    if (!skinny_supported(p, epi)) rc = -5;
    else {
        launch_skinny(p, epi, s);
        if (epi == EPI_F16 || epi == EPI_GELU_F16 || epi == EPI_QGELU_F16)
            launch_f16_to_f32((const half_t *)dout.p, (int)N, (float *)dout32.p, (int)N, (int)M, (int)N, s);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = -4;
        else {
            (void)hipMemcpy(y, dout32.p, (size_t)M * N * 4, hipMemcpyDeviceToHost);
            if (stats_out && epi == EPI_RESID_F32) (void)hipMemcpy(stats_out, (float2 *)dst.p + (size_t)SKINNY_MAX_ROWS * 128, (size_t)128 * 128 * 8, hipMemcpyDeviceToHost);   // (rows 0..127)
        }
    }
    (void)hipFree(wbase);
    return rc;
}

int clip_amd_test_layernorm(const float * x, const float * w, const float * b, float eps, int64_t rows, int64_t h, float * y, int out_f16) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return -1; }
    DBuf dx((size_t)rows * h * 4), dw((size_t)h * 4), db((size_t)h * 4), dy((size_t)rows * h * 4), dh_((size_t)rows * h * 2);
    (void)hipMemcpy(dx.p, x, (size_t)rows * h * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dw.p, w, (size_t)h * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db.p, b, (size_t)h * 4, hipMemcpyHostToDevice);
    if (out_f16) {
        launch_layernorm((const float *)dx.p, (int)h, nullptr, 1, (const float *)dw.p, (const float *)db.p, eps, (int)rows, (int)h, (half_t *)dh_.p, (int)h, nullptr, 0, nullptr);
        launch_f16_to_f32((const half_t *)dh_.p, (int)h, (float *)dy.p, (int)h, (int)rows, (int)h, nullptr);
    } else {
        launch_layernorm((const float *)dx.p, (int)h, nullptr, 1, (const float *)dw.p, (const float *)db.p, eps, (int)rows, (int)h, nullptr, 0, (float *)dy.p, (int)h, nullptr);
    }
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -4;
    (void)hipMemcpy(y, dy.p, (size_t)rows * h * 4, hipMemcpyDeviceToHost);
    return 0;
}

int clip_amd_test_attention(const float * qkv, int nseq, int T, int h, int n_head, int causal, float * out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return -1; }
    const size_t rows = (size_t)nseq * T;
    DBuf d32(rows * 3 * h * 4), d16(rows * 3 * h * 2), o16(rows * h * 2), o32(rows * h * 4);
    (void)hipMemcpy(d32.p, qkv, rows * 3 * h * 4, hipMemcpyHostToDevice);
    launch_f32_to_f16((const float *)d32.p, 3 * h, (half_t *)d16.p, 3 * h, (int)rows, 3 * h, 3 * h, nullptr);
    if (!launch_attention((const half_t *)d16.p, (half_t *)o16.p, nseq, T, nullptr, T, h, n_head, causal != 0, nullptr)) return -2;
    launch_f16_to_f32((const half_t *)o16.p, h, (float *)o32.p, h, (int)rows, h, nullptr);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -4;
    (void)hipMemcpy(out, o32.p, rows * h * 4, hipMemcpyDeviceToHost);
    return 0;
}

#endif  // CLIPAMD_TEST_HOOKS

}  // extern "C"
