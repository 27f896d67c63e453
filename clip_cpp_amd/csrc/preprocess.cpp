// preprocess.cpp — host image preprocessing (reference clip.cpp:728-1008).
//
// Resize the shorter side to image_size with a PIL-style separable, antialiased bicubic filter
// (a = -0.5, support 2 x scale, coefficients in double, clamp to [0,255] after each pass), centre-crop
// to image_size x image_size and normalise (v/255 - mean)/std.  Arithmetic order follows the
// reference so that results are bit-identical: accumulate in double over the taps in increasing
// source index, round to float, clamp.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "model.h"

namespace clipamd {

namespace {

inline double cubic_weight(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Taps {
    int ksize = 0;
    std::vector<double> w;    // [out][ksize]
    std::vector<int> first;   // first source index per output
    std::vector<int> count;   // number of taps per output
};

// Resample.c-style coefficient table for mapping in_size -> out_size samples (reference clip.cpp:743-794)
Taps make_taps(int in_size, int out_size) {
    Taps t;
    const float in0 = 0.0f, in1 = (float)in_size;
    double filterscale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.w.assign((size_t)out_size * t.ksize, 0.0);
    t.first.resize(out_size);
    t.count.resize(out_size);
    const double ss = 1.0 / filterscale;
    for (int o = 0; o < out_size; o++) {
        const double center = in0 + (o + 0.5) * (in1 - in0) / out_size;
        int lo = (int)(center - support + 0.5);
        if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5);
        if (hi > in_size) hi = in_size;
        const int n = hi - lo;
        double * k = &t.w[(size_t)o * t.ksize];
        double total = 0.0;
        for (int x = 0; x < n; x++) {
            k[x] = cubic_weight((x + lo - center + 0.5) * ss);
            total += k[x];
        }
        if (total != 0.0)
            for (int x = 0; x < n; x++) k[x] /= total;
        t.first[o] = lo;
        t.count[o] = n;
    }
    return t;
}

inline float clamp255(float v) { return std::min(std::max(v, 0.0f), 255.0f); }

}  // namespace

bool preprocess_image(const clip_ctx * ctx, const clip_image_u8 * img, clip_image_f32 * res) {
    if (!ctx->has_vision_encoder) {
        printf("This gguf file seems to have no vision encoder\n");
        return false;
    }
    const int nx = img->nx, ny = img->ny;
    const int S = ctx->vision_hparams.image_size;
    if (nx <= 0 || ny <= 0 || !img->data) return false;

    res->nx = S;
    res->ny = S;
    res->size = (size_t)3 * S * S;
    res->data = new float[res->size]();

    // shorter side -> S, aspect ratio kept (reference clip.cpp:819-821)
    const float scale = std::min((float)nx, (float)ny) / (float)S;
    const int rx = (int)(nx / scale + 0.5f);
    const int ry = (int)(ny / scale + 0.5f);

    const Taps th = make_taps(nx, rx), tv = make_taps(ny, ry);

    // horizontal pass: [ny][nx] u8 -> [ny][rx] f32
    std::vector<float> hbuf((size_t)3 * rx * ny);
    for (int y = 0; y < ny; y++) {
        const uint8_t * srow = img->data + (size_t)3 * y * nx;
        float * drow = &hbuf[(size_t)3 * y * rx];
        for (int o = 0; o < rx; o++) {
            const double * k = &th.w[(size_t)o * th.ksize];
            const int lo = th.first[o], n = th.count[o];
            for (int c = 0; c < 3; c++) {
                double acc = 0.0;
                for (int x = 0; x < n; x++) acc += (double)srow[3 * (lo + x) + c] * k[x];
                drow[3 * o + c] = clamp255((float)acc);
            }
        }
    }
    // vertical pass: [ny][rx] -> [ry][rx]
    std::vector<float> vbuf((size_t)3 * rx * ry);
    for (int o = 0; o < ry; o++) {
        const double * k = &tv.w[(size_t)o * tv.ksize];
        const int lo = tv.first[o], n = tv.count[o];
        float * drow = &vbuf[(size_t)3 * o * rx];
        for (int x = 0; x < rx; x++) {
            for (int c = 0; c < 3; c++) {
                double acc = 0.0;
                for (int y = 0; y < n; y++) acc += (double)hbuf[3 * ((size_t)(lo + y) * rx + x) + c] * k[y];
                drow[3 * x + c] = clamp255((float)acc);
            }
        }
    }
    // centre crop + normalise (reference clip.cpp:903-917)
    const int x0 = (rx - S) / 2, y0 = (ry - S) / 2;
    const float * mean = ctx->image_mean;
    const float * stdv = ctx->image_std;
    for (int y = 0; y < S; y++) {
        for (int x = 0; x < S; x++) {
            const float * s = &vbuf[3 * ((size_t)(y + y0) * rx + (x + x0))];
            float * d = res->data + 3 * ((size_t)y * S + x);
            for (int c = 0; c < 3; c++) d[c] = ((s[c] / 255.0f) - mean[c]) / stdv[c];
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Device path: the same resize/crop/normalise on the GPU (k_preproc.hip).  The host computes the tap
// tables (one per distinct (source size, target size) pair), packs {descriptors, tables, raw u8 pixels} into one pinned
// blob, ships it with a single H2D copy and launches the two passes.  d_out: [n][S][S][3] f32 in HBM.
// ---------------------------------------------------------------------------------------------
void free_preprocess_slots(clip_ctx * ctx) {
    for (auto & sl : ctx->pre_slot) {
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        sl = clip_ctx::PreSlot();
    }
    if (ctx->pre_copy_stream) (void)hipStreamDestroy(ctx->pre_copy_stream);
    ctx->pre_copy_stream = nullptr;
}

// slot < 0: one staging blob, the stream is synchronised first (the blob of the previous call may still be in use), H2D on ctx->stream.
// slot >= 0 (calls of more than one staging piece, api.cpp encode_u8_to_device): a ring of PRE_SLOTS staging slots — the host fills pinned slot s
// while the GPU still works on the pieces in the other slots, the copy travels on a copy stream (under the previous chunk's forward
// pass), and the only waits are events (this slot's previous H2D before the host overwrites the pinned blob; its previous kernels before
// the copy overwrites the device blob).  d_out regions of one buffer: the kernels writing them are stream-ordered behind the forward reading it.
bool preprocess_batch_device(clip_ctx * ctx, const clip_image_u8 * imgs, int n, float * d_out, int slot) {
    if (!ctx->has_vision_encoder) {
        printf("This gguf file seems to have no vision encoder\n");
        return false;
    }
    if (ctx->device < 0) {
        fprintf(stderr, "clip_amd_image_batch_preprocess_device: no HIP device bound to this context\n");
        return false;
    }
    if (n <= 0) return true;
    const int S = ctx->vision_hparams.image_size;
    std::vector<PreImg> desc(n);
    std::vector<PreTaps> dir;
    std::vector<double> wpool;
    std::vector<int> ipool;
    std::map<std::pair<int, int>, int> seen;
    std::vector<Taps> tabs;
    auto table = [&](int in_size, int out_size) {
        auto it = seen.find({in_size, out_size});
        if (it != seen.end()) return it->second;
        Taps t = make_taps(in_size, out_size);
        PreTaps e;
        e.w_off = (long long)wpool.size();
        e.first_off = (int)ipool.size();
        e.count_off = e.first_off + out_size;
        e.ksize = t.ksize;
        wpool.insert(wpool.end(), t.w.begin(), t.w.end());
        ipool.insert(ipool.end(), t.first.begin(), t.first.end());
        ipool.insert(ipool.end(), t.count.begin(), t.count.end());
        dir.push_back(e);
        tabs.push_back(std::move(t));
        const int idx = (int)dir.size() - 1;
        seen[{in_size, out_size}] = idx;
        return idx;
    };
    size_t raw_bytes = 0, hbuf_floats = 0;
    int max_rows = 0;
    for (int i = 0; i < n; i++) {
        const clip_image_u8 & im = imgs[i];
        if (im.nx <= 0 || im.ny <= 0 || !im.data) {
            fprintf(stderr, "clip_amd_image_batch_preprocess_device: image %d is empty\n", i);
            return false;
        }
        const float scale = std::min((float)im.nx, (float)im.ny) / (float)S;   // same arithmetic as preprocess_image above
        const int rx = (int)(im.nx / scale + 0.5f), ry = (int)(im.ny / scale + 0.5f);
        if (rx < S || ry < S) {
            fprintf(stderr, "clip_amd_image_batch_preprocess_device: image %d (%dx%d) resizes below %d\n", i, im.nx, im.ny, S);
            return false;
        }
        PreImg & d = desc[i];
        d.nx = im.nx; d.ny = im.ny;
        d.x0 = (rx - S) / 2; d.y0 = (ry - S) / 2;
        d.th = table(im.nx, rx);
        d.tv = table(im.ny, ry);
        const Taps & tv = tabs[d.tv];
        d.ylo = tv.first[d.y0];
        int yhi = 0;
        for (int o = d.y0; o < d.y0 + S; o++) yhi = std::max(yhi, tv.first[o] + tv.count[o]);
        d.nrows = yhi - d.ylo;
        max_rows = std::max(max_rows, d.nrows);
        d.src_off = (long long)raw_bytes;
        raw_bytes += ((size_t)3 * im.nx * im.ny + 15) & ~(size_t)15;
        d.hbuf_off = (long long)hbuf_floats;
        hbuf_floats += (size_t)d.nrows * S * 3;
    }
    // blob layout (all 16-byte aligned): descriptors | directory | ipool | wpool | raw pixels ; then (device only) hbuf
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_desc = 0, o_dir = up(o_desc + desc.size() * sizeof(PreImg)), o_ip = up(o_dir + dir.size() * sizeof(PreTaps));
    const size_t o_wp = up(o_ip + ipool.size() * sizeof(int)), o_raw = up(o_wp + wpool.size() * sizeof(double));
    const size_t host_bytes = up(o_raw + raw_bytes), total = host_bytes + hbuf_floats * sizeof(float);
    (void)hipSetDevice(ctx->device);
    uint8_t * h = nullptr, * dv = nullptr;
    clip_ctx::PreSlot * sl = slot >= 0 ? &ctx->pre_slot[slot % clip_ctx::PRE_SLOTS] : nullptr;
    if (!sl) {
        (void)hipStreamSynchronize(ctx->stream);   // the pinned blob / device buffer of the previous call may still be in use
        if (!ensure_pinned(ctx, host_bytes)) { fprintf(stderr, "clip (hip): cannot pin %zu MB\n", host_bytes >> 20); return false; }
        if (ctx->pre_bytes < total) {
            if (ctx->pre_buf) (void)hipFree(ctx->pre_buf);
            ctx->pre_buf = nullptr; ctx->pre_bytes = 0;
            if (hipMalloc(&ctx->pre_buf, total) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "clip (hip): cannot allocate %zu MB for preprocessing\n", total >> 20); return false; }
            ctx->pre_bytes = total;
        }
        h = (uint8_t *)ctx->pinned;
        dv = (uint8_t *)ctx->pre_buf;
    } else {
        if (!ctx->pre_copy_stream && hipStreamCreateWithFlags(&ctx->pre_copy_stream, hipStreamNonBlocking) != hipSuccess) return false;
        if (!sl->ev_h2d && (hipEventCreateWithFlags(&sl->ev_h2d, hipEventDisableTiming) != hipSuccess ||
                            hipEventCreateWithFlags(&sl->ev_done, hipEventDisableTiming) != hipSuccess)) return false;
        if (sl->pin_bytes < host_bytes || sl->dev_bytes < total) {       // (re)allocation: nothing may still use the old buffers
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->pre_copy_stream);
            if (sl->pin_bytes < host_bytes) {
                if (sl->pin) (void)hipHostFree(sl->pin);
                sl->pin = nullptr; sl->pin_bytes = 0;
                const size_t want = host_bytes + host_bytes / 8;
                if (hipHostMalloc(&sl->pin, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "clip (hip): cannot pin %zu MB\n", want >> 20); return false; }
                sl->pin_bytes = want;
            }
            if (sl->dev_bytes < total) {
                if (sl->dev) (void)hipFree(sl->dev);
                sl->dev = nullptr; sl->dev_bytes = 0;
                const size_t want = total + total / 8;
                if (hipMalloc(&sl->dev, want) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "clip (hip): cannot allocate %zu MB for preprocessing\n", want >> 20); return false; }
                sl->dev_bytes = want;
            }
            sl->used = false;
        }
        if (sl->used && hipEventSynchronize(sl->ev_h2d) != hipSuccess) return false;     // the previous copy out of this pinned blob has finished
        h = (uint8_t *)sl->pin;
        dv = (uint8_t *)sl->dev;
    }
    memcpy(h + o_desc, desc.data(), desc.size() * sizeof(PreImg));
    memcpy(h + o_dir, dir.data(), dir.size() * sizeof(PreTaps));
    memcpy(h + o_ip, ipool.data(), ipool.size() * sizeof(int));
    memcpy(h + o_wp, wpool.data(), wpool.size() * sizeof(double));
    {
        static const int max_thr = [] { const char * e = getenv("CLIP_AMD_U8_THREADS"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : v > 64 ? 64 : v; }();
        const int nthr = std::max(1, std::min(max_thr, n));     // the pixel copy into pinned memory is the host cost of this path
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; t++)
            pool.emplace_back([&, t]() {
                for (int i = t; i < n; i += nthr) memcpy(h + o_raw + desc[i].src_off, imgs[i].data, (size_t)3 * imgs[i].nx * imgs[i].ny);
            });
        for (auto & th : pool) th.join();
    }
    if (!sl) {
        if (hipMemcpyAsync(dv, h, host_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return false;
    } else {
        hipStream_t cs = ctx->pre_copy_stream;
        if (sl->used && hipStreamWaitEvent(cs, sl->ev_done, 0) != hipSuccess) return false;      // the kernels that read this device blob two chunks ago
        if (hipMemcpyAsync(dv, h, host_bytes, hipMemcpyHostToDevice, cs) != hipSuccess) return false;
        if (hipEventRecord(sl->ev_h2d, cs) != hipSuccess || hipStreamWaitEvent(ctx->stream, sl->ev_h2d, 0) != hipSuccess) return false;
    }
    launch_preprocess(dv + o_raw, (const PreImg *)(dv + o_desc), (const PreTaps *)(dv + o_dir), (const double *)(dv + o_wp), (const int *)(dv + o_ip),
                      (float *)(dv + host_bytes), d_out, n, S, max_rows, ctx->image_mean, ctx->image_std, ctx->stream);
    if (sl) {
        if (hipEventRecord(sl->ev_done, ctx->stream) != hipSuccess) return false;
        sl->used = true;
    }
    return hipGetLastError() == hipSuccess;
}

}  // namespace clipamd
