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

}  // namespace clipamd
