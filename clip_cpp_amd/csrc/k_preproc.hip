// k_preproc.hip — image preprocessing on the GPU (reference clip.cpp:728-1008, host twin: preprocess.cpp).
//
// Raw u8 RGB images of arbitrary size -> [S][S][3] f32 normalised encoder inputs, bit-identical to the host path:
// the separable antialiased bicubic taps are computed once per (source size, target size) on the host in double
// (preprocess.cpp make_taps) and uploaded; the two passes accumulate in double over the taps in increasing source index,
// round to float and clamp to [0,255] after each pass — the same operations in the same order as the host code (the
// build uses -ffp-contract=off, and HIP's f32 division is correctly rounded), so the results match bit for bit.
// Only what the centre crop needs is computed: S output columns of the horizontal pass and the source rows the S output
// rows of the vertical pass touch.  HBM-bound on the u8 source (3 B/pixel read once).
#include "kernels.h"

namespace clipamd {

namespace {

__device__ __forceinline__ float clamp255(float v) { return fminf(fmaxf(v, 0.0f), 255.0f); }

// horizontal pass: thread = (output column xo, needed source row r) of image blockIdx.z
__global__ void __launch_bounds__(256) preproc_h_kernel(const uint8_t * raw, const PreImg * imgs, const PreTaps * taps, const double * wpool,
                                                        const int * ipool, float * hbuf, int S) {
    const PreImg im = imgs[blockIdx.z];
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (xo >= S || r >= im.nrows) return;
    const PreTaps t = taps[im.th];
    const int o = im.x0 + xo;
    const double * k = wpool + t.w_off + (size_t)o * t.ksize;
    const int lo = ipool[t.first_off + o], n = ipool[t.count_off + o];
    const uint8_t * srow = raw + im.src_off + (size_t)3 * (im.ylo + r) * im.nx + (size_t)3 * lo;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int x = 0; x < n; x++) {
        const double w = k[x];
        a0 += (double)srow[3 * x + 0] * w;
        a1 += (double)srow[3 * x + 1] * w;
        a2 += (double)srow[3 * x + 2] * w;
    }
    float * d = hbuf + im.hbuf_off + ((size_t)r * S + xo) * 3;
    d[0] = clamp255((float)a0);
    d[1] = clamp255((float)a1);
    d[2] = clamp255((float)a2);
}

// vertical pass + centre crop + normalisation: thread = (xo, yo) of image blockIdx.z
__global__ void __launch_bounds__(256) preproc_v_kernel(const PreImg * imgs, const PreTaps * taps, const double * wpool, const int * ipool,
                                                        const float * hbuf, float * out, int S, float m0, float m1, float m2, float s0,
                                                        float s1, float s2) {
    const PreImg im = imgs[blockIdx.z];
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y;
    if (xo >= S) return;
    const PreTaps t = taps[im.tv];
    const int o = im.y0 + yo;
    const double * k = wpool + t.w_off + (size_t)o * t.ksize;
    const int lo = ipool[t.first_off + o], n = ipool[t.count_off + o];
    const float * src = hbuf + im.hbuf_off + ((size_t)(lo - im.ylo) * S + xo) * 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int y = 0; y < n; y++) {
        const double w = k[y];
        const float * s = src + (size_t)y * S * 3;
        a0 += (double)s[0] * w;
        a1 += (double)s[1] * w;
        a2 += (double)s[2] * w;
    }
    float * d = out + ((size_t)blockIdx.z * S * S + (size_t)yo * S + xo) * 3;
    d[0] = ((clamp255((float)a0) / 255.0f) - m0) / s0;
    d[1] = ((clamp255((float)a1) / 255.0f) - m1) / s1;
    d[2] = ((clamp255((float)a2) / 255.0f) - m2) / s2;
}

}  // namespace

void launch_preprocess(const uint8_t * raw, const PreImg * imgs, const PreTaps * taps, const double * wpool, const int * ipool, float * hbuf,
                       float * out, int n_imgs, int S, int max_rows, const float * mean, const float * stdv, hipStream_t stream) {
    if (n_imgs <= 0) return;
    const int bx = 256;
    dim3 gh((S + bx - 1) / bx, max_rows, n_imgs), gv((S + bx - 1) / bx, S, n_imgs);
    hipLaunchKernelGGL(preproc_h_kernel, gh, dim3(bx), 0, stream, raw, imgs, taps, wpool, ipool, hbuf, S);
    hipLaunchKernelGGL(preproc_v_kernel, gv, dim3(bx), 0, stream, imgs, taps, wpool, ipool, (const float *)hbuf, out, S, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2]);
}

}  // namespace clipamd
