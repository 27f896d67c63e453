// k_fold.hip — load-time companion of the LayerNorm fold (gemm_common.h, "LayerNorm folded into the GEMMs around it").
//
// The reference applies ggml_norm + mul(gamma) + add(beta) and then ggml_mul_mat with the projection weight
// (clip.cpp:1350-1380, :1400-1407; text :1071-1095, :1121-1127).  Algebraically
//     W (x_hat gamma + beta) + b  =  rstd (W (x gamma) - mean c) + b',     c_n = sum_k gamma_k W_nk,   b'_n = sum_k beta_k W_nk + b_n
// so two vectors per (LayerNorm, projection) pair, computed ONCE per model load, let the GEMM epilogue finish the normalisation.
// They are computed here, on the device, from the weights exactly as the GEMM kernels see them: the same dequant_wfrag arithmetic
// (packed fp16 (q - zero) d [+ m], one rounding), so mean * c_n cancels against what the MFMA actually accumulated.
// One thread per weight row, sequential over k in double precision: deterministic, and a few hundred microseconds per model.

#include "gemm_common.h"

namespace clipamd {

namespace {

template <int WT>
__global__ void __launch_bounds__(256) fold_kernel(const DevWeight W, const float * __restrict__ gamma, const float * __restrict__ beta,
                                                   const float * __restrict__ bias, float * __restrict__ c_out, float * __restrict__ b_out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= W.N) return;
    double sc = 0.0, sb = 0.0;
    if constexpr (WT == W_F16) {
        const half_t * row = (const half_t *)W.w16 + (size_t)n * W.Kpad;
        for (int k0 = 0; k0 < W.K; k0 += 8) {                  // (rows are padded to a multiple of 64 halfs: the 16-byte load stays inside)
            const h8 w8 = *(const h8 *)(row + k0);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (k0 + e < W.K) {
                    const double w = (double)(float)w8[e];
                    sc += (double)gamma[k0 + e] * w;
                    sb += (double)beta[k0 + e] * w;
                }
            }
        }
    } else {
        const int nkb = W.K / 32;
        for (int kb = 0; kb < nkb; kb++) {
            RawBlock<WT> r;
            load_block<WT>(r, W, (size_t)kb * W.Npad + n);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const h8 w8 = dequant_wfrag<WT>(block_word<WT>(r, j), j);     // elements 8j .. 8j+7 of the block, in order
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = kb * 32 + j * 8 + e;
                    const double w = (double)(float)w8[e];
                    sc += (double)gamma[k] * w;
                    sb += (double)beta[k] * w;
                }
            }
        }
    }
    c_out[n] = (float)sc;
    b_out[n] = (float)(sb + (bias ? (double)bias[n] : 0.0));
}

}  // namespace

void launch_fold_vectors(const DevWeight & W, const float * gamma, const float * beta, const float * bias, float * c_out, float * b_out,
                         hipStream_t stream) {
    const dim3 grid((W.N + 255) / 256), block(256);
    switch (W.wtype) {
    case W_F16: hipLaunchKernelGGL(fold_kernel<W_F16>, grid, block, 0, stream, W, gamma, beta, bias, c_out, b_out); break;
    case W_Q4_0: hipLaunchKernelGGL(fold_kernel<W_Q4_0>, grid, block, 0, stream, W, gamma, beta, bias, c_out, b_out); break;
    case W_Q4_1: hipLaunchKernelGGL(fold_kernel<W_Q4_1>, grid, block, 0, stream, W, gamma, beta, bias, c_out, b_out); break;
    case W_Q5_0: hipLaunchKernelGGL(fold_kernel<W_Q5_0>, grid, block, 0, stream, W, gamma, beta, bias, c_out, b_out); break;
    case W_Q5_1: hipLaunchKernelGGL(fold_kernel<W_Q5_1>, grid, block, 0, stream, W, gamma, beta, bias, c_out, b_out); break;
    case W_Q8_0: hipLaunchKernelGGL(fold_kernel<W_Q8_0>, grid, block, 0, stream, W, gamma, beta, bias, c_out, b_out); break;
    }
}

}  // namespace clipamd
