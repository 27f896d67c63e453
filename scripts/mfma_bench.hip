// mfma_bench.hip — how many waves per SIMD does v_mfma_f32_16x16x32_f16 need to approach the matrix-pipe peak on gfx950?
// Standalone tuning tool:  hipcc --offload-arch=gfx950 -O3 scripts/mfma_bench.hip -o scripts/mfma_bench
// Each wave keeps NACC independent 16x16 accumulators and issues back-to-back MFMAs on them (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_kernel(float * out, int iters, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f + i); b[i] = (_Float16)(seed * 0.5f + i); }
    f4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    }
    f4 s = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NACC; j++) s += acc[j];
    if (s[0] == 123.456f) out[0] = s[1];
}

template <int NACC>
void run(int waves_per_simd, float * out) {
    const int threads = 64 * 4 * (waves_per_simd > 2 ? 2 : waves_per_simd);   // workgroup = 1 or 2 waves per SIMD
    const int wgs_per_cu = waves_per_simd > 2 ? waves_per_simd / 2 : 1;
    const int grid = 256 * wgs_per_cu, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_kernel<NACC>, dim3(grid), dim3(threads), 0, nullptr, out, iters, 1.0f);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(mfma_kernel<NACC>, dim3(grid), dim3(threads), 0, nullptr, out, iters, 1.0f);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * grid * (threads / 64) * (double)iters * 4 * NACC * 16384.0;
    printf("  waves/SIMD %d  NACC %2d : %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", waves_per_simd, NACC, flops / ms / 1e9,
           2.4e9 * (ms / 5 * 1e-3) / ((double)iters * 4 * NACC * waves_per_simd));
}

int main() {
    float * out;
    (void)hipMalloc(&out, 64);
    for (int w : {1, 2, 4}) {
        run<2>(w, out); run<4>(w, out); run<8>(w, out); run<16>(w, out); run<32>(w, out);
    }
    return 0;
}
