// lds_mfma_bench.hip — ceiling of the LDS -> register -> MFMA inner loop of a tiled GEMM on gfx950, without global loads
// and without barriers.  Standalone tuning tool:  hipcc --offload-arch=gfx950 -O3 scripts/lds_mfma_bench.hip -o scripts/lds_mfma_bench
// A workgroup of 4 waves (2x2) owns a (32*TM*... ) tile exactly like k_gemm.hip: per K-step of 64 each wave reads
// 2*(TN+TM) 16-byte fragments (swizzled [rows][64] fp16 tiles) and issues 2*TN*TM v_mfma_f32_16x16x32_f16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int lds_off(int r, int c) { return r * 64 + ((c ^ (r & 7)) << 3); }

template <int TN, int TM, int PIN, int WGPC, int BAR = 0>
__global__ void __launch_bounds__(256, WGPC) loop_kernel(float * out, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BN = TN * 32, BM = TM * 32;
    half_t * Xs = (half_t *)smem;             // [2][BM*64]
    half_t * Ws = Xs + 2 * BM * 64;           // [2][BN*64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * (BM + BN) * 64; i += 256) Xs[i] = (_Float16)((i % 97) * 0.01f);
    __syncthreads();
    const int wn = wave >> 1, wm = wave & 1, frow = lane & 15, fgrp = lane >> 4;
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) {
        const half_t * xs = Xs + (s & 1) * BM * 64;
        const half_t * ws = Ws + (s & 1) * BN * 64;
        h8 xf[2][TM], wf[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
#pragma unroll
            for (int a = 0; a < TN; a++) wf[kk][a] = *(const h8 *)(ws + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fgrp));
#pragma unroll
            for (int b = 0; b < TM; b++) xf[kk][b] = *(const h8 *)(xs + lds_off(wm * (BM / 2) + b * 16 + frow, kk * 4 + fgrp));
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][a], xf[kk][b], acc[a][b], 0, 0, 0);
        if (PIN) {
            __builtin_amdgcn_sched_group_barrier(0x100, TN + TM, 0);
#pragma unroll
            for (int i = 0; i < TN + TM; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, (TN * TM) / (TN + TM), 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN * TM - (TN + TM) * ((TN * TM) / (TN + TM)), 0);
        }
        if (BAR == 1) __syncthreads();                 // what the real K loop needs: one workgroup barrier per step
        if (BAR == 2) { if ((s & 1) == 1) __syncthreads(); }   // a barrier every other step
        if (BAR == 3) __builtin_amdgcn_s_barrier();    // bare s_barrier, no s_waitcnt
    }
    f4 sum = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) sum += acc[a][b];
    if (sum[0] == 123.456f) out[0] = sum[1];
}

template <int TN, int TM, int PIN, int WGPC, int BAR = 0>
void run(float * out) {
    // wave tile = (TN/2*16 rows of W) x (TM/2*16 rows of X)?  no: as in k_gemm.hip each wave owns TN x TM fragments of its (BN/2)x(BM/2) quadrant,
    // so the workgroup tile here is BN = 32*TN by BM = 32*TM.
    constexpr int BN = TN * 32, BM = TM * 32;
    size_t smem = (size_t)2 * (BM + BN) * 64 * 2;
    const size_t want = WGPC == 1 ? 100 * 1024 : WGPC == 2 ? 70 * 1024 : 48 * 1024;    // pad LDS so that exactly WGPC workgroups fit a CU
    if (smem < want) smem = want;
    (void)hipFuncSetAttribute((const void *)loop_kernel<TN, TM, PIN, WGPC, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int grid = 256 * WGPC, steps = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<TN, TM, PIN, WGPC, BAR>), dim3(grid), dim3(256), smem, nullptr, out, steps);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((loop_kernel<TN, TM, PIN, WGPC, BAR>), dim3(grid), dim3(256), smem, nullptr, out, steps);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 3.0 * grid * 4.0 * steps * 2 * TN * TM * 16384.0;
    printf("  tile %3dx%3d (wave frags %dx%d)  WG/CU %d  pinned %d barrier %d : %7.1f TFLOP/s   reads:MFMA = %d:%d\n", BM, BN, TN, TM, WGPC, PIN, BAR, flops / ms / 1e9, 2 * (TN + TM), 2 * TN * TM);
}

int main() {
    float * out;
    (void)hipMalloc(&out, 64);
    run<4, 5, 1, 2, 1>(out); run<4, 5, 1, 2, 2>(out); run<4, 5, 1, 2, 3>(out); run<4, 4, 1, 2, 1>(out); run<4, 4, 1, 3, 1>(out); run<4, 6, 1, 2, 1>(out); run<8, 8, 1, 1, 1>(out);
    run<4, 4, 1, 2>(out); run<4, 4, 0, 2>(out); run<4, 4, 1, 1>(out); run<4, 4, 1, 3>(out);
    run<4, 5, 1, 2>(out); run<4, 5, 0, 2>(out); run<4, 5, 1, 1>(out);
    run<4, 6, 1, 2>(out); run<4, 6, 1, 1>(out);
    run<2, 2, 1, 2>(out); run<2, 2, 1, 3>(out);
    run<8, 4, 1, 1>(out); run<8, 5, 1, 1>(out); run<8, 8, 1, 1>(out); run<8, 8, 0, 1>(out);
    return 0;
}
