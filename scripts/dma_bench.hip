// dma_bench.hip — how fast can a GEMM-shaped workgroup grid stream its operand tiles L2 -> LDS on gfx950?
// Standalone tuning tool (not part of libclip.so):  hipcc --offload-arch=gfx950 -O3 scripts/dma_bench.hip -o scripts/dma_bench
//
// Every workgroup (256 threads) plays one 128x128 GEMM tile: per K-step it pulls a [128 x 64] fp16 slab of X and of W
// (16 KB each) into a double-buffered LDS ring with global_load_lds_dwordx4, waits, barriers — no MFMA, no ds_read.
// Modes (source address pattern per wave-instruction = 1 KB):
//   0  8 rows x 128 B, 16-byte chunks XOR-swizzled inside the row  (what k_gemm.hip does)
//   1  8 rows x 128 B, linear chunks
//   2  1 KB contiguous: operands pre-tiled in HBM as [tile][kstep][128 x 64]
//   3  like 0, but two K-steps in flight (3-slot ring)
//   4  plain global_load_dwordx4 into registers + ds_write_b128 (no LDS-DMA), rows as in mode 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 64, BM = 128, BN = 128;

template <int MODE>
__global__ void __launch_bounds__(256, 2) dma_kernel(const half_t * X, const half_t * W, int M, int N, int K, unsigned * sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SLOTS = MODE == 3 ? 3 : 2;
    half_t * Xs = (half_t *)smem;                  // [SLOTS][BM*BK]
    half_t * Ws = Xs + SLOTS * BM * BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N / BN, tiles_m = M / BM, nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int nk = K / BK;
    const int prow = lane >> 3;
    const int pchunk = MODE == 1 || MODE == 4 ? (lane & 7) : ((lane & 7) ^ prow);
    const half_t * xsrc[4];
    const half_t * wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (MODE == 2) {
            xsrc[i] = X + ((size_t)tile_m * nk) * (BM * BK) + (wave * 4 + i) * 512 + lane * 8;
            wsrc[i] = W + ((size_t)tile_n * nk) * (BN * BK) + (wave * 4 + i) * 512 + lane * 8;
        } else {
            xsrc[i] = X + (size_t)(tile_m * BM + (wave * 4 + i) * 8 + prow) * K + pchunk * 8;
            wsrc[i] = W + (size_t)(tile_n * BN + (wave * 4 + i) * 8 + prow) * K + pchunk * 8;
        }
    }
    const size_t kstride = MODE == 2 ? (size_t)BM * BK : (size_t)BK;
    unsigned acc = 0;
    auto issue = [&](int slot, int kt) {
        if constexpr (MODE == 4) {
            u32x4 rx[4], rw[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                rx[i] = *(const u32x4 *)(xsrc[i] + kt * kstride);
                rw[i] = *(const u32x4 *)(wsrc[i] + kt * kstride);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                *(u32x4 *)(Xs + slot * BM * BK + (wave * 4 + i) * 512 + lane * 8) = rx[i];
                *(u32x4 *)(Ws + slot * BN * BK + (wave * 4 + i) * 512 + lane * 8) = rw[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[i] + kt * kstride),
                    (__attribute__((address_space(3))) void *)(Xs + slot * BM * BK + (wave * 4 + i) * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[i] + kt * kstride),
                    (__attribute__((address_space(3))) void *)(Ws + slot * BN * BK + (wave * 4 + i) * 512), 16, 0, 0);
            }
        }
    };
    if constexpr (MODE == 3) {
        issue(0, 0);
        issue(1, nk > 1 ? 1 : 0);
        int slot = 2;
        for (int kt = 0; kt < nk; kt++) {
            const int t2 = kt + 2 < nk ? kt + 2 : nk - 1;
            issue(slot, t2);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the oldest K-step (8 pieces per wave) has landed
            __syncthreads();
            acc += *(const unsigned *)(Xs + ((slot + 1) % 3) * BM * BK + tid * 2);
            slot = (slot + 1) % 3;
        }
    } else {
        issue(0, 0);
        for (int kt = 0; kt < nk; kt++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int t1 = kt + 1 < nk ? kt + 1 : nk - 1;
            issue((kt + 1) & 1, t1);
            acc += *(const unsigned *)(Xs + (kt & 1) * BM * BK + tid * 2);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// Depth study at EQUAL occupancy (2 workgroups per CU): the same 128x128 tile streamed in 32-wide K-steps (16 KB per stage)
// through a 4-slot ring (64 KB of LDS) with DEPTH = 1, 2 or 3 stages requested ahead of the one being consumed.
template <int DEPTH>
__global__ void __launch_bounds__(256, 2) ring_kernel(const half_t * X, const half_t * W, int M, int N, int K, unsigned * sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = 32, SLOTS = 4, STAGE = (BM + BN) * KS;   // halfs per stage
    half_t * ring = (half_t *)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N / BN, tiles_m = M / BM, nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int nk = K / KS;
    // one wave-instruction = 16 rows x 64 B (4 lanes per row)
    const int prow = lane >> 2, pchunk = lane & 3;
    const half_t * xsrc[2];
    const half_t * wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        xsrc[i] = X + (size_t)(tile_m * BM + (wave * 2 + i) * 16 + prow) * K + pchunk * 8;
        wsrc[i] = W + (size_t)(tile_n * BN + (wave * 2 + i) * 16 + prow) * K + pchunk * 8;
    }
    auto issue = [&](int slot, int kt) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[i] + (size_t)kt * KS),
                (__attribute__((address_space(3))) void *)(ring + slot * STAGE + (wave * 2 + i) * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[i] + (size_t)kt * KS),
                (__attribute__((address_space(3))) void *)(ring + slot * STAGE + BM * KS + (wave * 2 + i) * 512), 16, 0, 0);
        }
    };
    unsigned acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) issue(d, d < nk ? d : nk - 1);
    for (int kt = 0; kt < nk; kt++) {
        // stage kt must have landed: DEPTH-1 younger stages (4 DMA instructions each) may still be in flight
        if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        const int tn = kt + DEPTH < nk ? kt + DEPTH : nk - 1;
        issue((kt + DEPTH) % SLOTS, tn);
        acc += *(const unsigned *)(ring + (kt % SLOTS) * STAGE + tid * 2);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH>
float run_ring(const half_t * X, const half_t * W, int M, int N, int K, unsigned * sink, int iters) {
    const size_t smem = (size_t)4 * (BM + BN) * 32 * 2;
    const int grid = (M / BM) * (N / BN);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(ring_kernel<DEPTH>, dim3(grid), dim3(256), smem, nullptr, X, W, M, N, K, sink);
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(ring_kernel<DEPTH>, dim3(grid), dim3(256), smem, nullptr, X, W, M, N, K, sink);
    (void)hipEventRecord(b, nullptr);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) return -1;
    return ms * 1000.f / iters;
}

template <int MODE>
float run(const half_t * X, const half_t * W, int M, int N, int K, unsigned * sink, int iters) {
    const size_t smem = (size_t)(MODE == 3 ? 3 : 2) * (BM + BN) * BK * 2;
    (void)hipFuncSetAttribute((const void *)dma_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int grid = (M / BM) * (N / BN);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(dma_kernel<MODE>, dim3(grid), dim3(256), smem, nullptr, X, W, M, N, K, sink);
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(dma_kernel<MODE>, dim3(grid), dim3(256), smem, nullptr, X, W, M, N, K, sink);
    (void)hipEventRecord(b, nullptr);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) return -1;
    return ms * 1000.f / iters;
}

int main() {
    const int shapes[][3] = {{12800, 2304, 768}, {12800, 768, 3072}, {12800, 3072, 768}, {65792, 1024, 4096}, {65792, 4096, 1024}, {12800, 768, 3008}, {65792, 1024, 4160}, {1664, 2304, 768}, {1664, 768, 3072}, {128, 3072, 768}, {128, 768, 3072}};
    for (auto & s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        half_t *X, *W;
        unsigned * sink;
        (void)hipMalloc(&X, (size_t)M * K * 2);
        (void)hipMalloc(&W, (size_t)N * K * 2);
        (void)hipMalloc(&sink, 64);
        (void)hipMemset(X, 0, (size_t)M * K * 2);
        (void)hipMemset(W, 0, (size_t)N * K * 2);
        const double bytes = (double)(M / BM) * (N / BN) * (K / BK) * (BM + BN) * BK * 2;
        const double flops = 2.0 * M * N * K;
        float t[5] = {run<0>(X, W, M, N, K, sink, 10), run<1>(X, W, M, N, K, sink, 10), run<2>(X, W, M, N, K, sink, 10), run<3>(X, W, M, N, K, sink, 10), run<4>(X, W, M, N, K, sink, 10)};
        printf("M=%6d N=%5d K=%5d  tile bytes %.2f GB |", M, N, K, bytes / 1e9);
        for (int m = 0; m < 5; m++) printf("  mode%d %7.1f us %5.1f TB/s (%4.0f TF-equiv)", m, t[m], bytes / t[m] / 1e6, flops / t[m] / 1e6);
        printf("\n");
        float r[3] = {run_ring<1>(X, W, M, N, K, sink, 10), run_ring<2>(X, W, M, N, K, sink, 10), run_ring<3>(X, W, M, N, K, sink, 10)};
        printf("    ring (4 x 16 KB slots, 2 WG/CU):");
        for (int d = 0; d < 3; d++) printf("  depth%d %7.1f us %5.1f TB/s (%4.0f TF-equiv)", d + 1, r[d], bytes / r[d] / 1e6, flops / r[d] / 1e6);
        printf("\n");
        (void)hipFree(X);
        (void)hipFree(W);
        (void)hipFree(sink);
    }
    return 0;
}
