// micro-benchmark: cost of a bare s_barrier loop iteration (cycles, s_memtime) by workgroups per CU, LDS size and loop shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long * out, int iters, int flag) {
    extern __shared__ unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int acc = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE >= 1) {      // wave-dependent counted waits as in the ring kernel
            if (wave == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (wave == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (wave == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE >= 2) {      // skipped sections (uniform branches on a kernel argument)
            if (flag & 1) { smem[threadIdx.x] = (unsigned char)i; acc += smem[(threadIdx.x + 7) & 255]; }
            if (flag & 2) { acc += __builtin_amdgcn_readfirstlane(i) * 3; asm volatile("s_nop 0" ::: "memory"); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
template <int MODE> void run(const char * name, int grid, size_t lds, int iters) {
    unsigned long long * d; hipMalloc(&d, grid * 16);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, iters, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; i++) s += h[i * 2];
    printf("%-28s grid %5d lds %6zu : %8.1f cycles / iteration\n", name, grid, lds, s / grid / iters);
    hipFree(d);
}
int main() {
    for (int grid : {256, 1024, 2048}) for (size_t lds : {(size_t)1024, (size_t)70000}) {
        run<0>("bare s_barrier", grid, lds, 1000);
        run<1>("+ per-wave waitcnt switch", grid, lds, 1000);
        run<2>("+ skipped sections", grid, lds, 1000);
    }
    return 0;
}
