// micro-test: is a VALU write to the SOURCE register of the trans op issued just before it safe on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
template <int GAP>
__global__ void k(const float * x, float * out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = x[i], b = a + 1.0f, r1 = 0.f, r2 = 0.f, acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float t;
        if constexpr (GAP == 0)
            asm volatile("v_mul_f32 %2, 0.5, %3\n v_exp_f32 %0, %2\n v_mul_f32 %2, 0.5, %4\n v_exp_f32 %1, %2\n"
                         : "=&v"(r1), "=&v"(r2), "=&v"(t) : "v"(a), "v"(b));
        else if constexpr (GAP == 1)
            asm volatile("v_mul_f32 %2, 0.5, %3\n v_exp_f32 %0, %2\n v_nop\n v_mul_f32 %2, 0.5, %4\n v_exp_f32 %1, %2\n"
                         : "=&v"(r1), "=&v"(r2), "=&v"(t) : "v"(a), "v"(b));
        else
            asm volatile("v_mul_f32 %2, 0.5, %3\n v_exp_f32 %0, %2\n s_nop 3\n v_mul_f32 %2, 0.5, %4\n v_exp_f32 %1, %2\n"
                         : "=&v"(r1), "=&v"(r2), "=&v"(t) : "v"(a), "v"(b));
        asm volatile("s_nop 7\n s_nop 7" ::: "memory");
        acc += (r1 != exp2f(0.5f * a)) ? 1.0f : 0.0f;      // (v_exp_f32 is exp2)
        acc += (r2 != exp2f(0.5f * b)) ? 1.0f : 0.0f;
    }
    out[i] = acc;
}
template <int GAP> void run(const float * dx, float * dout, int n, const char * name) {
    hipLaunchKernelGGL(k<GAP>, dim3(n / 256), dim3(256), 0, 0, dx, dout, 200);
    std::vector<float> h(n);
    hipMemcpy(h.data(), dout, n * 4, hipMemcpyDeviceToHost);
    double bad = 0; int lanes = 0;
    for (float v : h) { bad += v; lanes += v != 0; }
    printf("%s: mismatching results %.0f (lanes affected %d of %d)\n", name, bad, lanes, n);
}
int main() {
    const int n = 256 * 2048;
    std::vector<float> hx(n);
    for (int i = 0; i < n; i++) hx[i] = (float)(i % 997) * 0.01f;
    float * dx, * dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        run<0>(dx, dout, n, "trans src overwritten by the next VALU    ");
        run<1>(dx, dout, n, "one v_nop in between                       ");
        run<2>(dx, dout, n, "s_nop 3 in between                         ");
    }
    return 0;
}
