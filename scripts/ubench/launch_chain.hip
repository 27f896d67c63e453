// launch_chain.hip — what ONE dependent kernel boundary costs on this part, at the grid sizes of the small-M ("skinny") layer kernels
// (DESIGN.md's "the floor of a dependent launch here is ~4.5-5 us" had no measurement behind it).
//
// A batch-1 ViT-B/32 forward is ~66 dependent launches of 192-768 workgroups x 256 threads.  This program times chains of 66 launches
// on one stream, eager and as a replayed hipGraph, for three kernel bodies:
//   empty   : s_endpgm only                                              -> dispatch + completion + boundary
//   touch   : every thread reads 16 B written by the previous launch and writes 16 B (a dependent memory round trip through L2)
//   stream  : every workgroup reads 64 KB (what a skinny GEMM workgroup pulls: weights + activation rows) and writes 1 KB
//   xtouch  : as touch, but the 16 B come from the workgroup on the NEXT die (cross-XCD hand-off through the kernel boundary)
// Output: microseconds per launch = chain time / 66.   Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/launch_chain scripts/ubench/launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}

__global__ void __launch_bounds__(256) k_touch(const float4 * __restrict__ in, float4 * __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = in[i];
    v.x += 1.0f;
    out[i] = v;
}

// the same round trip, but every workgroup reads what ANOTHER XCD's workgroup wrote in the previous launch (block b runs on XCD b % 8:
// block b + 1 is the neighbour die) — the case of a layer chain, where a row written by one die is read by all of them
__global__ void __launch_bounds__(256) k_xtouch(const float4 * __restrict__ in, float4 * __restrict__ out) {
    const int src = ((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x;
    float4 v = in[src];
    v.x += 1.0f;
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

// 64 KB per workgroup: 256 threads x 16 loads x 16 B, all in flight, then one 16-byte store per 4th thread
__global__ void __launch_bounds__(256) k_stream(const float4 * __restrict__ in, float4 * __restrict__ out, int wrap) {
    const int base = (blockIdx.x % wrap) * 4096 + threadIdx.x;
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = in[base + j * 256];
    float4 s = v[0];
#pragma unroll
    for (int j = 1; j < 16; j++) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    if ((threadIdx.x & 3) == 0) out[blockIdx.x * 64 + (threadIdx.x >> 2)] = s;
}

int main() {
    const int CHAIN = 66, REPS = 200;
    const int grids[] = {192, 576, 768, 2304};
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float4 * a = nullptr, * b = nullptr;
    const size_t n4 = (size_t)4096 * 4096;              // 256 MB each: the stream body wraps inside it
    CK(hipMalloc(&a, n4 * sizeof(float4)));
    CK(hipMalloc(&b, n4 * sizeof(float4)));
    CK(hipMemset(a, 0, n4 * sizeof(float4)));
    CK(hipMemset(b, 0, n4 * sizeof(float4)));
    printf("# chain of %d dependent launches on one stream, %d repetitions; us per launch (wall clock around the chain, host synchronised at both ends)\n", CHAIN, REPS);
    printf("# %-8s %6s | %10s %10s\n", "body", "grid", "eager", "graph");
    for (int body = 0; body < 4; body++) {
        for (int g : grids) {
            auto launch_chain = [&](hipStream_t st) {
                for (int i = 0; i < CHAIN; i++) {
                    float4 * in = (i & 1) ? b : a, * out = (i & 1) ? a : b;
                    if (body == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, st);
                    else if (body == 1) hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, st, in, out);
                    else if (body == 2) hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, st, in, out, 4096);
                    else hipLaunchKernelGGL(k_xtouch, dim3(g), dim3(256), 0, st, in, out);
                }
            };
            double us[2] = {0, 0};
            // eager
            for (int w = 0; w < 5; w++) launch_chain(s);
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < REPS; r++) launch_chain(s);
            CK(hipStreamSynchronize(s));
            us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REPS * CHAIN);
            // graph replay
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            launch_chain(s);
            CK(hipStreamEndCapture(s, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int w = 0; w < 5; w++) CK(hipGraphLaunch(exec, s));
            CK(hipStreamSynchronize(s));
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < REPS; r++) CK(hipGraphLaunch(exec, s));
            CK(hipStreamSynchronize(s));
            us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REPS * CHAIN);
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
            printf("  %-8s %6d | %10.2f %10.2f\n", body == 0 ? "empty" : body == 1 ? "touch" : body == 2 ? "stream" : "xtouch", g, us[0], us[1]);
            fflush(stdout);
        }
    }
    return 0;
}
