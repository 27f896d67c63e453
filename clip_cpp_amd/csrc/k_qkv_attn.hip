// k_qkv_attn.hip — small-M path, first kernel of a layer: LayerNorm1 + q/k/v projection + attention in ONE launch.
//
//   out[rows][h] = softmax(Q K^T [causal]) V  per (sequence, head),   Q/K/V = LN1(x) · W_{q,k,v}^T + b   (Q scaled by 1/sqrt(d_head))
//   for sequences of <= 64 rows (one ViT-B/32 image: 50 token rows; a text of <= 64 tokens) and d_head = 64
//
// At this size a layer is a chain of dependent launches that each cost ~5 us before they do anything (r02: 5 launches per layer,
// 36 us per ViT-B/32 layer at batch 1).  The q/k/v projection (k_skinny.hip) and the attention (k_attn.hip) are fused by giving one
// workgroup everything ONE (sequence, head) needs: its 3 x 64 projection columns for all rows of the sequence (a 64 x 192 x K GEMM:
// 83 KB of q4_0 weights), then the attention of that head.  VERDICT r1 item 4 ("QKV + attention in one workgroup cluster").
//
//   * 512 threads = 8 waves.  Wave w owns column group cs = w & 3 of each of q, k, v (16 columns each: 3 weight fragments per
//     k-block).  Sequences of more than 16 rows: the two wave quartets split the ROW blocks (each over the whole K, no reduction);
//     16 rows or fewer (short texts): they split K, and the upper half is added through LDS in a fixed order (deterministic).
//     LN1 of the sequence's rows is computed ONCE per workgroup (row statistics from the producer's partial slots, as k_skinny.hip) and
//     parked in LDS as fp16 — the MFMA B operand of every wave (99 KB for 64 rows of 768); weights come as MFMA fragments straight
//     from the block-column-major planes, twelve k-blocks in flight per wave, and are dequantised in registers;
//   * bias + Q scale applied, the head's q/k/v rows written to the q/k/v buffer in HBM/L2 (fp16, the layout attn_kernel reads) — then
//     one barrier, and the workgroup runs attn_body on the rows it just wrote (read back through its own XCD's L2: the same CU wrote
//     them, vmcnt(0) + barrier in between);
//   * n_seq x n_head workgroups (12 for one ViT-B/32 image): few, but each is one memory round trip + ~150 MFMAs per wave + the
//     attention — and a launch boundary plus the 144-workgroup projection kernel's ramp are gone.
//
// Numerics: as k_skinny.hip (fp32 accumulation in two partial sums per output, LayerNorm variance as E[x^2] - mean^2), attention as
// k_attn.hip.  Reference ops replaced: clip.cpp:1350-1388 (vision), :1064-1108 (text).

#include <cstdlib>

#include "gemm_common.h"
#include "attn_body.h"

namespace clipamd {

namespace {

constexpr int QA_THREADS = 512;

__device__ __forceinline__ h8 qa_normalise8(const f4 lo, const f4 hi, float mean, float rstd, const f4 g0, const f4 g1, const f4 b0, const f4 b1) {
    const f4 y0 = ((lo - mean) * rstd) * g0 + b0;     // same expression order as layernorm_kernel / k_skinny.hip
    const f4 y1 = ((hi - mean) * rstd) * g1 + b1;
    return (h8){(_Float16)y0[0], (_Float16)y0[1], (_Float16)y0[2], (_Float16)y0[3], (_Float16)y1[0], (_Float16)y1[1], (_Float16)y1[2], (_Float16)y1[3]};
}

// NT = 16-row blocks of the sequence = 16-key tiles of its attention (1..4)
template <int WT, int NT>
__global__ void __launch_bounds__(QA_THREADS) qkv_attn_kernel(const QkvAttnParams p) {
    constexpr bool RS = NT >= 2;                      // row blocks split between the wave quartets (else K split)
    constexpr int MB = RS ? (NT + 1) / 2 : 1;         // row blocks per wave
    constexpr int PF = (WT == W_Q5_0 || WT == W_Q5_1) ? 3 : 4;   // k-blocks of weight fragments in flight per half of the double buffer (bounded by the 256 VGPRs of an 8-wave workgroup)
    constexpr size_t ATT_BYTES = ((size_t)NT * 16 * (64 + 8) + (size_t)64 * (((NT + 1) / 2) * 32 + 8)) * sizeof(half_t);   // attn_body<NT, 2, 4>
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];   // [attention tiles of phase 2][normalised rows Xs]
    half_t * Xs = (half_t *)(smem_raw + ((ATT_BYTES + 255) & ~(size_t)255));   // [NT * 16][K + 8] fp16 = LN1(x) rows of the sequence
    __shared__ f4 red[RS ? 1 : 4 * 3 * 64];           // K split: partial fragments of the upper half
    __shared__ float2 lnst[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int seq = blockIdx.x / p.n_head, head = blockIdx.x % p.n_head;
    int row0, len;
    if (p.seq_start) {
        row0 = p.seq_start[seq];
        len = p.seq_start[seq + 1] - row0;
    } else {
        row0 = seq * p.T_uniform;
        len = p.T_uniform;
    }
    const int h = p.h, K = p.W.K, xs_ld = K + 8;
    const int cs = wave & 3, kh = wave >> 2;
    const int nkb = p.W.Kpad / 32;
    const int kb_lo = RS ? 0 : kh * nkb / 2, kb_hi = RS ? nkb : (kh + 1) * nkb / 2;
    const int b0 = RS ? kh * MB : 0;                  // first row block of this wave
    int ncol[3];                                      // first projection column of this wave's q / k / v group
#pragma unroll
    for (int g = 0; g < 3; g++) ncol[g] = g * h + head * 64 + cs * 16;

    struct WReg { WFrag<WT> q; h8 hh; };
    auto load_w = [&](WReg & w, int kb, int g) {
        if constexpr (WT == W_F16) {
            w.hh = *(const h8 *)((const half_t *)p.W.w16 + (size_t)(ncol[g] + frow) * p.W.Kpad + kb * 32 + fgrp * 8);
        } else {
            const size_t idx = (size_t)kb * p.W.Npad + ncol[g] + frow;
            if constexpr (WT == W_Q8_0) {
                const uint2 q = ((const uint2 *)p.W.qs)[idx * 4 + fgrp];
                w.q.q = q.x;
                w.q.q1 = q.y;
            } else {
                w.q.q = ((const uint32_t *)p.W.qs)[idx * 4 + fgrp];
            }
            if constexpr (WT == W_Q5_0 || WT == W_Q5_1) w.q.h = ((const uint32_t *)p.W.qh)[idx];
            if constexpr (WT == W_Q4_1 || WT == W_Q5_1) w.q.dm = ((const h2 *)p.W.dm)[idx];
            else w.q.d = ((const half_t *)p.W.dm)[idx];
        }
    };
    WReg wA[PF][3], wB[PF][3];
#define QA_LOADW(W_, c_)                                                                       \
    _Pragma("unroll") for (int i = 0; i < PF; i++)                                             \
        if ((c_) + i < kb_hi) {                                                                \
            _Pragma("unroll") for (int g = 0; g < 3; g++) load_w(W_[i][g], (c_) + i, g);       \
        }
    QA_LOADW(wA, kb_lo);                              // the first weight fragments go out before anything else
    f4 bias_pre[3];
#pragma unroll
    for (int g = 0; g < 3; g++) bias_pre[g] = p.bias ? *(const f4 *)(p.bias + ncol[g] + fgrp * 4) : (f4){0.f, 0.f, 0.f, 0.f};

    // ---- phase 0: LN1 of the sequence's rows, once per workgroup, into LDS as fp16 (the MFMA B operand of every wave).
    // Wave w normalises rows w, w + 8, ...; a lane holds the float4 columns lane, lane + 64, ... of a row (K <= 1024: at most 4).
    {
        constexpr int TPR = QA_THREADS / 64;          // row statistics: 8 threads per row add the producer's partial slots (fixed tree)
        const int r = tid / TPR, sub = tid % TPR;     // r: row inside the sequence (clamped)
        const float2 * row = p.stats_in + (size_t)(row0 + (r < len ? r : len - 1)) * p.stats_cap;
        float s1 = 0.f, s2 = 0.f;
        for (int base = 0; base < p.stats_slots; base += TPR * 16) {
            float2 v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int sl = base + i * TPR + sub;
                v[i] = sl < p.stats_slots ? row[sl] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) { s1 += v[i].x; s2 += v[i].y; }
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if (sub == 0) {
            const float mu = s1 / (float)K;
            float var = s2 / (float)K - mu * mu;
            var = var > 0.f ? var : 0.f;
            lnst[r] = make_float2(mu, 1.0f / sqrtf(var + p.eps));
        }
    }
    {
        constexpr int RPW = NT * 16 / 8;              // rows per wave (2 .. 8)
        constexpr int RB = RPW > 4 ? RPW / 2 : RPW;   // ... in batches of at most 4 (registers)
        f4 gw[4], gb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = (i * 64 + lane) * 4;
            gw[i] = c < K ? *(const f4 *)(p.ln_w + c) : (f4){0.f, 0.f, 0.f, 0.f};
            gb[i] = c < K ? *(const f4 *)(p.ln_b + c) : (f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q0 = 0; q0 < RPW; q0 += RB) {
            f4 xv[RB][4];
#pragma unroll
            for (int q = 0; q < RB; q++) {
                const int r = wave + 8 * (q0 + q);
                const float * xr = p.x32 + (size_t)(row0 + (r < len ? r : len - 1)) * p.ldx;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = (i * 64 + lane) * 4;
                    xv[q][i] = c < K ? *(const f4 *)(xr + c) : (f4){0.f, 0.f, 0.f, 0.f};
                }
            }
            if (q0 == 0) __syncthreads();             // row statistics are in lnst
#pragma unroll
            for (int q = 0; q < RB; q++) {
                const int r = wave + 8 * (q0 + q);
                const float2 st = lnst[r];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int c = (i * 64 + lane) * 4;
                    if (c < K) {
                        const f4 y = ((xv[q][i] - st.x) * st.y) * gw[i] + gb[i];     // same expression order as layernorm_kernel / k_skinny.hip
                        const h2 lo = (h2){(_Float16)y[0], (_Float16)y[1]}, hi = (h2){(_Float16)y[2], (_Float16)y[3]};
                        *(uint2 *)(Xs + (size_t)r * xs_ld + c) = make_uint2(h2u(lo), h2u(hi));
                    }
                }
            }
        }
    }
    __syncthreads();                                  // Xs complete

    // ---- phase 1: q/k/v of the head.  A fragments (weights) from the planes, B fragments (rows) from Xs
    f4 acc[3][MB];
    int xrow[MB];                                     // Xs row of fragment row frow in block b0 + b (a block past the sequence's NT re-reads the last one; never stored)
#pragma unroll
    for (int b = 0; b < MB; b++) { const int r = (b0 + b) * 16 + frow; xrow[b] = r < NT * 16 ? r : NT * 16 - 16 + frow; }
#pragma unroll
    for (int g = 0; g < 3; g++)
#pragma unroll
        for (int b = 0; b < MB; b++) acc[g][b] = (f4){0.f, 0.f, 0.f, 0.f};
#define QA_COMPUTE(W_, c_)                                                                     \
    _Pragma("unroll") for (int i = 0; i < PF; i++)                                             \
        if ((c_) + i < kb_hi) {                                                                \
            h8 xf[MB];                                                                         \
            _Pragma("unroll") for (int b = 0; b < MB; b++)                                     \
                xf[b] = *(const h8 *)(Xs + (size_t)xrow[b] * xs_ld + ((c_) + i) * 32 + fgrp * 8);    \
            _Pragma("unroll") for (int g = 0; g < 3; g++) {                                    \
                h8 wf;                                                                         \
                if constexpr (WT == W_F16) wf = W_[i][g].hh; else wf = dequant_wfrag<WT>(W_[i][g].q, fgrp); \
                _Pragma("unroll") for (int b = 0; b < MB; b++)                                 \
                    acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[g][b], 0, 0, 0); \
            }                                                                                  \
            __builtin_amdgcn_sched_barrier(0);   /* one k-block at a time: hoisted, the dequantised fragments of a whole chunk are live at once (spills) */ \
        }
    for (int c = kb_lo; c < kb_hi; c += 2 * PF) {
        QA_LOADW(wB, c + PF);
        QA_COMPUTE(wA, c);
        QA_LOADW(wA, c + 2 * PF);
        QA_COMPUTE(wB, c + PF);
    }
#undef QA_LOADW
#undef QA_COMPUTE

    // ---- (K split: the upper half through LDS, added as lower + upper;) bias, Q scale, fp16, the head's q/k/v rows to memory
    if constexpr (!RS) {
        if (kh == 1) {
#pragma unroll
            for (int g = 0; g < 3; g++) red[(cs * 3 + g) * 64 + lane] = acc[g][0];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int g = 0; g < 3; g++) acc[g][0] = acc[g][0] + red[(cs * 3 + g) * 64 + lane];
        }
    }
    if (RS || kh == 0) {
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int b = 0; b < MB; b++) {
                f4 v = acc[g][b] + bias_pre[g];
                if (g == 0) v = v * p.qscale;         // Q = (W_q x + b_q) / sqrt(d_head): scale after the bias (clip.cpp:1363)
                const int m = (b0 + b) * 16 + frow;
                const h2 lo = (h2){(_Float16)v[0], (_Float16)v[1]};
                const h2 hi = (h2){(_Float16)v[2], (_Float16)v[3]};
                if (m < len && (b0 + b) < NT) *(uint2 *)(p.qkv + (size_t)(row0 + m) * (3 * h) + ncol[g] + fgrp * 4) = make_uint2(h2u(lo), h2u(hi));
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows are in L2 before anybody in the workgroup reads them back
    __syncthreads();

    AttnParams ap;
    ap.qkv = p.qkv;
    ap.out = p.out;
    ap.seq_start = p.seq_start;
    ap.T_uniform = p.T_uniform;
    ap.h = h;
    ap.n_head = p.n_head;
    ap.causal = p.causal;
    attn_body<NT, 2, 4, QA_THREADS>(ap, seq, head, 0, 1, smem_raw);
}

template <int WT, int NT>
void launch_qa(const QkvAttnParams & p, hipStream_t stream) {
    constexpr size_t att = ((size_t)NT * 16 * (64 + 8) + (size_t)64 * (((NT + 1) / 2) * 32 + 8)) * sizeof(half_t);   // as attn_kernel<NT, 2, 4>
    const size_t smem = ((att + 255) & ~(size_t)255) + (size_t)NT * 16 * (p.W.K + 8) * sizeof(half_t);             // + the normalised rows
    constexpr size_t smem_max = ((att + 255) & ~(size_t)255) + (size_t)NT * 16 * (1024 + 8) * sizeof(half_t);        // K <= 1024
    static unsigned long long lds_ok = 0;
    if (smem_max > 64 * 1024) opt_in_dynamic_lds(qkv_attn_kernel<WT, NT>, smem_max, lds_ok);
    hipLaunchKernelGGL((qkv_attn_kernel<WT, NT>), dim3(p.nseq * p.n_head), dim3(QA_THREADS), smem, stream, p);
    if (getenv("CLIP_AMD_DEBUG_LAUNCH")) {
        const hipError_t e = hipPeekAtLastError();
        if (e != hipSuccess) fprintf(stderr, "qkv_attn_kernel<%d,%d>: grid %d smem %zu (max %zu): %s\n", (int)WT, NT, p.nseq * p.n_head, smem, smem_max, hipGetErrorString(e));
    }
}

template <int WT>
void launch_qa_nt(const QkvAttnParams & p, hipStream_t stream) {
    const int nt = (p.max_len + 15) / 16;
    if (nt <= 1) launch_qa<WT, 1>(p, stream);
    else if (nt <= 2) launch_qa<WT, 2>(p, stream);
    else if (nt <= 3) launch_qa<WT, 3>(p, stream);
    else launch_qa<WT, 4>(p, stream);
}

}  // namespace

#ifdef CLIPAMD_QA_WT
#define CLIPAMD_QCAT2(a, b) a##b
#define CLIPAMD_QCAT(a, b) CLIPAMD_QCAT2(a, b)
void CLIPAMD_QCAT(launch_qkv_attn_wt, CLIPAMD_QA_WT)(const QkvAttnParams & p, hipStream_t stream) { launch_qa_nt<CLIPAMD_QA_WT>(p, stream); }
#else
void launch_qkv_attn_wt0(const QkvAttnParams &, hipStream_t);
void launch_qkv_attn_wt1(const QkvAttnParams &, hipStream_t);
void launch_qkv_attn_wt2(const QkvAttnParams &, hipStream_t);
void launch_qkv_attn_wt3(const QkvAttnParams &, hipStream_t);
void launch_qkv_attn_wt4(const QkvAttnParams &, hipStream_t);
void launch_qkv_attn_wt5(const QkvAttnParams &, hipStream_t);

bool qkv_attn_supported(const QkvAttnParams & p) {
    return p.nseq > 0 && p.max_len > 0 && p.max_len <= 64 && p.n_head > 0 && p.h == p.n_head * 64 && p.W.N == 3 * p.h && p.W.K == p.W.Kpad &&
           p.W.K == p.h && p.W.K <= 1024 && p.W.K % 32 == 0 && p.ldx % 4 == 0 && p.ln_w && p.ln_b && p.stats_in && p.stats_slots > 0 && p.x32 && p.qkv && p.out;
}

void launch_qkv_attn(const QkvAttnParams & p, hipStream_t stream) {
    switch (p.W.wtype) {
    case W_F16: launch_qkv_attn_wt0(p, stream); break;
    case W_Q4_0: launch_qkv_attn_wt1(p, stream); break;
    case W_Q4_1: launch_qkv_attn_wt2(p, stream); break;
    case W_Q5_0: launch_qkv_attn_wt3(p, stream); break;
    case W_Q5_1: launch_qkv_attn_wt4(p, stream); break;
    case W_Q8_0: launch_qkv_attn_wt5(p, stream); break;
    }
}
#endif

}  // namespace clipamd
