// k_skinny.hip — the small-M ("skinny") weight GEMM of the hot path for gfx950: up to 64 rows — one ViT-B/32 image (50 token
// rows), one text of up to 64 tokens, a few short labels.
//
//   out[M][N] = epilogue( A[M][K] · W[N][K]^T + bias ),   M <= 64,   A = fp16 rows  OR  LayerNorm(x) computed on the fly
//
// At this size a forward pass is a chain of ~90 dependent kernels that each move a few hundred KB: what counts is the LATENCY
// of a kernel, not its throughput (r01: 10-14 us per GEMM launch, 0.64 ms per ViT-B/32 image = 1.1 % of the HBM roofline).
// The tiled kernels (k_gemm.hip) pay per launch: LDS staging + a barrier per 64-wide K-step (12-48 dependent steps), a
// split-K ticket hand-off through L2, and LayerNorm as a separate launch in front of every q/k/v and FFN-up projection.
// Here the structure is that of a decode GEMV (cdna_hip_programming.md §5 "GEMV / M <= 16": no LDS round trip, deep unroll):
//
//   * one workgroup per 16 output columns x 16 rows (N / 16 x ceil(M / 16) workgroups: 192-768 for one ViT-B/32 image; grid.x =
//     column group, a multiple of 8 for every CLIP width, so the row blocks of a column group share one XCD's L2 copy of the
//     weight slab); its NW waves split K into NW contiguous ranges — an intra-workgroup split-K that is reduced through LDS in
//     wave order (deterministic).  (First form, r02: all <= 64 rows in one workgroup per column group — 48 workgroups for the
//     N = 768 GEMMs, 13 us per launch; with the rows over grid.y: one image 0.572 -> 0.430 ms.)
//   * weights never touch LDS: a lane loads ONE 32-bit word of packed quants (+ the block scale) per 16 x 32 MFMA A-fragment
//     straight from the block-column-major planes (16 rows x 16 B = 256 contiguous bytes per wave instruction) and
//     dequantises it in registers (dequant_wfrag: the same packed-fp16 arithmetic as the tiled kernels); fp16 weights are read
//     as fragments from their row-major image;
//   * activations are read as MFMA B-fragments from L2 (fp16 rows), or — LayerNorm fused — as f32 rows of the residual
//     stream that are normalised in registers ((x - mean) * rstd * gamma + beta, rounded to fp16 exactly where the
//     LayerNorm kernel rounds).  Row statistics come from the PRODUCER of x: the residual epilogue of the previous skinny
//     GEMM (or the pre-LayerNorm / embedding kernel) leaves per-row partial (sum, sum of squares) over its 16 columns in
//     slot blockIdx.x; the consumer adds the slots in order — no atomics, bit-reproducible;
//   * loads are issued in chunks two deep (next chunk in flight while the current one is multiplied), so a workgroup costs
//     about one memory round trip + <= 24 MFMAs + the LDS reduction.
//   A transformer layer at M <= 64 is then 5 launches (LN1+QKV, attention, out-proj+residual, LN2+FFN-up+GELU, FFN-down+residual)
//   instead of 7, each a few microseconds.  (Fusing the first two — LN1 + q/k/v + attention of a (sequence, head) in one workgroup — was
//   built and measured SLOWER: 23 us against 10.5 + 6.6 for one image; profiles/r02_second_session_experiments.txt section 6.)
//
// Numerics: fp32 accumulation split in NW partial sums per output (fixed order) and LayerNorm variance as E[x^2] - mean^2 —
// the same class of fp32 re-association as the split-K path this replaces (tests: batch-1 vs batch-N rows agree to 1e-6 in
// cosine); parity against the oracle is tested end to end and per kernel (tests/test_gpu_kernels.py).
// Reference ops replaced: ggml_mul_mat + ggml_norm / mul / add chains of clip.cpp:1350-1423 (vision), :1064-1143 (text).

#include "gemm_common.h"

namespace clipamd {

namespace {

__device__ __forceinline__ h8 normalise8(const f4 lo, const f4 hi, float mean, float rstd, const f4 g0, const f4 g1, const f4 b0, const f4 b1) {
    const f4 y0 = ((lo - mean) * rstd) * g0 + b0;     // same expression order as layernorm_kernel: (v * scale) * w + b
    const f4 y1 = ((hi - mean) * rstd) * g1 + b1;
    return (h8){(_Float16)y0[0], (_Float16)y0[1], (_Float16)y0[2], (_Float16)y0[3], (_Float16)y1[0], (_Float16)y1[1], (_Float16)y1[2], (_Float16)y1[3]};
}

template <int WT, int MF, int NW, int EPI, bool LNA>
__global__ void __launch_bounds__(NW * 64) skinny_kernel(const SkinnyParams p) {
    constexpr int CH = LNA ? 3 : (MF == 1 ? 6 : 4);   // k-blocks per chunk; two chunks in flight = 6 (f32 rows) / 8-12 (fp16 rows) k-blocks
                                                      // of loads per lane: the whole K range of a wave at K = 768 / 1024 (and at K = 3072
                                                      // with 8 waves and one fragment row)
    __shared__ f4 red[NW * MF * 64];
    constexpr bool FOLDC = !LNA && (EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16);   // can consume a folded LayerNorm (p.ln_c)
    __shared__ float2 lnst[(LNA || FOLDC) ? 128 : 1];
    __shared__ float lng[LNA ? 2048 : 1], lnb[LNA ? 2048 : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m_base = blockIdx.y * (MF * 16);        // grid.y splits the rows: with MF = 1 a workgroup owns ONE 16 x 16 output fragment
    const int nkb = p.W.Kpad / 32;
    const int kb_lo = wave * nkb / NW, kb_hi = (wave + 1) * nkb / NW;

    int mrow[MF];
#pragma unroll
    for (int b = 0; b < MF; b++) { const int m = m_base + b * 16 + frow; mrow[b] = m < p.M ? m : p.M - 1; }
#ifdef CLIPAMD_SK_TIMING   // tuning builds: phase stamps (shader clock; [6]/[7] the 100 MHz real-time clock) of the first and the last workgroup
    const bool sk_first = blockIdx.x == 0 && blockIdx.y == 0, sk_last = blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;
    const bool stamper = p.stamps && tid == 0 && (sk_first || sk_last);
    unsigned long long * stamp = p.stamps + (sk_first ? 0 : 8);
#define SK_STAMP(i_) if (stamper) stamp[i_] = __builtin_amdgcn_s_memtime()
    if (stamper) stamp[6] = __builtin_amdgcn_s_memrealtime();
#else
#define SK_STAMP(i_)
#endif
    SK_STAMP(0);

    // ---- operand loaders (one k-block = one 16 x 32 weight fragment + MF 16 x 32 activation fragments)
    struct WReg { WFrag<WT> q; h8 h; };
    auto load_w = [&](WReg & w, int kb) {
        if constexpr (WT == W_F16) {
            w.h = *(const h8 *)((const half_t *)p.W.w16 + (size_t)(n0 + frow) * p.W.Kpad + kb * 32 + fgrp * 8);
        } else {
            const size_t idx = (size_t)kb * p.W.Npad + n0 + frow;
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
    struct XReg { h8 h; f4 lo, hi; };
    auto load_x = [&](XReg & x, int kb, int b) {
        if constexpr (LNA) {
            const float * src = p.x32 + (size_t)mrow[b] * p.ldx + kb * 32 + fgrp * 8;
            x.lo = *(const f4 *)src;
            x.hi = *(const f4 *)(src + 4);
        } else {
            x.h = *(const h8 *)(p.A16 + (size_t)mrow[b] * p.lda + kb * 32 + fgrp * 8);
        }
    };

    WReg wA[CH], wB[CH];
    XReg xA[CH][MF], xB[CH][MF];
#define SK_LOAD(W_, X_, c_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < CH; i++)                                             \
        if ((c_) + i < kb_hi) {                                                                \
            load_w(W_[i], (c_) + i);                                                           \
            _Pragma("unroll") for (int b = 0; b < MF; b++) load_x(X_[i][b], (c_) + i, b);      \
        }
    // first chunk goes out before anything else: the LayerNorm prologue below runs under its latency
    SK_LOAD(wA, xA, kb_lo);
    // so do the epilogue's operands (bias, residual) of the fragment row this wave will finalise: nothing in the tail of the
    // kernel waits for memory any more
    static_assert(MF <= NW, "each wave finalises at most one fragment row");
    f4 bias_pre = (f4){0.f, 0.f, 0.f, 0.f}, resid_pre = (f4){0.f, 0.f, 0.f, 0.f}, c_pre = (f4){0.f, 0.f, 0.f, 0.f}, gam_pre = (f4){0.f, 0.f, 0.f, 0.f};
    const bool foldc = FOLDC && p.ln_c != nullptr;
    float mu_pre = 0.f;                                // producer: the offset this lane's new operand row is centred on (SkinnyParams::xg_mu)
    {
        const int n = n0 + fgrp * 4, m = m_base + wave * 16 + frow;
        if constexpr (EPI == EPI_RESID_F32) { if (p.xg_out && p.xg_mu && wave < MF) mu_pre = p.xg_mu[m < p.M ? m : p.M - 1]; }
        if (EPI != EPI_PATCH_F32 && p.bias && n < p.W.N) bias_pre = *(const f4 *)(p.bias + n);
        if constexpr (FOLDC) { if (foldc && n < p.W.N) c_pre = *(const f4 *)(p.ln_c + n); }
        if constexpr (EPI == EPI_RESID_F32) { if (p.xg_out && n < p.W.N) gam_pre = *(const f4 *)(p.xg_gamma + n); }
        if constexpr (EPI == EPI_RESID_F32) {
            if (wave < MF && m < p.M && n < p.W.N) resid_pre = *(const f4 *)(p.resid + (size_t)m * p.ldc + n);
        }
    }

    SK_STAMP(1);           // first chunk + epilogue operands requested
    float mean[MF], rstd[MF];
    if constexpr (LNA) {
        // gamma / beta of the whole row into LDS once (one 16-byte load per thread and array for K <= 1024)
        for (int k4 = tid; k4 < p.W.K / 4; k4 += NW * 64) {
            *(f4 *)(lng + 4 * k4) = *(const f4 *)(p.ln_w + 4 * k4);
            *(f4 *)(lnb + 4 * k4) = *(const f4 *)(p.ln_b + 4 * k4);
        }
        // row statistics: TPR threads per row, each adding every TPR-th partial slot of its row (16 independent loads in flight
        // per round), then a butterfly over the TPR lanes: a fixed summation tree (bit-reproducible), one memory round trip
        {
            constexpr int TPR = (NW * 64) / (MF * 16);
            const int r = tid / TPR, sub = tid % TPR;          // r: row inside this workgroup's block of MF * 16 rows
            const int gr = m_base + r;
            const float2 * row = p.stats_in + (size_t)(gr < p.M ? gr : p.M - 1) * p.stats_cap;
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
            if (sub == 0) {                                   // (rows past M hold the statistics of the clamped row M - 1, like their data)
                const float mu = s1 / (float)p.W.K;
                float var = s2 / (float)p.W.K - mu * mu;
                var = var > 0.f ? var : 0.f;
                lnst[r] = make_float2(mu, 1.0f / sqrtf(var + p.eps));
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < MF; b++) { const float2 st = lnst[b * 16 + frow]; mean[b] = st.x; rstd[b] = st.y; }
    }

    if constexpr (FOLDC) {
        if (foldc) {
            // LayerNorm folded into this GEMM: (mean, rstd) of the workgroup's 16 rows from the partial statistics — 16 threads per row,
            // thread `sub` takes slots sub, sub + 16, ... (all in flight with the operand loads above), Chan merges in a fixed order:
            // sequential inside a thread, then a butterfly over the 16 lanes, lower lane first.  Read back in the epilogue, behind the
            // barrier of the partial-sum exchange.
            constexpr int TPR = (NW * 64) / (MF * 16);
            static_assert(TPR == 16 || TPR == 32, "16 or 32 threads per row");
            const int r = tid / TPR, sub = tid % TPR;
            const int gr = m_base + r;
            const float2 * st = p.fstats + (gr < p.M ? gr : p.M - 1);
            const float mu_in = (sub == 0 && p.ln_mu) ? p.ln_mu[gr < p.M ? gr : p.M - 1] : 0.f;      // offset the operand was centred on
            const float w = (float)p.fslotw, invw = 1.0f / w;
            float n_ = 0.f, mean_ = 0.f, m2_ = 0.f;
            for (int base = 0; base < p.fslots; base += TPR * 8) {
                float2 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int sl = base + i * TPR + sub;
                    v[i] = st[(size_t)(sl < p.fslots ? sl : p.fslots - 1) * p.fstride];
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (base + i * TPR + sub < p.fslots) {
                        const float nn = n_ + w, d = v[i].x * invw - mean_, inv = __builtin_amdgcn_rcpf(nn);
                        mean_ += d * (w * inv);
                        m2_ += v[i].y + (d * d) * (n_ * w * inv);
                        n_ = nn;
                    }
                }
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) {
                const float nb = __shfl_xor(n_, o), mb = __shfl_xor(mean_, o), qb = __shfl_xor(m2_, o);
                const bool upper = (sub & o) != 0;
                const float n0_ = upper ? nb : n_, m0_ = upper ? mb : mean_, q0_ = upper ? qb : m2_;      // the lower lane's aggregate first
                const float n1_ = upper ? n_ : nb, m1_ = upper ? mean_ : mb, q1_ = upper ? m2_ : qb;
                const float nn = n0_ + n1_, d = m1_ - m0_, inv = __builtin_amdgcn_rcpf(nn > 0.f ? nn : 1.f);
                mean_ = m0_ + d * (n1_ * inv);
                m2_ = q0_ + q1_ + (d * d) * (n0_ * n1_ * inv);
                n_ = nn;
            }
            if (sub == 0) {
                if (p.mu_out && blockIdx.x == 0 && gr < p.M) p.mu_out[gr] = mean_;       // this LayerNorm's row mean, for the next producer
                lnst[r] = make_float2(mean_ - mu_in, 1.0f / sqrtf(m2_ / n_ + p.eps));
            }
        }
    }
    SK_STAMP(2);           // LayerNorm prologue done (statistics round trip + barrier)
    f4 acc[MF];
#pragma unroll
    for (int b = 0; b < MF; b++) acc[b] = (f4){0.f, 0.f, 0.f, 0.f};
#define SK_COMPUTE(W_, X_, c_)                                                                 \
    _Pragma("unroll") for (int i = 0; i < CH; i++)                                             \
        if ((c_) + i < kb_hi) {                                                                \
            h8 wf;                                                                             \
            if constexpr (WT == W_F16) wf = W_[i].h; else wf = dequant_wfrag<WT>(W_[i].q, fgrp); \
            f4 g0, g1, b0, b1;                                                                 \
            if constexpr (LNA) {                                                               \
                const int k0 = ((c_) + i) * 32 + fgrp * 8;                                     \
                g0 = *(const f4 *)(lng + k0); g1 = *(const f4 *)(lng + k0 + 4);                \
                b0 = *(const f4 *)(lnb + k0); b1 = *(const f4 *)(lnb + k0 + 4);                \
            }                                                                                  \
            _Pragma("unroll") for (int b = 0; b < MF; b++) {                                   \
                h8 xf;                                                                         \
                if constexpr (LNA) xf = normalise8(X_[i][b].lo, X_[i][b].hi, mean[b], rstd[b], g0, g1, b0, b1); \
                else xf = X_[i][b].h;                                                          \
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[b], 0, 0, 0);      \
            }                                                                                  \
        }
    for (int c = kb_lo; c < kb_hi; c += 2 * CH) {
        SK_LOAD(wB, xB, c + CH);
        SK_COMPUTE(wA, xA, c);
        SK_LOAD(wA, xA, c + 2 * CH);
        SK_COMPUTE(wB, xB, c + CH);
    }
#undef SK_LOAD
#undef SK_COMPUTE

    SK_STAMP(3);           // all MFMAs issued (the loads of every chunk have landed)
    // ---- intra-workgroup split-K: partial fragments through LDS, summed in wave order
#pragma unroll
    for (int b = 0; b < MF; b++) red[(wave * MF + b) * 64 + lane] = acc[b];
    __syncthreads();
    SK_STAMP(4);           // partial sums exchanged
    for (int b = wave; b < MF; b += NW) {
        f4 v = red[b * 64 + lane];
        for (int w = 1; w < NW; w++) v = v + red[(w * MF + b) * 64 + lane];
        const int n = n0 + fgrp * 4;
        const int m = m_base + b * 16 + frow;
        const bool ok = m < p.M && n < p.W.N;
        {
            // bias / LayerNorm-fold consumer / Q scale exactly as the tiled kernels (gemm_common.h ln_apply, round 6: two FMAs per output, the Q scale
            // folded into the coefficients of its 4-column strip); a wave finalises at most the one fragment row b == wave (MF <= NW)
            const f4 bb = b == wave ? bias_pre : (f4){0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16) {
                float2 st = make_float2(0.f, 1.f);
                f4 cc = (f4){0.f, 0.f, 0.f, 0.f};
                if constexpr (FOLDC) {
                    if (foldc) { st = lnst[b * 16 + frow]; cc = c_pre; }     // (mean, rstd) of this lane's row, written before the barrier above
                }
                v = ln_apply(st, (EPI == EPI_F16 && n < p.qcols) ? p.qscale : 1.0f, v, cc, bb);
            } else {
                v = v + bb;
            }
        }
        if constexpr (EPI == EPI_F32) {
            if (ok) *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = v;
        } else if constexpr (EPI == EPI_RESID_F32) {
            f4 o = (f4){0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const f4 r = resid_pre;
                o = r + v;
                *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = o;
            }
            if (p.xg_out) {      // LayerNorm fold, producer half: the next GEMM's operand fp16(x gamma_next) and the statistics of this
                                 // workgroup's 16 columns (sum, sum of squared deviations from their mean: two passes in registers)
                if (ok) {
                    const f4 g = (o - mu_pre) * gam_pre;
                    const h2 glo = (h2){(_Float16)g[0], (_Float16)g[1]}, ghi = (h2){(_Float16)g[2], (_Float16)g[3]};
                    *(uint2 *)(p.xg_out + (size_t)m * p.ldxg + n) = make_uint2(h2u(glo), h2u(ghi));
                }
                const float s = sum4_fgrp((o[0] + o[1]) + (o[2] + o[3]));
                const f4 dd = o - s * (1.0f / 16.0f);
                const float q = sum4_fgrp((dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]));
                if (fgrp == 0 && m < p.M) p.fstats_out[(size_t)blockIdx.x * p.fstride_out + m] = make_float2(s, q);
            }
            if (p.stats_out) {   // partial LayerNorm statistics of the NEW residual row over this workgroup's 16 columns
                float s1 = (o[0] + o[1]) + (o[2] + o[3]);
                float s2 = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (fgrp == 0 && m < p.M && (int)blockIdx.x < p.stats_cap) p.stats_out[(size_t)m * p.stats_cap + blockIdx.x] = make_float2(s1, s2);   // [row][slot]
            }
        } else if constexpr (EPI == EPI_PATCH_F32) {
            if (ok) {
                const int img = m / p.Np, pp = m % p.Np;
                const f4 pe = *(const f4 *)(p.pos + (size_t)(1 + pp) * p.ldc + n);
                *(f4 *)((float *)p.out + ((size_t)img * p.T + 1 + pp) * p.ldc + n) = v + pe;
            }
        } else {
            if constexpr (EPI == EPI_F16) {
            } else if constexpr (EPI == EPI_GELU_F16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_tanh(v[r]);
            } else if constexpr (EPI == EPI_QGELU_F16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_quick(v[r]);
            }
            const h2 lo = (h2){(_Float16)v[0], (_Float16)v[1]};
            const h2 hi = (h2){(_Float16)v[2], (_Float16)v[3]};
            if (ok) *(uint2 *)((half_t *)p.out + (size_t)m * p.ldc + n) = make_uint2(h2u(lo), h2u(hi));
        }
    }
#ifdef CLIPAMD_SK_TIMING
    SK_STAMP(5);           // stores issued
    if (stamper) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef SK_STAMP
}

template <int WT, int MF, int NW, int EPI, bool LNA>
void launch_sk(const SkinnyParams & p, hipStream_t stream) {
    // grid.x = column groups (a multiple of 8 for every CLIP width: the row blocks of one column group land on one XCD and share its
    // L2 copy of the weight slab), grid.y = row blocks
    hipLaunchKernelGGL((skinny_kernel<WT, MF, NW, EPI, LNA>), dim3((p.W.N + 15) / 16, (p.M + MF * 16 - 1) / (MF * 16)), dim3(NW * 64), 0, stream, p);
}

template <int WT, int MF>
void launch_sk_epi(const SkinnyParams & p, int epi, hipStream_t stream) {
    const bool lna = p.x32 != nullptr;
    const bool long_k = p.W.Kpad >= 2048;
    if (lna) {          // LayerNorm-fused A: the q/k/v and FFN-up projections (K = hidden size <= 2048)
        switch (epi) {
        case EPI_F16: launch_sk<WT, MF, 4, EPI_F16, true>(p, stream); break;
        case EPI_GELU_F16: launch_sk<WT, MF, 4, EPI_GELU_F16, true>(p, stream); break;
        case EPI_QGELU_F16: launch_sk<WT, MF, 4, EPI_QGELU_F16, true>(p, stream); break;
        default: break;
        }
        return;
    }
    switch (epi) {
    case EPI_F32: launch_sk<WT, MF, 4, EPI_F32, false>(p, stream); break;
    case EPI_F16: launch_sk<WT, MF, 4, EPI_F16, false>(p, stream); break;               // fp16 A + fp16 epilogues: the consumers of the LayerNorm fold
    case EPI_GELU_F16: launch_sk<WT, MF, 4, EPI_GELU_F16, false>(p, stream); break;
    case EPI_QGELU_F16: launch_sk<WT, MF, 4, EPI_QGELU_F16, false>(p, stream); break;
    case EPI_RESID_F32:
        if (long_k) launch_sk<WT, MF, 8, EPI_RESID_F32, false>(p, stream);
        else launch_sk<WT, MF, 4, EPI_RESID_F32, false>(p, stream);
        break;
    case EPI_PATCH_F32:
        if constexpr (WT == W_F16) {
            if (long_k) launch_sk<WT, MF, 8, EPI_PATCH_F32, false>(p, stream);
            else launch_sk<WT, MF, 4, EPI_PATCH_F32, false>(p, stream);
        }
        break;
    default: break;
    }
}

}  // namespace

#ifdef CLIPAMD_SKINNY_WT
#define CLIPAMD_SCAT2(a, b) a##b
#define CLIPAMD_SCAT(a, b) CLIPAMD_SCAT2(a, b)
void CLIPAMD_SCAT(launch_skinny_wt, CLIPAMD_SKINNY_WT)(const SkinnyParams & p, int epilogue, hipStream_t stream) {
    // one 16-row fragment per workgroup: N / 16 x ceil(M / 16) workgroups (192-768 for one ViT-B/32 image instead of 48-192), each a
    // single memory round trip + <= 12 MFMAs per wave; weights are re-read by the row blocks from L2
    launch_sk_epi<CLIPAMD_SKINNY_WT, 1>(p, epilogue, stream);
}
#else
void launch_skinny_wt0(const SkinnyParams &, int, hipStream_t);
void launch_skinny_wt1(const SkinnyParams &, int, hipStream_t);
void launch_skinny_wt2(const SkinnyParams &, int, hipStream_t);
void launch_skinny_wt3(const SkinnyParams &, int, hipStream_t);
void launch_skinny_wt4(const SkinnyParams &, int, hipStream_t);
void launch_skinny_wt5(const SkinnyParams &, int, hipStream_t);

// which (epilogue, operand) combinations the skinny path covers; everything else stays on launch_gemm
bool skinny_supported(const SkinnyParams & p, int epilogue) {
    if (p.M <= 0 || p.M > SKINNY_MAX_ROWS || p.W.N % 4 || p.ldc % 4 || p.W.wtype == W_F32) return false;   // (f32 weights: k_gemm_f32.hip)
    if (p.x32) {
        if (epilogue != EPI_F16 && epilogue != EPI_GELU_F16 && epilogue != EPI_QGELU_F16) return false;
        return p.W.K == p.W.Kpad && p.W.K <= 2048 && p.ldx % 4 == 0 && p.ln_w && p.ln_b && p.stats_in && p.stats_slots > 0;
    }
    if (epilogue == EPI_PATCH_F32) return p.W.wtype == W_F16 && p.lda % 8 == 0;
    if (epilogue == EPI_F16 || epilogue == EPI_GELU_F16 || epilogue == EPI_QGELU_F16)      // fold consumer: statistics + c vector required
        return p.lda % 8 == 0 && p.ln_c && p.fstats && p.fslots > 0 && p.fslotw > 0 && p.M <= 128;
    return (epilogue == EPI_F32 || epilogue == EPI_RESID_F32) && p.lda % 8 == 0;
}

void launch_skinny(const SkinnyParams & p, int epilogue, hipStream_t stream) {
    switch (p.W.wtype) {
    case W_F16: launch_skinny_wt0(p, epilogue, stream); break;
    case W_Q4_0: launch_skinny_wt1(p, epilogue, stream); break;
    case W_Q4_1: launch_skinny_wt2(p, epilogue, stream); break;
    case W_Q5_0: launch_skinny_wt3(p, epilogue, stream); break;
    case W_Q5_1: launch_skinny_wt4(p, epilogue, stream); break;
    case W_Q8_0: launch_skinny_wt5(p, epilogue, stream); break;
    }
}

// Partial-statistics form of a plain row pass (slot 0 only): sum and sum of squares of every row of x [rows][h] — the LayerNorm
// statistics of the FIRST layer's input (the later ones come out of the residual epilogues above).
__global__ void __launch_bounds__(256) row_stats_kernel(const float * __restrict__ x, int ldx, int rows, int h, float2 * __restrict__ stats, int cap) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float * xr = x + (size_t)r * ldx;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < h; c += 256) {
        const f4 v = *(const f4 *)(xr + c);
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) stats[(size_t)r * cap] = make_float2(s1, s2);      // slot 0 of row r in the [row][slot] layout
}

void launch_row_stats(const float * x, int ldx, int rows, int h, float2 * stats, hipStream_t stream) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(row_stats_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, rows, h, stats, 128);
}
#endif

}  // namespace clipamd
