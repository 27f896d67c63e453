// gemm_common.h — device helpers shared by the GEMM kernels (k_gemm.hip: 4-wave fused dequant GEMM, small / medium M;
// k_gemm8.hip: 8-wave ping-pong GEMM for large M + the weight dequantisation kernel): packed-fp16 dequantisation of the
// block-quantised weight planes, activation functions, and the epilogues (bias / Q-scale / GELU / residual / patch scatter).
#pragma once

#include "kernels.h"

namespace clipamd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 64;          // K step per iteration (two 32-wide quant blocks)

__device__ __forceinline__ h2 u2h(uint32_t u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ uint32_t h2u(h2 h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ h2 splat(float x) { return (h2){(_Float16)x, (_Float16)x}; }

// offset (in halfs) of 16-byte chunk c (0..7) of row r in a swizzled [rows][64] fp16 LDS tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * BK + ((c ^ (r & 7)) << 3); }

// ---------------------------------------------------------------------------------------------
// Quantised weights never touch LDS: every wave loads, per 16x32 MFMA A-fragment, ONE 32-bit word of
// packed quants per lane (plus the block scale) straight from the block-column-major planes and
// dequantises it in registers into the 8 fp16 values the MFMA wants.
//
// Packed nibble layout (load.cpp repack_rows): 32-bit word j of a block holds elements 8j..8j+7;
// nibble p<4 is element 2p, nibble p>=4 is element 2(p-4)+1.  Hence
//   ((w >> 4s) & 0x000F000F) = { lo half: element 2s, hi half: element 2s+1 }  (adjacent pair)
// and OR-ing 0x6400 into each half gives the fp16 number 1024+q exactly.  Lane (row = lane&15,
// k-group g = lane>>4) of the MFMA A operand owns elements 8g..8g+7 of the block = word g.
// q5: fifth bits in a parallel word: bit pi = 4j+s -> element 2s of word j, bit 16+pi -> element 2s+1.
// q8_0: bytes stored as (int8 ^ 0x80) in the order [e0,e2,e1,e3] per word: (w & 0x00FF00FF) = {e0,e1},
//   ((w >> 8) & 0x00FF00FF) = {e2,e3}; 0x6400|u8 = 1024 + (q+128); lane g owns words 2g, 2g+1.
// ---------------------------------------------------------------------------------------------
template <int WT> struct WFrag;
template <> struct WFrag<W_F16> { uint32_t q; };   // unused
template <> struct WFrag<W_Q4_0> { uint32_t q; half_t d; };
template <> struct WFrag<W_Q4_1> { uint32_t q; h2 dm; };
template <> struct WFrag<W_Q5_0> { uint32_t q, h; half_t d; };
template <> struct WFrag<W_Q5_1> { uint32_t q, h; h2 dm; };
template <> struct WFrag<W_Q8_0> { uint32_t q, q1; half_t d; };

// One pair of weights (output word s = 0..3 of the fragment: elements 2s, 2s+1 of the lane's 8) — the unit the ring kernel
// (k_gemm_ring.hip) slots between its MFMAs; dequant_wfrag below is the four of them, so every kernel computes the same bits.
// hb: fifth-bit word of the lane's k-group (dequant_hbits), q5 only.
template <int WT>
__device__ __forceinline__ uint32_t dequant_hbits(const WFrag<WT> & f, int g) {
    if constexpr (WT == W_Q5_0 || WT == W_Q5_1) return (uint32_t)(((uint64_t)f.h << 4) >> (4 * g));   // pair bits of word g at 4+s / 20+s
    else return 0u;
}
template <int WT>
__device__ __forceinline__ uint32_t dequant_wpair(const WFrag<WT> & f, uint32_t hb, int s) {
    if constexpr (WT == W_F16) return f.q;
    else if constexpr (WT == W_Q8_0) {
        const h2 scale = (h2){f.d, f.d};
        const h2 sub = splat(1152.0f);
        const uint32_t w = s < 2 ? f.q : f.q1;
        const uint32_t u = ((s & 1) ? (w >> 8) : w) & 0x00FF00FFu;
        return h2u((u2h(u | 0x64006400u) - sub) * scale);
    } else {
        h2 scale, sub, add;
        if constexpr (WT == W_Q4_0) { scale = (h2){f.d, f.d}; sub = splat(1032.0f); }
        if constexpr (WT == W_Q5_0) { scale = (h2){f.d, f.d}; sub = splat(1040.0f); }
        if constexpr (WT == W_Q4_1 || WT == W_Q5_1) {
            scale = (h2){f.dm[0], f.dm[0]};
            add = (h2){f.dm[1], f.dm[1]};
            sub = splat(1024.0f);
        }
        uint32_t u = ((f.q >> (4 * s)) & 0x000F000Fu) | 0x64006400u;
        if constexpr (WT == W_Q5_0 || WT == W_Q5_1) u |= (hb >> s) & 0x00100010u;
        h2 v = u2h(u) - sub;  // exact small integer
        if constexpr (WT == W_Q4_1 || WT == W_Q5_1) v = __builtin_elementwise_fma(v, scale, add);  // q*d + m, one rounding
        else v = v * scale;
        return h2u(v);
    }
}

template <int WT>
__device__ __forceinline__ h8 dequant_wfrag(const WFrag<WT> & f, int g) {
    if constexpr (WT == W_F16) return __builtin_bit_cast(h8, (u32x4){f.q, 0u, 0u, 0u});
    else {
        const uint32_t hb = dequant_hbits<WT>(f, g);
        return __builtin_bit_cast(h8, (u32x4){dequant_wpair<WT>(f, hb, 0), dequant_wpair<WT>(f, hb, 1), dequant_wpair<WT>(f, hb, 2), dequant_wpair<WT>(f, hb, 3)});
    }
}

// One whole 32-weight block per thread (LDS-staged path): the 4 (q8_0: 8) packed words + fifth bits + scale.
template <int WT> struct RawBlock { u32x4 qs, qs1; uint32_t h; half_t d; h2 dm; };

template <int WT>
__device__ __forceinline__ void load_block(RawBlock<WT> & r, const DevWeight & W, size_t idx) {
    if constexpr (WT == W_Q8_0) {
        const u32x4 * q = (const u32x4 *)W.qs + idx * 2;
        r.qs = q[0];
        r.qs1 = q[1];
    } else if constexpr (WT != W_F16) {
        r.qs = ((const u32x4 *)W.qs)[idx];
    }
    if constexpr (WT == W_Q5_0 || WT == W_Q5_1) r.h = ((const uint32_t *)W.qh)[idx];
    if constexpr (WT == W_Q4_1 || WT == W_Q5_1) r.dm = ((const h2 *)W.dm)[idx];
    else if constexpr (WT != W_F16) r.d = ((const half_t *)W.dm)[idx];
}

// word j (8 weights) of a block as a register fragment
template <int WT>
__device__ __forceinline__ WFrag<WT> block_word(const RawBlock<WT> & r, int j) {
    WFrag<WT> f;
    if constexpr (WT == W_Q8_0) {
        const uint32_t w[8] = {r.qs[0], r.qs[1], r.qs[2], r.qs[3], r.qs1[0], r.qs1[1], r.qs1[2], r.qs1[3]};
        f.q = w[2 * j];
        f.q1 = w[2 * j + 1];
    } else {
        f.q = r.qs[j];
    }
    if constexpr (WT == W_Q5_0 || WT == W_Q5_1) f.h = r.h;
    if constexpr (WT == W_Q4_1 || WT == W_Q5_1) f.dm = r.dm;
    else if constexpr (WT != W_F16) f.d = r.d;
    return f;
}

// Activations of the FFN-up epilogue.  The result is rounded to fp16 right after, so the reciprocal is the hardware
// v_rcp_f32 (1 ulp) instead of an IEEE division sequence (~10 instructions per element; at 96 outputs per thread the
// epilogue was ~6 % of a K = 768 tile's time).  The reference evaluates both through fp16 lookup tables.
__device__ __forceinline__ float gelu_tanh(float x) {
    // ggml_gelu_f32: 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)));  0.5 (1 + tanh(u)) = 1 - 1/(exp(2u) + 1)
    const float u = 0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x);
    const float e = __expf(2.0f * u);
    return x - x * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float gelu_quick(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs around it (kernels.h GemmParams::ln_* / xg_*).
//   W LN(x) + b  =  rstd (W (x gamma) - mean c) + b'        c_n = sum_k gamma_k W_nk,  b'_n = sum_k beta_k W_nk + b_n
// so the LayerNorm LAUNCH disappears: the residual epilogue that produces x also emits fp16(x gamma) — the next GEMM's A operand —
// and per-row partial statistics; the consuming GEMM's epilogue applies mean / rstd.  K loops, tiles and weight planes are untouched.
//
// Statistics format: slot t of row m = (s, q) = (sum, sum of squared deviations from the slot's own mean) over columns
// [t w, (t + 1) w) of the f32 row: within a slot two passes in registers, across slots Chan's pairwise update — no E[x^2] - mean^2
// cancellation anywhere.  All merges run in a fixed order (deterministic, independent of tile shape: a slot is always reduced
// from the f32 values of its own columns in the same lane order; the slot WIDTH depends on the producing kernel, see fold_slotw).
// ---------------------------------------------------------------------------------------------
template <int TM> struct LnRows { float mu[TM], rstd[TM]; };

struct LnAgg { float n, mean, m2; };
__device__ __forceinline__ void ln_merge(LnAgg & a, float nb, float meanb, float m2b) {
    const float n = a.n + nb;
    const float d = meanb - a.mean;
    const float inv = 1.0f / (n > 0.f ? n : 1.f);
    a.mean += d * (nb * inv);
    a.m2 += m2b + d * d * (a.n * nb * inv);
    a.n = n;
}

// (mean, rstd) of the rows this lane's accumulators belong to (row mbase + b * 16 + frow, b < TM).  The 4 lanes that share a row
// (fgrp = 0..3) each reduce units fgrp, fgrp + 4, ... of the row — a unit is one slot, or with 32-column slots the PAIR (2t, 2t + 1)
// merged first, which reproduces the 64-column slot of the wider kernels bit for bit, so (mean, rstd) do not depend on the kernel that
// produced the statistics — TM x ceil(units / 4) x (1 or 2) loads per lane, JC units per row in flight at a time, and combine with
// two xor-shuffle rounds in lane order.
template <int TM, int JC = (TM <= 4 ? 8 : 4)>     // JC x TM (x 2) float2 registers in flight
__device__ __forceinline__ void ln_rows_load(LnRows<TM> & L, const GemmParams & p, int mbase, int frow, int fgrp) {
    const bool pairs = p.ln_slotw == 32;
    const int units = pairs ? p.ln_slots >> 1 : p.ln_slots;
    const float w = pairs ? 64.f : (float)p.ln_slotw, invw = 1.0f / w;
    LnAgg ag[TM];
    int mrow[TM];
#pragma unroll
    for (int b = 0; b < TM; b++) {
        ag[b].n = 0.f; ag[b].mean = 0.f; ag[b].m2 = 0.f;
        const int m = mbase + b * 16 + frow;
        mrow[b] = m < p.M ? m : p.M - 1;
    }
    for (int j0 = 0; j0 < units; j0 += 4 * JC) {
        if (pairs) {
            float2 v0[TM][JC], v1[TM][JC];
#pragma unroll
            for (int j = 0; j < JC; j++) {
                const int u = j0 + 4 * j + fgrp;
                const int uc = u < units ? u : units - 1;
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    v0[b][j] = p.ln_stats[(size_t)(2 * uc) * p.ln_stride + mrow[b]];
                    v1[b][j] = p.ln_stats[(size_t)(2 * uc + 1) * p.ln_stride + mrow[b]];
                }
            }
#pragma unroll
            for (int j = 0; j < JC; j++) {
                const bool ok = j0 + 4 * j + fgrp < units;
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    LnAgg pr;
                    pr.n = 32.f; pr.mean = v0[b][j].x * (1.0f / 32.0f); pr.m2 = v0[b][j].y;
                    ln_merge(pr, 32.f, v1[b][j].x * (1.0f / 32.0f), v1[b][j].y);
                    if (ok) ln_merge(ag[b], 64.f, pr.mean, pr.m2);
                }
            }
        } else {
            float2 v[TM][JC];
#pragma unroll
            for (int j = 0; j < JC; j++) {
                const int u = j0 + 4 * j + fgrp;
                const int uc = u < units ? u : units - 1;
#pragma unroll
                for (int b = 0; b < TM; b++) v[b][j] = p.ln_stats[(size_t)uc * p.ln_stride + mrow[b]];
            }
#pragma unroll
            for (int j = 0; j < JC; j++) {
                const bool ok = j0 + 4 * j + fgrp < units;
#pragma unroll
                for (int b = 0; b < TM; b++)
                    if (ok) ln_merge(ag[b], w, v[b][j].x * invw, v[b][j].y);
            }
        }
    }
    const float invh = 1.0f / ((float)p.ln_slots * (float)p.ln_slotw);
#pragma unroll
    for (int b = 0; b < TM; b++) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float nb = __shfl_xor(ag[b].n, o), mb_ = __shfl_xor(ag[b].mean, o), qb = __shfl_xor(ag[b].m2, o);
            // (both partners must end with the same bits: merge in lane order, the lower lane's aggregate first)
            LnAgg lo, hi;
            const bool upper = (fgrp * 16) & o;
            lo.n = upper ? nb : ag[b].n; lo.mean = upper ? mb_ : ag[b].mean; lo.m2 = upper ? qb : ag[b].m2;
            hi.n = upper ? ag[b].n : nb; hi.mean = upper ? ag[b].mean : mb_; hi.m2 = upper ? ag[b].m2 : qb;
            ln_merge(lo, hi.n, hi.mean, hi.m2);
            ag[b] = lo;
        }
        L.mu[b] = ag[b].mean;
        L.rstd[b] = 1.0f / sqrtf(ag[b].m2 * invh + p.ln_eps);
    }
}

// Residual epilogue, producer half of the fold.  acc holds the NEW residual rows (resid + acc + bias, already stored as f32).
// Writes the partial statistics of the wave's sub-tile — one slot per SW = 64 columns (32 when the wave spans only 32: the ring kernel
// and the BN = 64 tiles; gemm_fold_slotw() tells the consumer which) — and xg = fp16(x gamma_next), the next GEMM's A operand.
template <int TN> constexpr int fold_slotw() { return TN * 16 >= 64 ? 64 : 32; }

template <int TN, int TM>
__device__ __forceinline__ void resid_fold_tail(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp) {
    constexpr int SW = fold_slotw<TN>(), SA = SW / 16;         // strips (16 columns each) per slot
    static_assert(TN % SA == 0, "a wave's columns are whole statistics slots");
    const int N = p.W.N;
    f4 gam[TN];
#pragma unroll
    for (int a = 0; a < TN; a++) {
        int n = nbase + a * 16 + fgrp * 4;
        n = n < N ? n : 0;
        gam[a] = *(const f4 *)(p.xg_gamma + n);
    }
#pragma unroll
    for (int sp = 0; sp < TN / SA; sp++) {
        const int n = nbase + sp * SW;
        if (n >= N) continue;                                  // (uniform; N is a multiple of 64 on this path: launch_gemm checks)
#pragma unroll
        for (int b = 0; b < TM; b++) {
            // canonical unit: 32 columns = 2 strips x 4 columns in this lane x the 4 lanes (fgrp) that share the row, two passes
            float s32[SA / 2], q32[SA / 2];
#pragma unroll
            for (int i = 0; i < SA / 2; i++) {
                const f4 u = acc[sp * SA + 2 * i][b], v = acc[sp * SA + 2 * i + 1][b];
                float s = ((u[0] + u[1]) + (u[2] + u[3])) + ((v[0] + v[1]) + (v[2] + v[3]));
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mean = s * (1.0f / 32.0f);
                const f4 du = u - mean, dv = v - mean;
                float q = ((du[0] * du[0] + du[1] * du[1]) + (du[2] * du[2] + du[3] * du[3])) +
                          ((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]));
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                s32[i] = s; q32[i] = q;
            }
            float2 o;
            if constexpr (SW == 64) {       // the 64-column slot = Chan merge of its two halves (what the consumer does with 32-column slots)
                LnAgg pr;
                pr.n = 32.f; pr.mean = s32[0] * (1.0f / 32.0f); pr.m2 = q32[0];
                ln_merge(pr, 32.f, s32[1] * (1.0f / 32.0f), q32[1]);
                o = make_float2(pr.mean * 64.f, pr.m2);
            } else {
                o = make_float2(s32[0], q32[0]);
            }
            const int m = mbase + b * 16 + frow;
            if (fgrp == 0 && m < p.M) p.stats_out[(size_t)(n / SW) * p.stats_stride + m] = o;
        }
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        if (n >= N) continue;
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = mbase + b * 16 + frow;
            if (m >= p.M) continue;
            const f4 g = acc[a][b] * gam[a];
            const h2 lo = (h2){(_Float16)g[0], (_Float16)g[1]};
            const h2 hi = (h2){(_Float16)g[2], (_Float16)g[3]};
            *(uint2 *)(p.xg_out + (size_t)m * p.ldxg + n) = make_uint2(h2u(lo), h2u(hi));
        }
    }
}

// consumer half: v = acc + bias, or with the fold rstd_m (acc - mean_m c_n) + b'_n
template <int TM>
__device__ __forceinline__ f4 ln_apply(bool ln, const LnRows<TM> & L, int b, const f4 & acc, const f4 & c, const f4 & bias) {
    if (ln) return (acc - c * L.mu[b]) * L.rstd[b] + bias;
    return acc + bias;
}

// ---- epilogue.  D[i][j]: i = weight row n (row = 4*(lane>>4)+reg), j = activation row m (col = lane&15).
// nbase / mbase: first weight row / activation row of this wave's sub-tile.
template <int EPI, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp,
                                              const LnRows<TM> * ln_pre = nullptr) {
    const int N = p.W.N;
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;   // epilogues that can consume a folded LayerNorm
    const bool ln = LNE && p.ln_c != nullptr;
    // Everything the K loop requested has landed.  Said with the BUILTIN so that hipcc's waitcnt pass sees it: an LDS-DMA request
    // (a FLAT-encoded instruction touching two address spaces) leaves that pass in its "pending flat" state, in which every later
    // wait is vmcnt(0) / lgkmcnt(0) — in an epilogue that means each group of loads also waits for all earlier STORES to be acked.
    __builtin_amdgcn_s_waitcnt(0x0070);
    // all bias vectors of the wave's columns in one batch (one memory round trip instead of TN dependent ones)
    f4 biasv[TN];
#pragma unroll
    for (int a = 0; a < TN; a++) {
        int n = nbase + a * 16 + fgrp * 4;
        n = n < N ? n : 0;                             // (clamped: columns past N are never stored)
        biasv[a] = (EPI != EPI_PATCH_F32 && p.bias) ? *(const f4 *)(p.bias + n) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    f4 cv[LNE ? TN : 1];
    LnRows<TM> L;
    if constexpr (LNE) {
        if (ln) {          // same batch of loads as the bias vectors: one memory round trip
#pragma unroll
            for (int a = 0; a < TN; a++) {
                int n = nbase + a * 16 + fgrp * 4;
                n = n < N ? n : 0;
                cv[a] = *(const f4 *)(p.ln_c + n);
            }
            if (ln_pre) L = *ln_pre;
            else ln_rows_load<TM>(L, p, mbase, frow, fgrp);
        }
    }
    if constexpr (EPI == EPI_RESID_F32) {
        // Residual rows are fetched TM at a time with clamped (always valid) row indices, so the loads of one column strip are all
        // in flight together, and one strip AHEAD: the loads of strip a+1 are issued before strip a is added and stored (written as
        // load-add-store under an `m < M` branch each fragment waited out its own memory round trip: TN x TM dependent round trips
        // per tile, 15-44 % of the ViT-B/32 residual GEMMs; one strip at a time still left TN exposed round trips).
        f4 r[2][TM];
        auto fetch = [&](int a, f4 (&dst)[TM]) {
            int n = nbase + a * 16 + fgrp * 4;
            n = n < N ? n : 0;
#pragma unroll
            for (int b = 0; b < TM; b++) {
                const int m = mbase + b * 16 + frow;
                const int mc = m < p.M ? m : p.M - 1;
                dst[b] = *(const f4 *)(p.resid + (size_t)mc * p.ldc + n);
            }
        };
        fetch(0, r[0]);
#pragma unroll
        for (int a = 0; a < TN; a++) {
            if (a + 1 < TN) fetch(a + 1, r[(a + 1) & 1]);
            const int n = nbase + a * 16 + fgrp * 4;
            if (n >= N) continue;
#pragma unroll
            for (int b = 0; b < TM; b++) {
                const int m = mbase + b * 16 + frow;
                acc[a][b] = r[a & 1][b] + (acc[a][b] + biasv[a]);     // (kept: the fold tail below reads the new rows from acc)
                if (m < p.M) *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = acc[a][b];
            }
        }
        if (p.xg_out) resid_fold_tail<TN, TM>(p, acc, nbase, mbase, frow, fgrp);
        return;
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        if (n >= N) continue;
        const f4 bias = biasv[a];
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = mbase + b * 16 + frow;
            if (m >= p.M) continue;
            f4 v;
            if constexpr (LNE) v = ln_apply<TM>(ln, L, b, acc[a][b], cv[a], bias);
            else v = acc[a][b] + bias;
            if constexpr (EPI == EPI_F32) {
                *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = v;
            } else if constexpr (EPI == EPI_RESID_F32) {
                const f4 r = *(const f4 *)(p.resid + (size_t)m * p.ldc + n);
                *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = r + v;
            } else if constexpr (EPI == EPI_PATCH_F32) {
                const int img = m / p.Np, pp = m % p.Np;
                const f4 pe = *(const f4 *)(p.pos + (size_t)(1 + pp) * p.ldc + n);
                *(f4 *)((float *)p.out + ((size_t)img * p.T + 1 + pp) * p.ldc + n) = v + pe;
            } else {
                if constexpr (EPI == EPI_F16) {
                    if (n < p.qcols) v = v * p.qscale;
                } else if constexpr (EPI == EPI_GELU_F16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = gelu_tanh(v[r]);
                } else if constexpr (EPI == EPI_QGELU_F16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = gelu_quick(v[r]);
                }
                const h2 lo = (h2){(_Float16)v[0], (_Float16)v[1]};
                const h2 hi = (h2){(_Float16)v[2], (_Float16)v[3]};
                *(uint2 *)((half_t *)p.out + (size_t)m * p.ldc + n) = make_uint2(h2u(lo), h2u(hi));
            }
        }
    }
}

// ---- the same epilogue with its operands already in registers (k_gemm_ring.hip): the bias vectors — and, for the residual epilogue,
// the residual fragments — were requested before the K loop, so the tail of a workgroup that has its CU to itself does not wait out
// a memory round trip per strip (measured there: 7 us of a 60 us kernel at 64 x 256 tiles).  Same expressions, same rounding.
template <int EPI, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue_pre(const GemmParams & p, f4 (&acc)[TN][TM], const f4 (&biasv)[TN], const f4 (&rpre)[TN][TM],
                                                  int nbase, int mbase, int frow, int fgrp, bool ln, const f4 (&cv)[TN], const LnRows<TM> & L) {
    const int N = p.W.N;
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;
    if constexpr (EPI == EPI_RESID_F32) {
        if (p.xg_out) {          // producer half of the LayerNorm fold: every lane keeps its new rows (also past M: never stored)
#pragma unroll
            for (int a = 0; a < TN; a++) {
                const int n = nbase + a * 16 + fgrp * 4;
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    const int m = mbase + b * 16 + frow;
                    acc[a][b] = rpre[a][b] + (acc[a][b] + biasv[a]);
                    if (n < N && m < p.M) *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = acc[a][b];
                }
            }
            resid_fold_tail<TN, TM>(p, acc, nbase, mbase, frow, fgrp);
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        if (n >= N) continue;
        const f4 bias = biasv[a];
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = mbase + b * 16 + frow;
            if (m >= p.M) continue;
            f4 v;
            if constexpr (LNE) v = ln_apply<TM>(ln, L, b, acc[a][b], cv[a], bias);
            else v = acc[a][b] + bias;
            if constexpr (EPI == EPI_F32) {
                *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = v;
            } else if constexpr (EPI == EPI_RESID_F32) {
                *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = rpre[a][b] + v;
            } else if constexpr (EPI == EPI_PATCH_F32) {
                const int img = m / p.Np, pp = m % p.Np;
                const f4 pe = *(const f4 *)(p.pos + (size_t)(1 + pp) * p.ldc + n);
                *(f4 *)((float *)p.out + ((size_t)img * p.T + 1 + pp) * p.ldc + n) = v + pe;
            } else {
                if constexpr (EPI == EPI_F16) {
                    if (n < p.qcols) v = v * p.qscale;
                } else if constexpr (EPI == EPI_GELU_F16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = gelu_tanh(v[r]);
                } else if constexpr (EPI == EPI_QGELU_F16) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = gelu_quick(v[r]);
                }
                const h2 lo = (h2){(_Float16)v[0], (_Float16)v[1]};
                const h2 hi = (h2){(_Float16)v[2], (_Float16)v[3]};
                *(uint2 *)((half_t *)p.out + (size_t)m * p.ldc + n) = make_uint2(h2u(lo), h2u(hi));
            }
        }
    }
}

// ---- f16-output epilogue staged through LDS (qkv / FFN-up): written straight from the accumulator layout, a wave store
// covers 16 rows x 32 B — quarter cache lines, measured ~3 TB/s and 27-31 % of those GEMMs' time (profiles/).  Instead each
// wave parks its (BM/2) x (BN/2) fp16 sub-tile in its own LDS region (rows padded to 136 B: conflict-free 8-byte writes)
// and re-reads it row-contiguous, so every global store instruction writes 8 full 128-byte lines.
// Requires BN/2 == 64, the whole n range of the wave inside N, and a 16-byte aligned output row (ldc % 8 == 0).
template <int EPI, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue_f16_staged(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp,
                                                         half_t * stage, int lane, const LnRows<TM> * ln_pre = nullptr) {
    constexpr int RS = 68;                                     // halfs per staged row (64 + 4 pad = 136 B)
    __builtin_amdgcn_s_waitcnt(0x0070);                        // (see gemm_epilogue: lets hipcc count its waits again)
    f4 biasv[TN];                                              // one batch of loads, not TN dependent round trips
#pragma unroll
    for (int a = 0; a < TN; a++) biasv[a] = p.bias ? *(const f4 *)(p.bias + nbase + a * 16 + fgrp * 4) : (f4){0.f, 0.f, 0.f, 0.f};
    const bool ln = p.ln_c != nullptr;                         // LayerNorm folded into this GEMM (see ln_rows_load)
    f4 cv[TN];
    LnRows<TM> L;
    if (ln) {
#pragma unroll
        for (int a = 0; a < TN; a++) cv[a] = *(const f4 *)(p.ln_c + nbase + a * 16 + fgrp * 4);
        if (ln_pre) L = *ln_pre;
        else ln_rows_load<TM>(L, p, mbase, frow, fgrp);
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        const f4 bias = biasv[a];
#pragma unroll
        for (int b = 0; b < TM; b++) {
            f4 v = ln_apply<TM>(ln, L, b, acc[a][b], cv[a], bias);
            if constexpr (EPI == EPI_F16) {
                if (n < p.qcols) v = v * p.qscale;
            } else if constexpr (EPI == EPI_GELU_F16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_tanh(v[r]);
            } else if constexpr (EPI == EPI_QGELU_F16) {
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = gelu_quick(v[r]);
            }
            const h2 lo = (h2){(_Float16)v[0], (_Float16)v[1]};
            const h2 hi = (h2){(_Float16)v[2], (_Float16)v[3]};
            *(uint2 *)(stage + (b * 16 + frow) * RS + a * 16 + fgrp * 4) = make_uint2(h2u(lo), h2u(hi));
        }
    }
    // the region is private to this wave: no barrier, the LDS writes are ordered before the reads by lgkmcnt
    const int rrow = lane >> 3, rchunk = lane & 7;
#pragma unroll
    for (int i = 0; i < TM * 2; i++) {
        const int ml = i * 8 + rrow;
        const int m = mbase + ml;
        const u32x4 v = *(const u32x4 *)(stage + ml * RS + rchunk * 8);
        if (m < p.M) *(u32x4 *)((half_t *)p.out + (size_t)m * p.ldc + nbase + rchunk * 8) = v;
    }
}

}  // namespace

}  // namespace clipamd
