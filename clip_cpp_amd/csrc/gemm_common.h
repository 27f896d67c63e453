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
        h2 v;
#ifdef CLIPAMD_DEQUANT_R5
        constexpr bool inplace = false;       // A/B: every pair through its own shift
#else
        constexpr bool inplace = WT == W_Q4_0 || WT == W_Q4_1;
#endif
        if constexpr (inplace) {
            // round 6: the odd pairs (nibbles at bits 4-7 / 20-23 of the word shifted by 0 or 8) are taken IN PLACE — 0x6400 | (q << 4) is 1024 + 16 q, and
            // fma(., 1/16, -(64 + zero)) = q - zero exactly — so a word costs one shift (by 8, shared by its pairs 2 and 3) instead of three:
            // 13 VALU instructions per 8 weights instead of 15, same bits (every intermediate is an exact small integer)
            const uint32_t w = s < 2 ? f.q : f.q >> 8;
            // (x & mask) | magic as ONE v_and_or_b32: a VOP3 instruction of gfx9 cannot carry a literal, so with compile-time constants hipcc emits
            // v_and_b32 + v_or_b32 (two VOP2 with a literal each: 64 of the 136 dequantisation instructions of a K-step pair of the 160 x 128 kernel);
            // made opaque — the mask in an SGPR, the magic in a VGPR, hoisted out of the loop as any loop-invariant value — it folds into one
            uint32_t magic = 0x64006400u, mask = (s & 1) ? 0x00F000F0u : 0x000F000Fu;
#ifndef CLIPAMD_DEQUANT_R5
            asm("" : "+v"(magic));
            asm("" : "+s"(mask));
#endif
            const uint32_t u = (w & mask) | magic;
            if (s & 1) {
                const float z = WT == W_Q4_0 ? 72.0f : 64.0f;
                v = __builtin_elementwise_fma(u2h(u), splat(0.0625f), splat(-z));
            } else {
                v = u2h(u) - sub;  // exact small integer
            }
        } else {
            uint32_t u = ((f.q >> (4 * s)) & 0x000F000Fu) | 0x64006400u;
            if constexpr (WT == W_Q5_0 || WT == W_Q5_1) u |= (hb >> s) & 0x00100010u;
            v = u2h(u) - sub;  // exact small integer
        }
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
// Round 6: the constants of the exponent are folded by hand (the build has no fast-math: __expf(a x) was v_mul (a x), v_mul (log2 e), v_exp) and the
// last two operations are one FMA — 7 / 5 VALU instructions per output instead of 10 / 6: the FFN-up epilogue is VALU-issue-bound work next to a
// co-resident K loop.  -DCLIPAMD_GELU_R5: the old expressions.
//
// GELU_SCALAR_FENCE — a correctness fence, not tuning.  With the two-FMA ln_apply below, hipcc's SLP vectoriser packed the whole strip
// (v_pk_fma_f32 -> v_exp_f32 / v_rcp_f32 -> v_pk_mul_f32 on the reciprocals, one independent instruction after the second v_rcp_f32), and on
// gfx950 / ROCm 7.2 that code returned wrong values on single quarter-waves (16 rows x 1 column of a fragment) a few dozen times per 2 M
// outputs, differently from run to run: the text tower's FFN-up at 1027 rows showed it first (profiles/r06_experiments.txt section 8:
// the same source with -fno-slp-vectorize, or with this fence, is bit-stable; barriers, s_sleep and s_nop around the epilogue are not a cure,
// a plain v_exp_f32 whose source register the next VALU overwrites is fine — scripts/ubench/trans_war.hip).  The empty asm pins the
// reciprocal and x as scalar values, so no packed instruction consumes a transcendental's result within the hazard recogniser's one wait
// state; tests/test_gpu_kernels.py::test_activation_epilogues_are_bit_stable_run_to_run holds every kernel family to it.
#define GELU_SCALAR_FENCE(t, x) asm volatile("" : "+v"(t), "+v"(x))
__device__ __forceinline__ float gelu_tanh(float x) {
    // ggml_gelu_f32: 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)));  0.5 (1 + tanh(u)) = 1 - 1/(exp(2u) + 1)
#ifdef CLIPAMD_GELU_R5
    const float u = 0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x);
    const float e = __expf(2.0f * u);
    return x - x * __builtin_amdgcn_rcpf(e + 1.0f);
#else
    constexpr float C1 = 2.0f * 0.79788456080286535587989211986876f * 1.44269504088896340736f;   // exp(2u) = exp2(x (C1 + C2 x^2))
    constexpr float C2 = C1 * 0.044715f;
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, C2, C1));
    float t = __builtin_amdgcn_rcpf(e + 1.0f);
    GELU_SCALAR_FENCE(t, x);
    float r = __builtin_fmaf(-x, t, x);
    // ... and the result: where this FMA's only use is the fp16 conversion, hipcc fuses the two into v_fma_mixlo_f16 (ONE rounding) in some
    // instantiations and not in others (57 / 25 of the ring kernel's 128- / 64-wide tiles): kernels then differ by an fp16 ulp on a few
    // outputs per million.  Pinned, every kernel rounds to f32 and then to fp16.
    asm volatile("" : "+v"(r));
    return r;
#endif
}
__device__ __forceinline__ float gelu_quick(float x) {
#ifdef CLIPAMD_GELU_R5
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
#else
    float t = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.44269504088896340736f)));
    GELU_SCALAR_FENCE(t, x);
    return x * t;
#endif
}

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs around it (kernels.h GemmParams::ln_* / xg_*).
//   W LN(x) + b  =  rstd (W (x gamma) - mean c) + b'        c_n = sum_k gamma_k W_nk,  b'_n = sum_k beta_k W_nk + b_n
// so the LayerNorm LAUNCH disappears: the residual epilogue that produces x also emits fp16(x gamma) — the next GEMM's A operand —
// and per-row partial statistics; the consuming GEMM's epilogue applies mean / rstd.  K loops, tiles and weight planes are untouched.
//
// Statistics format: slot t of row m = (s, q) = (sum, sum of squared deviations from the slot's own mean) over columns
// [t w, (t + 1) w) of the f32 row: within a slot two passes in registers, across slots Chan's pairwise update — no E[x^2] - mean^2
// cancellation anywhere.  All merges run in a fixed order (deterministic, and independent of tile shapes: the unit every kernel
// reduces is the 32 columns of two accumulator strips in the same lane order; a 64-column slot is the Chan merge of its two halves,
// which is also what the consumer does with a pair of 32-column slots).
//
// Where the work sits (r03a measured the first version — statistics reduced in the consumer's EPILOGUE by the 4 lanes that share a
// row, with ds_bpermute exchanges and IEEE divisions — at +15-21 us per GEMM launch, more than the LayerNorm launch it replaced):
//   * consumer: in the kernel PROLOGUE thread t < BM reduces row m0 + t to (mean, rstd) in two registers (ln_row_final: coalesced
//     8-byte loads, all of a row's slots in flight, v_rcp instead of divisions) while the first K-tiles are in flight; after the K loop
//     the values are exchanged through the then idle LDS (ln_rows_exchange) — nothing is added to the epilogue's memory round trip;
//   * producer: the 4-lane reductions use v_permlane16_swap / v_permlane32_swap (VALU, gfx950) instead of ds_bpermute — the LDS pipe
//     belongs to the co-resident workgroup's K loop — and xg is staged through LDS so that every global store writes full 128-byte lines.
// ---------------------------------------------------------------------------------------------
// Chan merge of the two 32-column halves of a 64-column slot: (mean, sum of squared deviations) of the 64 columns.  Shared by the
// producer (64-column kernels) and the consumer (32-column slots arrive in pairs), so both see the same bits.
__device__ __forceinline__ void ln_pair32(float s0, float q0, float s1, float q1, float & mean, float & m2) {
    const float m0 = s0 * (1.0f / 32.0f), m1 = s1 * (1.0f / 32.0f);
    const float d = m1 - m0;
    mean = m0 + d * 0.5f;
    m2 = q0 + (q1 + (d * d) * 16.0f);
}

// value + value of lane ^ 16, then + lane ^ 32: the sum over the 4 lanes (fgrp = 0..3) that hold one accumulator row.  gfx950 VALU
// swaps: v_permlane16_swap exchanges the odd 16-lane rows of the first operand with the even rows of the second, v_permlane32_swap
// the upper half-wave of the first with the lower half-wave of the second; with both operands = s the two results add up to the
// xor-16 / xor-32 butterfly in every lane (inline asm: the builtin form miscompiled the second result in ROCm 7.2).
__device__ __forceinline__ float sum4_fgrp(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    s = a + b;
    a = s; b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// Consumer prologue: (mean, rstd) of row m from its partial statistics, sequential over the units of 64 columns (a pair of 32-column
// slots merged first, or one slot of any width) in a fixed order.  Called by thread t < BM for row m0 + t: a wave's loads are
// coalesced.  Straight-line code: chunks of 16 units with every load of a chunk in flight, merge coefficients 1 / (t + 1) and
// t / (t + 1) folded at compile time, units past the end masked by zero coefficients (r03b: the first version, with a scalar branch
// per unit and a v_rcp per merge, cost the q/k/v and FFN-up GEMMs 7-9 us per launch).
template <int C0>
__device__ __forceinline__ void ln_row_chunk(const float2 * st, size_t stride, int units, bool pairs, float w, float invw, float & mean, float & m2) {
    constexpr int CH = 16;
    float mu[CH], q[CH];
    if (pairs) {
        float2 v0[CH], v1[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int u = C0 + j < units ? C0 + j : units - 1;
            v0[j] = st[(size_t)(2 * u) * stride];
            v1[j] = st[(size_t)(2 * u + 1) * stride];
        }
#pragma unroll
        for (int j = 0; j < CH; j++) ln_pair32(v0[j].x, v0[j].y, v1[j].x, v1[j].y, mu[j], q[j]);
    } else {
        float2 v[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int u = C0 + j < units ? C0 + j : units - 1;
            v[j] = st[(size_t)u * stride];
        }
#pragma unroll
        for (int j = 0; j < CH; j++) { mu[j] = v[j].x * invw; q[j] = v[j].y; }
    }
#pragma unroll
    for (int j = 0; j < CH; j++) {
        constexpr float one = 1.0f;
        const int t = C0 + j;
        const float ok = t < units ? one : 0.0f;                      // (wave-uniform: a scalar select)
        const float k1 = ok * (one / (float)(t + 1));
        const float k2 = ok * (w * ((float)t / (float)(t + 1)));
        const float d = mu[j] - mean;
        mean = mean + d * k1;
        m2 = m2 + ok * q[j] + (d * d) * k2;
    }
}

__device__ __forceinline__ float2 ln_row_final(const GemmParams & p, int m) {
    const bool pairs = p.ln_slotw == 32;
    const int units = pairs ? p.ln_slots >> 1 : p.ln_slots;
    const float w = pairs ? 64.f : (float)p.ln_slotw, invw = 1.0f / w;
    const float2 * st = p.ln_stats + m;
    const size_t stride = (size_t)p.ln_stride;
    float mean = 0.f, m2 = 0.f;
    ln_row_chunk<0>(st, stride, units, pairs, w, invw, mean, m2);
    if (units > 16) ln_row_chunk<16>(st, stride, units, pairs, w, invw, mean, m2);      // hidden sizes up to 2048 = 32 units
    const float invh = 1.0f / ((float)p.ln_slots * (float)p.ln_slotw);
    return make_float2(mean, 1.0f / sqrtf(m2 * invh + p.ln_eps));
}

// ln_row_final + the centring of the operand (GemmParams::ln_mu / mu_out): the epilogue's "mean" becomes mean - mu_m, and the workgroups
// of the first column tile (`writer`) leave the true mean for the next producer.  Rows past M arrive clamped to M - 1 (same value).
__device__ __forceinline__ float2 ln_row_centred(const GemmParams & p, int m, bool writer) {
    const float mu = p.ln_mu ? p.ln_mu[m] : 0.f;          // in flight with the statistics loads
    float2 r = ln_row_final(p, m);
    if (writer && p.mu_out) p.mu_out[m] = r.x;
    r.x -= mu;
    return r;
}

// Consumer, after the K loop: thread t < BM parks its row's (mean, rstd) in LDS (the tile buffers are idle: the caller's barrier
// `sync` separates the last fragment reads from these writes); the epilogues read the TM rows of a lane's accumulators from there
// (ln_rows_read).  rs: LDS area of BM float2 that does not overlap the fp16 staging areas of the epilogue.
template <typename SYNC>
__device__ __forceinline__ void ln_rows_publish(float2 * rs, float2 mine, int tid, int BM, SYNC sync) {
    sync();
    if (tid < BM) rs[tid] = mine;
    sync();
}
// rs_lane = rs + (first row of the wave's sub-tile) + frow
template <int TM>
__device__ __forceinline__ void ln_rows_read(float2 (&mr)[TM], bool ln, const float2 * rs_lane) {
#pragma unroll
    for (int b = 0; b < TM; b++) mr[b] = ln ? rs_lane[b * 16] : make_float2(0.f, 1.f);
}

// Residual epilogue, producer half of the fold.  acc holds the NEW residual rows (resid + acc + bias, already stored as f32).
// Writes the partial statistics of the wave's sub-tile — one slot per SW = 64 columns (32 when the wave spans only 32: the ring kernel
// and the BN = 64 tiles; gemm_fold_slotw() tells the consumer which) — and xg = fp16(x gamma_next), the next GEMM's A operand.
// stage: this wave's private LDS staging area ((TM * 16) rows x 68 halfs, idle tile buffers) or nullptr.  Staged, every global store
// instruction writes 8 full 128-byte lines; unstaged (ring kernel, 64-row tiles) a store covers 16 rows x 32 bytes.
template <int TN> constexpr int fold_slotw() { return TN * 16 >= 64 ? 64 : 32; }

template <int TN, int TM>
__device__ __forceinline__ void resid_fold_tail(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp,
                                                half_t * stage, int lane) {
    constexpr int SW = fold_slotw<TN>(), SA = SW / 16;         // strips (16 columns each) per slot
    static_assert(TN % SA == 0, "a wave's columns are whole statistics slots");
    const int N = p.W.N;
    const bool staged = SA == 4 && stage != nullptr && (p.ldxg & 7) == 0;
    // centring offset of row m (0 = the uncentred form: same bits as r03).  Fetched per row right where it is used — an L1 / L2 hit after the
    // first slot — instead of TM values held across the tail: the 192-row tile has no registers left for them (it spilled with an array)
    auto mu_of = [&](int b) {
        const int m = mbase + b * 16 + frow;
        return p.xg_mu ? p.xg_mu[m < p.M ? m : p.M - 1] : 0.f;
    };
    // one statistics slot (SA strips) at a time: its gamma values, its statistics, its xg columns (a 128 x 128 wave sub-tile of
    // k_gemm4.hip would otherwise hold 8 strips of gamma beside its accumulators)
#pragma unroll
    for (int sp = 0; sp < TN / SA; sp++) {
        const int n = nbase + sp * SW;
        if (n >= N) continue;                                  // (uniform; N is a multiple of 64 on this path: the loader checks)
        f4 gam[SA];
#pragma unroll
        for (int i = 0; i < SA; i++) gam[i] = *(const f4 *)(p.xg_gamma + n + i * 16 + fgrp * 4);
#pragma unroll
        for (int b = 0; b < TM; b++) {
            // canonical unit: 32 columns = 2 strips x 4 columns in this lane x the 4 lanes (fgrp) that share the row, two passes
            float s32[SA / 2], q32[SA / 2];
#pragma unroll
            for (int i = 0; i < SA / 2; i++) {
                const f4 u = acc[sp * SA + 2 * i][b], v = acc[sp * SA + 2 * i + 1][b];
                const float s = sum4_fgrp(((u[0] + u[1]) + (u[2] + u[3])) + ((v[0] + v[1]) + (v[2] + v[3])));
                const float mean = s * (1.0f / 32.0f);
                const f4 du = u - mean, dv = v - mean;
                const float q = sum4_fgrp(((du[0] * du[0] + du[1] * du[1]) + (du[2] * du[2] + du[3] * du[3])) +
                                          ((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])));
                s32[i] = s; q32[i] = q;
            }
            float2 o;
            if constexpr (SW == 64) {       // the 64-column slot = Chan merge of its two halves (what the consumer does with 32-column slots)
                float mean64, q64;
                ln_pair32(s32[0], q32[0], s32[1], q32[1], mean64, q64);
                o = make_float2(mean64 * 64.f, q64);
            } else {
                o = make_float2(s32[0], q32[0]);
            }
            const int m = mbase + b * 16 + frow;
            if (fgrp == 0 && m < p.M) p.stats_out[(size_t)(n / SW) * p.stats_stride + m] = o;
        }
        if constexpr (SA == 4) {
            if (staged) {
                // 64 columns through the wave's staging rows (136-byte pitch: conflict-free 8-byte writes), re-read row-contiguous:
                // every global store instruction then writes 8 full 128-byte lines
                constexpr int RS = 68;
                const int rrow = lane >> 3, rchunk = lane & 7;
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    const float mub = mu_of(b);
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const f4 g = (acc[sp * 4 + a][b] - mub) * gam[a];
                        const h2 lo = (h2){(_Float16)g[0], (_Float16)g[1]};
                        const h2 hi = (h2){(_Float16)g[2], (_Float16)g[3]};
                        *(uint2 *)(stage + (b * 16 + frow) * RS + a * 16 + fgrp * 4) = make_uint2(h2u(lo), h2u(hi));
                    }
                }
#pragma unroll
                for (int i = 0; i < TM * 2; i++) {
                    const int ml = i * 8 + rrow;
                    const int m = mbase + ml;
                    const u32x4 v = *(const u32x4 *)(stage + ml * RS + rchunk * 8);
                    if (m < p.M) *(u32x4 *)(p.xg_out + (size_t)m * p.ldxg + n + rchunk * 8) = v;
                }
                continue;
            }
        }
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = mbase + b * 16 + frow;
            if (m >= p.M) continue;
            const float mub = mu_of(b);
#pragma unroll
            for (int i = 0; i < SA; i++) {
                const f4 g = (acc[sp * SA + i][b] - mub) * gam[i];
                const h2 lo = (h2){(_Float16)g[0], (_Float16)g[1]};
                const h2 hi = (h2){(_Float16)g[2], (_Float16)g[3]};
                *(uint2 *)(p.xg_out + (size_t)m * p.ldxg + n + i * 16 + fgrp * 4) = make_uint2(h2u(lo), h2u(hi));
            }
        }
    }
}

// consumer half: v = acc + bias, or with the fold rstd_m (acc - mean_m c_n) + b'_n
// Round 6: two FMAs per output — fma(acc, rstd q, fma(-c, mean rstd q, b' q)) — instead of mul, sub, mul, add and a multiply + select per output for the
// Q scale q (clip.cpp:1363: after the bias; q = qscale on the Q columns, else 1: decided per 4-column strip, not per output).  The build runs
// -ffp-contract=off, so the FMAs are spelled out: the fp16-output GEMMs of a ViT-B/32 batch spent ~10 % of their VALU instructions on the old forms
// (+0.9 % on the two-tower step, profiles/r06_experiments.txt section 8).  Same algebra, fewer roundings; every kernel (k_skinny.hip restates it) uses
// this form, so outputs stay bit-identical across kernels and tiles.  Without the fold: mr = (0, 1), c = 0 -> fma(acc, q, b q).
__device__ __forceinline__ f4 ln_apply(const float2 & mr, float q, const f4 & acc, const f4 & c, const f4 & bias) {
#ifdef CLIPAMD_LNAPPLY_R5              // A/B: the round-5 arithmetic (sub, mul, add, then the Q scale)
    return ((acc - c * mr.x) * mr.y + bias) * q;
#else
    const float rq = mr.y * q, mq = mr.x * rq;
    f4 v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = __builtin_fmaf(acc[r], rq, __builtin_fmaf(-c[r], mq, bias[r] * q));
    return v;
#endif
}
// the Q scale of the 4 columns n .. n + 3 of an EPI_F16 strip (qcols is a multiple of 4 wherever it is set: h)
template <int EPI>
__device__ __forceinline__ float strip_qscale(const GemmParams & p, int n) { return (EPI == EPI_F16 && n < p.qcols) ? p.qscale : 1.0f; }

// ---- epilogue.  D[i][j]: i = weight row n (row = 4*(lane>>4)+reg), j = activation row m (col = lane&15).
// nbase / mbase: first weight row / activation row of this wave's sub-tile.
// FOLD = false compiles the LayerNorm-fold branches out (k_gemm4.hip: with 256 accumulator registers per lane the fold tail spills, and
// a kernel that has its CU to itself exposes the tail anyway — forward.cpp fold_pays keeps such shapes on the LayerNorm launches)
template <int EPI, int TN, int TM, bool FOLD = true>
__device__ __forceinline__ void gemm_epilogue(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp,
                                              bool ln_on, const float2 * rs_lane, half_t * stage = nullptr, int lane = 0) {
    const int N = p.W.N;
    constexpr bool LNE = FOLD && (EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16);   // epilogues that can consume a folded LayerNorm
    const bool ln = LNE && ln_on;      // rs_lane: (mean, rstd) of this lane's rows in LDS (ln_row_final + ln_rows_publish in the kernel)
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
    float2 mr[LNE ? TM : 1];
    if constexpr (LNE) {
        ln_rows_read<TM>(mr, ln, rs_lane);
        if (ln) {          // same batch of loads as the bias vectors: one memory round trip
#pragma unroll
            for (int a = 0; a < TN; a++) {
                int n = nbase + a * 16 + fgrp * 4;
                n = n < N ? n : 0;
                cv[a] = *(const f4 *)(p.ln_c + n);
            }
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
        if constexpr (FOLD) {
            if (p.xg_out) resid_fold_tail<TN, TM>(p, acc, nbase, mbase, frow, fgrp, stage, lane);
        }
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
            if constexpr (LNE) v = ln_apply(mr[b], strip_qscale<EPI>(p, n), acc[a][b], ln ? cv[a] : (f4){0.f, 0.f, 0.f, 0.f}, bias);
            else if constexpr (EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16)      // FOLD = false (k_gemm_f32.hip): same expression, no statistics
                v = ln_apply(make_float2(0.f, 1.f), strip_qscale<EPI>(p, n), acc[a][b], (f4){0.f, 0.f, 0.f, 0.f}, bias);
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
                if constexpr (EPI == EPI_GELU_F16) {
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

// ---- residual epilogue with the residual fragments already in registers (k_gemm8.hip, round 6: requested before the K loop, so the tile's f32 rows —
// 40 % of the epilogue's memory traffic — cross the fabric under the MFMAs instead of after them).  Same expressions as gemm_epilogue: same bits.
template <int TN, int TM>
__device__ __forceinline__ void gemm_epilogue_resid_pre(const GemmParams & p, f4 (&acc)[TN][TM], const f4 (&rpre)[TN][TM], int nbase, int mbase, int frow, int fgrp,
                                                        half_t * stage, int lane) {
    const int N = p.W.N;
    __builtin_amdgcn_s_waitcnt(0x0070);                // (see gemm_epilogue)
    f4 biasv[TN];
#pragma unroll
    for (int a = 0; a < TN; a++) {
        int n = nbase + a * 16 + fgrp * 4;
        n = n < N ? n : 0;
        biasv[a] = p.bias ? *(const f4 *)(p.bias + n) : (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        if (n >= N) continue;
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = mbase + b * 16 + frow;
            acc[a][b] = rpre[a][b] + (acc[a][b] + biasv[a]);
            if (m < p.M) *(f4 *)((float *)p.out + (size_t)m * p.ldc + n) = acc[a][b];
        }
    }
    if (p.xg_out) resid_fold_tail<TN, TM>(p, acc, nbase, mbase, frow, fgrp, stage, lane);
}

// ---- the same epilogue with its operands already in registers (k_gemm_ring.hip): the bias vectors — and, for the residual epilogue,
// the residual fragments — were requested before the K loop, so the tail of a workgroup that has its CU to itself does not wait out
// a memory round trip per strip (measured there: 7 us of a 60 us kernel at 64 x 256 tiles).  Same expressions, same rounding.
template <int EPI, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue_pre(const GemmParams & p, f4 (&acc)[TN][TM], const f4 (&biasv)[TN], const f4 (&rpre)[TN][TM],
                                                  int nbase, int mbase, int frow, int fgrp, bool ln, const f4 (&cv)[TN], const float2 * rs_lane) {
    const int N = p.W.N;
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;
    float2 mr[LNE ? TM : 1];
    if constexpr (LNE) ln_rows_read<TM>(mr, ln, rs_lane);
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
            resid_fold_tail<TN, TM>(p, acc, nbase, mbase, frow, fgrp, nullptr, 0);
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
            if constexpr (LNE) v = ln_apply(mr[b], strip_qscale<EPI>(p, n), acc[a][b], ln ? cv[a] : (f4){0.f, 0.f, 0.f, 0.f}, bias);
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
                if constexpr (EPI == EPI_GELU_F16) {
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
template <int EPI, int TN, int TM, bool FOLD = true>
__device__ __forceinline__ void gemm_epilogue_f16_staged(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp,
                                                         half_t * stage, int lane, bool ln_on, const float2 * rs_lane) {
    const bool ln = FOLD && ln_on;
    constexpr int RS = 68;                                     // halfs per staged row (64 + 4 pad = 136 B)
    __builtin_amdgcn_s_waitcnt(0x0070);                        // (see gemm_epilogue: lets hipcc count its waits again)
    f4 biasv[TN];                                              // one batch of loads, not TN dependent round trips
#pragma unroll
    for (int a = 0; a < TN; a++) biasv[a] = p.bias ? *(const f4 *)(p.bias + nbase + a * 16 + fgrp * 4) : (f4){0.f, 0.f, 0.f, 0.f};
    f4 cv[TN];                                                 // ln: LayerNorm folded into this GEMM, rs_lane = (mean, rstd) of this lane's rows (LDS)
    float2 mr[TM];
    ln_rows_read<TM>(mr, ln, rs_lane);
    if (ln) {
#pragma unroll
        for (int a = 0; a < TN; a++) cv[a] = *(const f4 *)(p.ln_c + nbase + a * 16 + fgrp * 4);
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        const f4 bias = biasv[a];
#pragma unroll
        for (int b = 0; b < TM; b++) {
            f4 v = ln_apply(mr[b], strip_qscale<EPI>(p, n), acc[a][b], ln ? cv[a] : (f4){0.f, 0.f, 0.f, 0.f}, bias);
            if constexpr (EPI == EPI_GELU_F16) {
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
