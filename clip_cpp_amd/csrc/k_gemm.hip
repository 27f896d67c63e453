// k_gemm.hip — fused dequant + MFMA GEMM for gfx950 (CDNA4).
//
//   out[M][N] = epilogue( X[M][K] (fp16) · W[N][K]^T (f16 | q4_0 | q4_1 | q5_0 | q5_1 | q8_0) + bias )
//
// This one kernel carries >= 96 % of the FLOPs of the hot path: the q/k/v/out projections, both FFN
// mat-muls, the patch-embedding contraction and the final projection — i.e. every ggml_mul_mat
// with a weight operand in reference clip.cpp:1360-1380,1392,1407,1416,1443 (vision) and
// :1079-1095,1112,1127,1136,1160 (text), plus ggml_conv_2d at :1309.
//
// Design (MI355X-first, not a port of ggml's vec_dot kernels):
//   * block-quantised weights stay quantised in HBM (4.5-8.5 bits per weight); a workgroup streams the
//     [BN rows x 2 blocks] slab of its tile with one coalesced 16 B load per lane (block-column-major planes,
//     kernels.h), dequantises it in registers with packed-fp16 VALU (magic-number 0x6400|q trick: and-or,
//     v_pk_add, v_pk_mul / v_pk_fma) and stages the fp16 tile in LDS (two register stages: the block of tile k+2
//     is in flight while tile k is multiplied and tile k+1 is dequantised);
//   * fp16 tiles (activations always, weights of f16 files) go L2 -> LDS by LDS-DMA (global_load_lds_dwordx4):
//     no VGPR round trip, no ds_write; the 16-byte-chunk XOR swizzle (chunk ^= row & 7, conflict-free ds_read_b128
//     fragment reads) is applied on the per-lane SOURCE address because the DMA image is lane-linear;
//   * 256 threads = 4 waves in a 2x2 grid; each wave owns a (BN/2)x(BM/2) sub-tile as 16x16 MFMA
//     fragments (v_mfma_f32_16x16x32_f16), fp32 accumulation in k order (deterministic); the weight is the MFMA
//     "A" operand so that each lane ends up with 4 consecutive output columns of one row -> 8/16-byte stores;
//   * two LDS buffers, one barrier per K-step; the barrier's vmcnt(0) is what lands the DMA, so nothing inside a
//     K-step waits on memory; every prefetch is unconditional (clamped tile index) so hipcc's s_waitcnt counting
//     stays exact;
//   * the K-step of the big tiles (BM >= 128) is written as an explicit instruction interleave: groups of two MFMAs
//     fenced by sched_barrier(0), with the fragment reads, the LDS-DMA pieces of the next tile and the dequantisation
//     + ds_write of the next weight tile placed between the groups in source order (STEP_ILV), so that VALU / LDS /
//     VMEM issue runs under the matrix pipe; hipcc's own scheduling (plain, sched_group_barrier pipelines,
//     iglp_opt) clumps that work after the MFMA chain (+4-7 % from the explicit interleave, profiles/);
//   * workgroup -> tile mapping: XCD-contiguous chunks, n fastest, so the workgroups that share an activation
//     row-panel run back to back on one XCD/L2 (measured: L2 hit rate 32 % -> 85 %, HBM traffic = algorithmic).
//
// Roofline: compute (MFMA fp16, 2.5 PFLOP/s dense) for M >= ~256; HBM (weight bytes) for M <= 64.
// Measured (r01, MI355X): 0.65-0.93 PFLOP/s on the ViT-B/32 / L/14 shapes (hipBLASLt, plain f16, no epilogue:
// 0.88-1.23).  Micro-benchmarks (scripts/*_bench.hip, profiles/): the matrix pipe alone 2.2-2.46 PFLOP/s, the bare
// LDS->register->MFMA loop of this tile structure 1.5-1.8, tile streaming L2->LDS ~20 TB/s and not latency-bound.
// The remaining step is a 256-wide register tile with a hand-scheduled (assembly) K loop.

#include <vector>

#include "gemm_common.h"

#ifndef CLIPAMD_TEST_HOOKS
#define CLIPAMD_TEST_HOOKS 1
#endif

namespace clipamd {

namespace {

constexpr int NTHREADS = 256;

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: fp16 tiles (X always, W when the weights are f16) go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write): one wave-instruction moves a 1 KB piece = 8 tile rows;
// the LDS image is lane-linear, so the 16-byte-chunk XOR swizzle is applied on the per-lane SOURCE address
// (lane l writes row l>>3, position l&7, and therefore fetches chunk (l&7)^(l>>3)); fragment reads use lds_off().
// Quantised weights: one raw 32-weight block per thread in registers (two stages), dequantised into LDS.
// Per K-step: issue DMA for tile k+1 + register loads for tile k+2, multiply tile k, dequant-store tile k+1, barrier
// (the barrier's vmcnt(0) is what lands the DMA, so nothing inside a step waits on memory).
// tiles instantiated with the deterministic split-K hand-off: the 64-row tiles.  (r03 also instantiated it for the 192 x 128 tile — VERDICT r2
// item 1 —: slower on every text-tower shape, and the extra code cost the split-free FFN-up launch 25 % (49.9 -> 62.6 us); removed again,
// the implementation is in commit 8f8234b, the numbers in profiles/r03_lnfold_and_text_tiles.txt section 6.)
constexpr bool gemm_tile_splits_k(int bm, int bn) { return bm == 64 && bn > 0; }

// ---------------------------------------------------------------------------------------------
template <int WT, int BM, int BN, int EPI>
__global__ void __launch_bounds__(NTHREADS, ((BM + BN) * BK * 4 <= 80 * 1024 ? 2 : 1)) gemm_dma_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t * Xs = (half_t *)smem_raw;                 // [2][BM*BK]
    half_t * Ws = Xs + 2 * BM * BK;                   // [2][BN*BK]
    constexpr int TN = BN / 32, TM = BM / 32;
    constexpr int XPW = BM / 32;                      // 1 KB pieces of the X tile per wave
    constexpr int WPW = BN / 32;                      // (f16 weights) pieces of the W tile per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, fgrp = lane >> 4;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    constexpr bool SK = gemm_tile_splits_k(BM, BN);
    const int nwg = tiles_m * tiles_n * (SK ? p.ksplit : 1);
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // split-K (BM = 64 tiles only, small-M problems): ksplit consecutive workgroups share one output tile, each
    // multiplying a contiguous range of K-steps; see the fix-up after the main loop.
    const int ksplit = SK ? p.ksplit : 1;
    const int tile_id = SK ? bid / ksplit : bid;
    const int split = SK ? bid - tile_id * ksplit : 0;
    const int tile_n = tile_id % tiles_n, tile_m = tile_id / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk_all = p.W.Kpad / BK;
    const int kbeg = SK ? (split * nk_all) / ksplit : 0;
    const int nk = SK ? ((split + 1) * nk_all) / ksplit - kbeg : nk_all;
    const int last = nk - 1;

    // DMA source addresses: piece = wave*XPW + i covers tile rows piece*8 .. +7
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    const half_t * xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; i++) {
        int gm = m0 + (wave * XPW + i) * 8 + prow;
        gm = gm < p.M ? gm : p.M - 1;
        xsrc[i] = p.A + (size_t)gm * p.lda + pchunk * 8 + kbeg * BK;
    }
    const half_t * wsrc[WT == W_F16 ? WPW : 1];
    if constexpr (WT == W_F16) {
#pragma unroll
        for (int i = 0; i < WPW; i++)
            wsrc[i] = (const half_t *)p.W.w16 + (size_t)(n0 + (wave * WPW + i) * 8 + prow) * p.W.Kpad + pchunk * 8 + kbeg * BK;
    }
    // quantised W: the tile is BN rows x 2 blocks; thread t owns blocks t, t + 256, ... (block id = kb * BN + row)
    constexpr int NB = (BN * 2 + NTHREADS - 1) / NTHREADS;
    RawBlock<WT> B0[NB], B1[NB];
    const int bnl = tid % BN, bkb = tid / BN;          // NB > 1 (BN = 256): block i is (row bnl, k-block i)
    const bool bact = (BN * 2 >= NTHREADS) || tid < BN * 2;

#define DMA_TILE(buf_, kt_)                                                                    \
    {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < XPW; i++)                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[i] + (kt_) * BK), \
                (__attribute__((address_space(3))) void *)(Xs + (buf_) * BM * BK + (wave * XPW + i) * 512), 16, 0, 0); \
        if constexpr (WT == W_F16) {                                                           \
            _Pragma("unroll") for (int i = 0; i < WPW; i++)                                    \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[i] + (kt_) * BK), \
                    (__attribute__((address_space(3))) void *)(Ws + (buf_) * BN * BK + (wave * WPW + i) * 512), 16, 0, 0); \
        }                                                                                      \
    }
#define LOAD_B(B, kt_)                                                                         \
    if constexpr (WT != W_F16) {                                                               \
        if (bact) {                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < NB; i_++)                                  \
                load_block<WT>(B[i_], p.W, (size_t)(((kt_) + kbeg) * 2 + bkb + i_ * (NTHREADS / BN)) * p.W.Npad + n0 + bnl); \
        }                                                                                      \
    }
#define STORE_B(B, buf_)                                                                       \
    if constexpr (WT != W_F16) {                                                               \
        if (bact) {                                                                            \
            half_t * wrow_ = Ws + (buf_) * BN * BK + bnl * BK;                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < NB; i_++)                                  \
                _Pragma("unroll") for (int j = 0; j < 4; j++)                                  \
                    *(h8 *)(wrow_ + ((((bkb + i_ * (NTHREADS / BN)) * 4 + j) ^ (bnl & 7)) << 3)) = dequant_wfrag<WT>(block_word<WT>(B[i_], j), j); \
        }                                                                                      \
    }
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
// All 16 fragment reads of a K-step are issued up front (both 32-wide k-slices) and pinned above the MFMA block:
// left to itself hipcc re-uses fragment registers and puts an s_waitcnt lgkmcnt(0) in front of every group of
// 4-8 MFMAs (four exposed LDS round trips per K-step: the ds_read+MFMA loop alone then tops out at ~44 % of the
// MFMA peak); in this order the waits become counted (lgkmcnt(15..0)) and only the first round trip is exposed.
#define COMPUTE(buf_)                                                                          \
    {                                                                                          \
        const half_t * xs = Xs + (buf_) * BM * BK;                                             \
        const half_t * ws = Ws + (buf_) * BN * BK;                                             \
        h8 xf[2][TM], wf[2][TN];                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                     \
            _Pragma("unroll") for (int a = 0; a < TN; a++)                                     \
                wf[kk][a] = *(const h8 *)(ws + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fgrp)); \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                     \
                xf[kk][b] = *(const h8 *)(xs + lds_off(wm * (BM / 2) + b * 16 + frow, kk * 4 + fgrp)); \
        }                                                                                      \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                       \
            _Pragma("unroll") for (int a = 0; a < TN; a++)                                     \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][a], xf[kk][b], acc[a][b], 0, 0, 0); \
        /* schedule: the TN+TM reads of slice 0, then slice-0 MFMAs with the slice-1 reads slotted in, then slice-1 MFMAs */ \
        COMPUTE_SCHED                                                                          \
    }
#define COMPUTE_SCHED                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, TN + TM, 0);                               \
        _Pragma("unroll") for (int i = 0; i < TN + TM; i++) {                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, (TN * TM) / (TN + TM), 0);             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
        }                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN * TM - (TN + TM) * ((TN * TM) / (TN + TM)), 0);

// Interleaved K-step (BM >= 128, BN = 128): the step is cut into groups of GS MFMAs fenced by
// sched_barrier(0), and the other work is placed BETWEEN the groups in source order — slice-1 fragment reads and the
// LDS-DMA pieces of the next activation tile among the slice-0 MFMAs, the dequantisation + ds_write of the next weight
// tile (one 8-weight word per slot) among the slice-1 MFMAs — so VALU / LDS / VMEM issue runs under the matrix pipe
// instead of after it.
    constexpr bool ILV = BM >= 128 && BN >= 128;
    constexpr int NPC = XPW + (WT == W_F16 ? WPW : 0);   // LDS-DMA pieces per wave per K-step
    constexpr int NM = TN * TM, GS = 2, NG = (NM + GS - 1) / GS;
#define FRAG(base_, row_, c_) (*(const h8 *)((base_) + lds_off((row_), (c_))))
#define STEP_ILV(buf_, dkt_, BL, lkt_, BS)                                                     \
    {                                                                                          \
        const half_t * xs = Xs + (buf_) * BM * BK;                                             \
        const half_t * ws = Ws + (buf_) * BN * BK;                                             \
        half_t * wrow_ = Ws + ((buf_) ^ 1) * BN * BK + bnl * BK;                               \
        h8 xf[2][TM], wf[2][TN];                                                               \
        _Pragma("unroll") for (int a = 0; a < TN; a++) wf[0][a] = FRAG(ws, wn * (BN / 2) + a * 16 + frow, fgrp); \
        _Pragma("unroll") for (int b = 0; b < TM; b++) xf[0][b] = FRAG(xs, wm * (BM / 2) + b * 16 + frow, fgrp); \
        LOAD_B(BL, lkt_);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int g = 0; g < NG; g++) {                                       \
            _Pragma("unroll") for (int i = 0; i < GS; i++) {                                   \
                const int idx = g * GS + i;                                                    \
                if (idx < NM) acc[idx / TM][idx % TM] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][idx / TM], xf[0][idx % TM], acc[idx / TM][idx % TM], 0, 0, 0); \
            }                                                                                  \
            if (g < TN) wf[1][g < TN ? g : 0] = FRAG(ws, wn * (BN / 2) + g * 16 + frow, 4 + fgrp); \
            else if (g < TN + TM) xf[1][g < TN + TM ? g - TN : 0] = FRAG(xs, wm * (BM / 2) + (g - TN) * 16 + frow, 4 + fgrp); \
            _Pragma("unroll") for (int pc = 0; pc < NPC; pc++)                                 \
                if (g == (pc * NG) / NPC) {                                                    \
                    if (pc < XPW)                                                              \
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[pc < XPW ? pc : 0] + (dkt_) * BK), \
                            (__attribute__((address_space(3))) void *)(Xs + ((buf_) ^ 1) * BM * BK + (wave * XPW + pc) * 512), 16, 0, 0); \
                    else if constexpr (WT == W_F16)                                            \
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc[pc - XPW >= 0 ? pc - XPW : 0] + (dkt_) * BK), \
                            (__attribute__((address_space(3))) void *)(Ws + ((buf_) ^ 1) * BN * BK + (wave * WPW + pc - XPW) * 512), 16, 0, 0); \
                }                                                                              \
            _Pragma("unroll") for (int fr = NG; fr < TN + TM; fr++)   /* more fragments than groups: the rest in the last group */ \
                if (g == NG - 1) {                                                             \
                    if (fr < TN) wf[1][fr < TN ? fr : 0] = FRAG(ws, wn * (BN / 2) + fr * 16 + frow, 4 + fgrp); \
                    else xf[1][fr - TN >= 0 ? fr - TN : 0] = FRAG(xs, wm * (BM / 2) + (fr - TN) * 16 + frow, 4 + fgrp); \
                }                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
        _Pragma("unroll") for (int g = 0; g < NG; g++) {                                       \
            _Pragma("unroll") for (int i = 0; i < GS; i++) {                                   \
                const int idx = g * GS + i;                                                    \
                if (idx < NM) acc[idx / TM][idx % TM] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][idx / TM], xf[1][idx % TM], acc[idx / TM][idx % TM], 0, 0, 0); \
            }                                                                                  \
            if constexpr (WT != W_F16) {                                                       \
                _Pragma("unroll") for (int wd = 0; wd < 4 * NB; wd++)                          \
                    if (g == (wd * NG) / (4 * NB))                                             \
                        *(h8 *)(wrow_ + ((((bkb + (wd >> 2) * (NTHREADS / BN)) * 4 + (wd & 3)) ^ (bnl & 7)) << 3)) =       \
                            dequant_wfrag<WT>(block_word<WT>(BS[wd >> 2], wd & 3), wd & 3);    \
            }                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    }

    DMA_TILE(0, 0);
    LOAD_B(B0, 0);
    { const int t1 = last < 1 ? last : 1; LOAD_B(B1, t1); }
    // LayerNorm folded into this GEMM (gemm_common.h): thread t < BM reduces the statistics of row m0 + t to (mean, rstd) while the
    // first tiles are in flight; the epilogue picks them up through LDS
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;
    const bool ln = LNE && p.ln_c != nullptr;
    float2 ln_mine = make_float2(0.f, 1.f);
    if constexpr (LNE) {
        if (ln && tid < BM) ln_mine = ln_row_centred(p, m0 + tid < p.M ? m0 + tid : p.M - 1, tile_n == 0 && split == 0);
    }
    STORE_B(B0, 0);
    __syncthreads();
    int kt = 0;
#ifdef CLIPAMD_ABLATION   // kernel-tuning builds only (scripts/build_variant.sh abl -DCLIPAMD_ABLATION): p.debug switches parts of the loop off
    if (p.debug != 0) {
    const bool noload = p.debug & 1, nomfma = p.debug & 2, nostore = p.debug & 4;
    for (; kt + 1 < nk; kt += 2) {
        if (!noload) { DMA_TILE(1, kt + 1); const int t2 = kt + 2 < last ? kt + 2 : last; LOAD_B(B0, t2); }
        if (!nomfma) COMPUTE(0);
        if (!nostore) { STORE_B(B1, 1); }
        __syncthreads();
        if (!noload) { const int t2 = kt + 2 < last ? kt + 2 : last; DMA_TILE(0, t2); const int t3 = kt + 3 < last ? kt + 3 : last; LOAD_B(B1, t3); }
        if (!nomfma) COMPUTE(1);
        if (!nostore) { STORE_B(B0, 0); }
        __syncthreads();
    }
    } else
#endif
    if constexpr (ILV) {
    for (; kt + 1 < nk; kt += 2) {
        { const int t2 = kt + 2 < last ? kt + 2 : last; STEP_ILV(0, kt + 1, B0, t2, B1); }
        __syncthreads();
        { const int t2 = kt + 2 < last ? kt + 2 : last; const int t3 = kt + 3 < last ? kt + 3 : last; STEP_ILV(1, t2, B1, t3, B0); }
        __syncthreads();
    }
    if (kt < nk) COMPUTE(0);
    } else {
    for (; kt + 1 < nk; kt += 2) {
        DMA_TILE(1, kt + 1);
        { const int t2 = kt + 2 < last ? kt + 2 : last; LOAD_B(B0, t2); }
        COMPUTE(0);
        STORE_B(B1, 1);
        __syncthreads();
        { const int t2 = kt + 2 < last ? kt + 2 : last; DMA_TILE(0, t2); }
        { const int t3 = kt + 3 < last ? kt + 3 : last; LOAD_B(B1, t3); }
        COMPUTE(1);
        STORE_B(B0, 0);
        __syncthreads();
    }
    if (kt < nk) COMPUTE(0);
    }
#undef STEP_ILV
#undef FRAG
#undef DMA_TILE
#undef LOAD_B
#undef STORE_B
#undef COMPUTE
#undef COMPUTE_SCHED
    asm volatile("" ::: "memory");
#ifdef CLIPAMD_ABLATION   // p.debug bit 3: the K loop alone — no epilogue (the never-true store keeps the accumulators alive)
    if (p.debug & 8) {
        float keep = 0.f;
#pragma unroll
        for (int a = 0; a < TN; a++)
#pragma unroll
            for (int b = 0; b < TM; b++) keep += (acc[a][b][0] + acc[a][b][1]) + (acc[a][b][2] + acc[a][b][3]);
        if (keep == 1.2345e33f) ((float *)p.out)[tid] = keep;
        return;
    }
#endif
    if constexpr (SK) {
        if (ksplit > 1) {
            // Deterministic split-K fix-up: every workgroup parks its partial tile in the workspace ([tile][split][frag][thread]
            // float4, coalesced); the LAST one to arrive (per-tile ticket counter, self-resetting) re-reads all ksplit partials
            // in split order 0..ksplit-1 — the same summation order whoever arrives last — and runs the epilogue.
            // Hand-off without cache-wide fences (MI355X_MICROARCH.md, inter-workgroup visibility): partials are written
            // with agent-scope (sc1, write-through) stores, drained with s_waitcnt vmcnt(0) before the ticket, and read
            // back with agent-scope loads that bypass the non-coherent L1 / per-XCD L2 copies.
            __shared__ int is_last;
            typedef unsigned long long u64;
            u64 * part = (u64 *)(p.sk_ws + ((size_t)tile_id * ksplit) * (BM * BN));
            constexpr int PER_SPLIT = BM * BN / 2;     // u64 per partial tile
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    const float2 lo = make_float2(acc[a][b][0], acc[a][b][1]), hi = make_float2(acc[a][b][2], acc[a][b][3]);
                    u64 * dst = part + (size_t)split * PER_SPLIT + ((a * TM + b) * 2) * NTHREADS + tid;
                    __hip_atomic_store(dst, __builtin_bit_cast(u64, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + NTHREADS, __builtin_bit_cast(u64, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(p.sk_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                is_last = (ticket == (unsigned)ksplit - 1u);
                if (is_last) __hip_atomic_store(p.sk_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
            }
            __syncthreads();
            if (!is_last) return;
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < ksplit; sp++) {
#pragma unroll
                for (int a = 0; a < TN; a++)
#pragma unroll
                    for (int b = 0; b < TM; b++) {
                        const u64 * src = part + (size_t)sp * PER_SPLIT + ((a * TM + b) * 2) * NTHREADS + tid;
                        const float2 lo = __builtin_bit_cast(float2, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        const float2 hi = __builtin_bit_cast(float2, __hip_atomic_load(src + NTHREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        acc[a][b] += (f4){lo.x, lo.y, hi.x, hi.y};
                    }
            }
        }
    }
    // LDS after the K loop: [0, BM * 272) the four waves' fp16 staging areas ((BM / 2) rows x 136 bytes each), then BM float2 of row statistics
    float2 * ln_rs = (float2 *)(smem_raw + ((BM * 272 + 15) & ~15));
    if constexpr (LNE) {
        if (ln) ln_rows_publish(ln_rs, ln_mine, tid, BM, [] { __syncthreads(); });
    }
    const float2 * rs_lane = ln_rs + wm * (BM / 2) + frow;
    if constexpr ((EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16) && BN == 128 && BM >= 128) {
        const int nb = n0 + wn * (BN / 2);
        if (nb + BN / 2 <= p.W.N && (p.ldc & 7) == 0) {                 // uniform per wave
            if ((nk & 1) && !ln) __syncthreads();                        // odd K-step count: the tail COMPUTE(0) had no barrier behind it
            gemm_epilogue_f16_staged<EPI, TN, TM>(p, acc, nb, m0 + wm * (BM / 2), frow, fgrp, (half_t *)smem_raw + wave * (BM / 2) * 68, lane, ln, rs_lane);
            return;
        }
    }
    if constexpr (EPI == EPI_RESID_F32 && BN == 128 && BM >= 128) {
        if (p.xg_out) {       // producer half of the fold: xg goes through the wave's staging area (full-line stores)
            if (nk & 1) __syncthreads();
            gemm_epilogue<EPI, TN, TM>(p, acc, n0 + wn * (BN / 2), m0 + wm * (BM / 2), frow, fgrp, false, rs_lane, (half_t *)smem_raw + wave * (BM / 2) * 68, lane);
            return;
        }
    }
    gemm_epilogue<EPI, TN, TM>(p, acc, n0 + wn * (BN / 2), m0 + wm * (BM / 2), frow, fgrp, ln, rs_lane);
}

template <int WT, int BM, int BN, int EPI>
void launch_dma(const GemmParams & p, hipStream_t stream) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + BN - 1) / BN;
    constexpr size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(half_t);
    static unsigned long long lds_ok = 0;
    if (smem > 64 * 1024) opt_in_dynamic_lds(gemm_dma_kernel<WT, BM, BN, EPI>, smem, lds_ok);
    int grid = tiles_m * tiles_n;
    if (gemm_tile_splits_k(BM, BN)) grid *= p.ksplit;   // p.ksplit validated by launch_gemm
    hipLaunchKernelGGL((gemm_dma_kernel<WT, BM, BN, EPI>), dim3(grid), dim3(NTHREADS), smem, stream, p);
}

// tile code: BM * 1000 + BN  (0 = heuristic)
template <int WT, int EPI>
void launch_tile(const GemmParams & p, int tile, hipStream_t stream) {
    switch (tile % 1000000) {
    case 128128: launch_dma<WT, 128, 128, EPI>(p, stream); break;
    case 160128: launch_dma<WT, 160, 128, EPI>(p, stream); break;
    case 192128: launch_dma<WT, 192, 128, EPI>(p, stream); break;
#ifdef CLIPAMD_BIGTILES   // tuning builds: BN = 256 tiles, one workgroup per CU, accumulators in AGPRs
    case 128256: launch_dma<WT, 128, 256, EPI>(p, stream); break;
    case 160256: launch_dma<WT, 160, 256, EPI>(p, stream); break;
    case 192256: launch_dma<WT, 192, 256, EPI>(p, stream); break;
    case 256256: launch_dma<WT, 256, 256, EPI>(p, stream); break;
#endif
    case 64128: launch_dma<WT, 64, 128, EPI>(p, stream); break;
    case 128064: launch_dma<WT, 128, 64, EPI>(p, stream); break;
    default: launch_dma<WT, 64, 64, EPI>(p, stream); break;
    }
}

template <int WT>
void launch_epi(const GemmParams & p, int epi, int tile, hipStream_t stream) {
    switch (epi) {
    case EPI_F32: launch_tile<WT, EPI_F32>(p, tile, stream); break;
    case EPI_F16: launch_tile<WT, EPI_F16>(p, tile, stream); break;
    case EPI_GELU_F16: launch_tile<WT, EPI_GELU_F16>(p, tile, stream); break;
    case EPI_QGELU_F16: launch_tile<WT, EPI_QGELU_F16>(p, tile, stream); break;
    case EPI_RESID_F32: launch_tile<WT, EPI_RESID_F32>(p, tile, stream); break;
    case EPI_PATCH_F32: launch_tile<WT, EPI_PATCH_F32>(p, tile, stream); break;
    }
}

// Tile heuristic, fitted to scripts/gemm_bench.py measurements (profiles/README.md).  Small problems take the smallest
// tiles (most workgroups).  Otherwise BN = 128 and BM in {64, 128, 160, 192} minimising
//     g(tiles / 512) x (BM + 32)            [x 1.15 for BM = 64]
// where 512 = 256 CUs x 2 co-resident workgroups, BM + 32 models the per-tile cost (the +32 is the W dequant that does
// not shrink with BM) and g() the wave quantisation: a lone workgroup per CU runs ~0.7x the time of a co-resident pair,
// one partial round costs a full round, later rounds overlap (3/4 fractional + 1/4 ceil).  E.g. M = 12800, N = 768:
// 600 tiles of 128x128 = 1.17 rounds, but 480 tiles of 160x128 = one round (measured 100.6 -> 79.8 us at K = 3072).
// CLIP_AMD_TILE_OVERRIDE="M,N,K,tile[;M,N,K,tile...]" (tuning aid; K = 0 matches any depth): the heuristic's answer for the listed problem
// sizes, so that a tile can be A/B-ed INSIDE the layer chain (isolated GEMM timings over-state a tile's worth there: profiles/r05_experiments.txt section 3)
int tile_override(int M, int N, int Kpad) {
#if !CLIPAMD_TEST_HOOKS
    (void)M; (void)N; (void)Kpad;
    return 0;           // (a tuning aid: compiled out of `make hooks=0` builds together with the kernel test hooks)
#else
    struct Ov { int M, N, K, tile; };
    static const std::vector<Ov> ovs = [] {
        std::vector<Ov> v;
        const char * e = getenv("CLIP_AMD_TILE_OVERRIDE");
        while (e && *e) {
            Ov o;
            int used = 0;
            if (sscanf(e, "%d,%d,%d,%d%n", &o.M, &o.N, &o.K, &o.tile, &used) == 4) v.push_back(o); else break;
            e += used;
            if (*e == ';') e++;
        }
        return v;
    }();
    for (const Ov & o : ovs) if (o.M == M && o.N == N && (o.K == 0 || o.K == Kpad)) return o.tile;
    return 0;
#endif
}

int pick_tile(int M, int N, int Kpad, bool quantised, bool shared = false) {
    auto wgs = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (const int ov = tile_override(M, N, Kpad)) return ov;
    if (M <= 64) return 64064;
    // (round 5, measured and NOT done — profiles/r05_experiments.txt: the fp16-output GEMMs of the ViT-B/32 batch on resident fp16 panels and the
    //  four-wave 256 x 256 kernel — isolated q/k/v 57.3 us against 64.1, text q/k/v 25.7 against 29.4 — leave the vision q/k/v launch of the
    //  layer chain where it was (0.791 against 0.798 ms per step) and cost the two-tower step 1.5 %: 130 KB workgroups keep the other tower's
    //  kernels off the CUs.  The same for a persistent 8-wave kernel with its stores deferred under the next tile's K loop: level at best.)
    if (M <= 4096) {
        // mid-M (a few hundred to a few thousand rows: batches of 2-64 ViT-B/32 images, batches of texts, single ViT-L/14 images): the
        // ring kernel of k_gemm_ring.hip where the sweep of profiles/r02_ring_sweep_*.txt (M = 130 ... 3200 x the model widths, q4_0 and
        // f16, every tile form) has it ahead: short K — 64 x 64 tiles while they number <= ~260, else 64 x 128 up to ~500 tiles;
        // long K (32+ K-tiles) — fp16 weights on 64 x 64 tiles, block-quantised weights on 64 x 128 from ~1000 rows (below that the
        // split-K form of the two-buffer kernel is level or ahead).  Gains there: 10-50 % per GEMM; outside, the kernels below.
        const int tm = (M + 63) / 64, t64 = tm * ((N + 63) / 64), t128 = tm * ((N + 127) / 128), nk = Kpad / BK;
        if (nk < 32) {
            if (t64 <= 260) return 65064;
            // (r03: at 1600 x 2304 x 768 — q/k/v of 32 ViT-B/32 images — 234 tiles of 128 x 128 run 12.7 us against 13.9 for the 450 ring tiles in
            //  isolation and the vision tower gains 1.7 %, but with the text tower on the second stream the step LOSES 4.5 %: the 64 KB
            //  workgroups leave the text kernels no room on the CUs; same-box A/B in profiles/r03_lnfold_and_text_tiles.txt section 6.  The ring stays.)
            if (t128 <= 500) return 65128;
        } else if (!quantised) {
            if (t64 <= 340) return 65064;
        } else if (M >= 1024 && t128 <= 340) {
            return 65128;
        }
    }
    if (wgs(128, 128) < 100) return wgs(64, 128) >= 256 ? 64128 : 64064;
    // large M: the 8-wave 160 x 256 kernel (k_gemm8.hip), one workgroup per CU -> rounds of 256 tiles.  Its K loop is the faster
    // one (3-stage ring, ping-pong: ~69 % of the MFMA peak in the loop) but with one workgroup per CU nothing overlaps a tile's
    // prologue and epilogue, so it wins where the K loop dominates the tile: long K, or many rounds (profiles/r02_gemm8_*.txt)
    // the largest problems (ViT-L/14 at batch 256: 65792 rows; ViT-B/32 from batch ~700): 256 x 256 tiles, four waves of 128 x 128 with
    // the accumulators in AGPRs (k_gemm4.hip), on the tile rows that fill whole rounds of 256 workgroups + a second launch for the rest.
    // Sustained (200 launches, profiles/r02_gemm8_experiments.txt section 11): l14.qkv 447 us vs 468 (160 x 256) / 485 (160 x 128),
    // l14.down 542 vs 579 / 607, l14.up 645 vs 652 / 705; at 32896 rows (batch 128): qkv 223 vs 238 / 241, down 287 vs 321 / 333,
    // out 110 vs 117 / 107, up 333 vs 332 / 360.  From two rounds of tiles up.
    // (round 6: q/k/v and FFN-up of a ViT-L/14-class batch alone on the device on the 320 x 256 tile of k_gemm32.hip — isolated 494 / 677 us against
    //  501 / 698 for the split below at 65792 rows, profiles/r06_experiments.txt section 5b; the residual GEMMs stay: 668 / 290 against 602 / 256)
    if (!shared && !quantised && M >= 32768 && N >= 3 * Kpad && (N & 63) == 0 && Kpad >= 128 && wgs(320, 256) >= 3 * 256) return 320261;
    if (M >= 32768 && wgs(256, 256) >= 2 * 256) return 256260;
    // Round 6 — the wide fp16-output GEMMs of a ViT-B/32-class batch (q/k/v: N = 3 K, FFN-up: N = 4 K; 8192-32767 rows) on fp16 weights
    // (f16 files, or a resident panel of a block-quantised weight: forward.cpp resident_panels asks with quantised = false): the
    // 32 x 32 x 16 kernel of k_gemm32.hip, one workgroup per CU, on whichever of its two tiles (256 / 320 x 256) wastes less of its last
    // round of 256 — and only where at least 80 % of the rounds' slots hold a tile (text FFN-up 10290 x 2048: 328 / 264 tiles = 64 / 52 %:
    // the two-per-CU kernel below stays ahead there).  Other epilogues at such a shape fall through to the 16 x 16 x 32 kernels (launch_gemm).
    // ONLY when the device is not shared (GemmParams::shared_device): a tower alone gains 2.3 % (vision) / 3.4 % (text) from it, but with the
    // other tower on a second stream the two-tower step LOSES 0.4-2.7 % (profiles/r06_experiments.txt section 3): a 135-150 KB workgroup has
    // its CU to itself, so its prologue and epilogue (45 % of its tile time) idle the matrix pipe, where two 72-80 KB workgroups of either
    // tower fill each other's.  At M >= 32768 (ViT-L/14) it is level with k_gemm4.hip (0.32-0.34 of the MFMA peak both) and not used.
    if (!shared && !quantised && M >= 8192 && N >= 2 * Kpad && (N & 63) == 0 && Kpad >= 128) {
        const int t4 = wgs(256, 256), t5 = wgs(320, 256);
        const int c4 = ((t4 + 255) / 256) * 256, c5 = ((t5 + 255) / 256) * 320;     // rounds x rows per tile ~ time
        const bool five = c5 < c4;
        const int t = five ? t5 : t4;
        if ((float)t >= 0.8f * (float)(((t + 255) / 256) * 256)) return five ? 320261 : 256261;
    }
    // Round 6: the out-projection (N = K) of a ViT-B/32-class batch on fp16 weights (f16 file or resident panel: forward.cpp resident_panels) on the 8-wave
    // 160 x 256 kernel, whose residual rows are requested before its K loop: one round of 240 workgroups whose memory-bound epilogue phase is 40 % shorter —
    // +1.0 ... +1.3 % on the two-tower step against the two-per-CU tile (profiles/r06_experiments.txt section 14)
    if (!quantised && M >= 8192 && M < 32768 && N == Kpad && Kpad >= 512 && wgs(160, 256) >= 200 && wgs(160, 256) <= 256) return 160256;
    {
        const int t8 = wgs(160, 256);
        const float rounds = (float)t8 / 256.f;
        const float eff = rounds / ceilf(rounds);          // fraction of the last round's CUs that have work
        // block-quantised weights additionally pay the panel dequantisation (a ~7 us launch per layer at ViT-B/32 size: more than
        // the 8-wave kernel gains there, r02h), so they take this path only where a layer is milliseconds long
        if (t8 >= 200 && eff >= 0.75f && (M >= 32768 || (Kpad >= 2048 && !quantised))) return 160256;
    }
    // One exception measured INSIDE the two-tower step (r05, profiles/r05_experiments.txt section 9): the q/k/v GEMM of a narrow tower (10290 x 1536 x 512, the
    // ViT-B/32 text tower of the BASELINE batch) on 192 x 128 instead of the model's 128 x 128: +1.0 % of the step in five same-box pairs — level in
    // isolation (33.4 vs 32.8 us): fewer, larger workgroups leave the other tower's kernels more room.  (A larger fixed part in the cost model, which also
    // moves the wide tower's q/k/v and FFN-up to 192 x 128, gains another 0.2 % there and LOSES 1-3 % at 96 / 176 / 224 / 400 / 512 images: section 9. Not kept.)
    if (Kpad <= 512 && N <= 1536 && N >= 1024 && M >= 8192 && M <= 16384) return 192128;
    int best = 128128;
    float best_cost = 0.f;
    const int cand[4] = {128, 160, 192, 64};
    for (int bm : cand) {
        const float x = (float)wgs(bm, 128) / 512.f;
        const float g = x <= 0.5f ? 0.7f : x <= 1.f ? 1.f : 0.75f * x + 0.25f * ceilf(x);
        const float cost = g * (float)(bm + 32) * (bm == 64 ? 1.35f : 1.f);   // (1.15 until r03: the 4500-8000-row sweep has 128 x 128 ahead of 64 x 128 by 8-22 %)
        if (best_cost == 0.f || cost < best_cost) { best_cost = cost; best = bm * 1000 + 128; }
    }
    return best;
}

}  // namespace

// One translation unit per weight type (compiled in parallel by build.py with -DCLIPAMD_GEMM_WT=<n>);
// the dispatcher is compiled once with no define.
#ifdef CLIPAMD_GEMM_WT
#define CLIPAMD_CAT2(a, b) a##b
#define CLIPAMD_CAT(a, b) CLIPAMD_CAT2(a, b)
void CLIPAMD_CAT(launch_gemm_wt, CLIPAMD_GEMM_WT)(const GemmParams & p, int epilogue, int tile, hipStream_t stream) {
    launch_epi<CLIPAMD_GEMM_WT>(p, epilogue, tile, stream);
}
#else
void launch_gemm_wt0(const GemmParams &, int, int, hipStream_t);
void launch_gemm_wt1(const GemmParams &, int, int, hipStream_t);
void launch_gemm_wt2(const GemmParams &, int, int, hipStream_t);
void launch_gemm_wt3(const GemmParams &, int, int, hipStream_t);
void launch_gemm_wt4(const GemmParams &, int, int, hipStream_t);
void launch_gemm_wt5(const GemmParams &, int, int, hipStream_t);

int gemm_tile_for(int M, int N, int Kpad, bool quantised, bool shared_device) { return pick_tile(M, N, Kpad, quantised, shared_device); }

// LayerNorm fold: columns per statistics slot written by the residual epilogue of the kernel behind a tile code (fold_slotw<TN>() of
// gemm_common.h: 64 where a wave spans >= 64 columns — the BN = 128 tiles of this file, k_gemm8.hip, k_gemm4.hip —, else 32)
int gemm_fold_slotw(int tile) {
    tile %= 1000000;
    if (gemm_tile_is_ring(tile)) return 32;                        // 2 x 2 and 4 x 1 waves over 64 / 128 columns: 32 per wave
    if (gemm_tile_uses_panel(tile)) return 64;                     // (also when the panel is missing: the fallback is the 160 x 128 tile)
    return tile % 1000 == 64 ? 32 : 64;
}
// the rows past the whole rounds of a split launch (tile codes 258 / 260): in fold mode they must leave the same slot width as the head
static int fold_rest_tile(int M, int N, int Kpad, bool quantised) {
    const int t = pick_tile(M, N, Kpad, quantised);
    return gemm_fold_slotw(t) == 64 && t % 1000 != 258 && t % 1000 != 260 ? t : 64128;
}
int gemm_fold_slotw_for(int M, int N, int Kpad, bool quantised, bool shared_device) { return gemm_fold_slotw(pick_tile(M, N, Kpad, quantised, shared_device)); }

// Split-K factor for a BM = 64 tile grid (small-M problems: batch 1 / 32, single texts), fitted with
// scripts/gemm_bench.py (profiles/r01_gemm_splitk.txt): a K-step costs ~0.5 us of serial latency, the fix-up ~3 us, so
// splitting pays when the K loop is long — ksplit ~ sqrt(nk / 1.5) (12 steps -> 3, 48 -> 6) — and, once the grid already
// fills the chip (>= 256 tiles), not at all (profiles/r03_midlarge_sweep.txt).
int pick_ksplit(int tiles, int nk) {
    if (nk < 8) return 1;
    int ks = (int)(sqrtf((float)nk / 1.5f) + 0.5f);
    if (tiles >= 256) ks = 1;      // (r03 sweep, 4500-8000 rows x K = 2048 / 3072: 3-4-way splits of 284-500 tiles ran 15-45 % SLOWER than unsplit)
    if (tiles * ks > 1536) ks = 1536 / tiles;
    return ks < 1 ? 1 : ks > 16 ? 16 : ks;
}

// Split-K for the ring kernel (fitted to profiles/r02_ring_sweep_*.txt): only the long-K GEMMs (FFN down: 32+ K-tiles) with few tiles.
int pick_ksplit_ring(int tiles, int nk, bool quantised) {
    if (nk < 32) return 1;
    if (quantised) return tiles <= 160 ? 3 : 1;
    return tiles <= 110 ? 2 : 1;
}

void launch_gemm(const GemmParams & p0, int epilogue, int tile, hipStream_t stream) {
    if (p0.M <= 0) return;
    if (p0.W.wtype == W_F32) {      // f32 GGUF file: f32 weights on the exact-f32 MFMA, whatever the tile code (k_gemm_f32.hip)
        launch_gemm_f32(p0, epilogue, stream);
        return;
    }
    GemmParams p = p0;
    const bool heuristic = (tile == 0);
    int ksplit = tile / 1000000;        // explicit: ksplit * 1000000 + BM * 1000 + BN  (no prefix = no split)
    tile %= 1000000;
    if (p.w16_pre && p.W.wtype != W_F16) {
        // the caller holds this weight dequantised (resident or per-layer fp16 panel, same values as the fused dequantisation):
        // every kernel multiplies it as an f16 weight
        p.W.wtype = W_F16;
        p.W.w16 = p.w16_pre;
    }
    if (heuristic) {
        tile = pick_tile(p.M, p.W.N, p.W.Kpad, p.W.wtype != W_F16, p.shared_device);
        // the 32 x 32 x 16 kernel carries the fp16-output epilogues only: any other launch of such a shape gets the choice made without it
        if (tile % 1000 == 261 && !gemm32_supported(p, epilogue)) tile = pick_tile(p.M, p.W.N, p.W.Kpad, p.W.wtype != W_F16, true);
    }
    if (tile % 1000 == 258 || tile % 1000 == 260) {
        // 256 x 256 tiles in whole rounds: one workgroup per CU means a launch costs ceil(tiles / 256) rounds, and ViT-L/14's
        // 65792 rows are 257 tile rows — one past a round boundary for every N.  The leading tile rows that fill whole rounds go to
        // the 256 x 256 kernel, the remaining rows to whatever the heuristic picks for that (much smaller) problem; the epilogues
        // are row-local, so the two launches write disjoint rows (not the patch epilogue: its rows map to (image, token)).
        const int tiles_n = (p.W.N + 255) / 256, tiles_m = (p.M + 255) / 256;
        const int mb = (int)(((long)tiles_m * tiles_n / 256) * 256 / tiles_n);     // tile rows inside the whole rounds
        tile = tile % 1000 == 258 ? 256256 : 256259;
        if (epilogue != EPI_PATCH_F32 && mb > 0 && mb < tiles_m) {
            const int m1 = mb * 256;
            const bool f16_out = epilogue == EPI_F16 || epilogue == EPI_GELU_F16 || epilogue == EPI_QGELU_F16;
            GemmParams head = p0, rest = p0;
            head.M = m1;
            rest.M = p.M - m1;
            rest.A = p.A + (size_t)m1 * p.lda;
            rest.out = (char *)p.out + (size_t)m1 * p.ldc * (f16_out ? sizeof(half_t) : sizeof(float));
            if (p.resid) rest.resid = p.resid + (size_t)m1 * p.ldc;
            // LayerNorm fold: the statistics are [slot][row] (row offset = pointer offset), xg is row-major
            if (p.ln_stats) rest.ln_stats = p.ln_stats + m1;
            if (p.ln_mu) rest.ln_mu = p.ln_mu + m1;
            if (p.mu_out) rest.mu_out = p.mu_out + m1;
            if (p.xg_mu) rest.xg_mu = p.xg_mu + m1;
            if (p.stats_out) rest.stats_out = p.stats_out + m1;
            if (p.xg_out) rest.xg_out = p.xg_out + (size_t)m1 * p.ldxg;
            launch_gemm(head, epilogue, tile, stream);
            launch_gemm(rest, epilogue, p.xg_out ? fold_rest_tile(rest.M, p.W.N, p.W.Kpad, p.W.wtype != W_F16) : 0, stream);
            return;
        }
    }
    if (gemm_tile_uses_panel(tile)) {
        // 8-wave large-M kernel: fp16 x fp16 from a row-major panel of W; block-quantised weights are dequantised into it first
        // (unless the caller already did, per layer)
        const half_t * panel = p.W.wtype == W_F16 ? (const half_t *)p.W.w16 : p.w16_pre;
        if (!panel && p.w16_scratch && p.w16_scratch_halfs >= (size_t)p.W.Npad * p.W.Kpad) {
            const DevWeight * w = &p.W;
            half_t * o = p.w16_scratch;
            launch_dequant(&w, &o, 1, stream);
            panel = p.w16_scratch;
        }
        if (panel) {
            p.W.wtype = W_F16;
            p.W.w16 = panel;
            if (tile % 1000 == 261 ) {   // k_gemm32.hip: the fp16-output GEMMs (q/k/v, FFN-up) on 32 x 32 x 16 fragments
                if (gemm32_supported(p, epilogue)) {
                    launch_gemm32(p, epilogue, tile / 1000 == 320 ? 5 : 4, stream);
                    return;
                }
                tile = 256259;                         // another epilogue / an odd shape: the 16 x 16 x 32 form of the same tile
            }
            if (tile % 1000 == 259 && !p.xg_out) {      // (consumer half of the LayerNorm fold: in the kernel since round 5)
                launch_gemm4(p, epilogue, stream);
                return;
            }
            if (tile % 1000 == 259) tile = 256256;     // fold PRODUCER launch: the 8-wave 256 x 256 tile carries the in-register tail (k_gemm4.hip says why it does not)
            launch_gemm8(p, epilogue, tile / 32000, stream);
            return;
        }
        tile = 160128;   // no panel available: the fused 4-wave kernel
    }
    const bool ring = gemm_tile_is_ring(tile);         // k_gemm_ring.hip: 64-row tiles, BN in {64, 128}
    const int bm = ring ? 64 : tile / 1000;
    int bn = tile % 1000;
    if (ring) bn = bn >= 128 ? 128 : 64;
    const int tiles = ((p.M + bm - 1) / bm) * ((p.W.N + bn - 1) / bn), nk = p.W.Kpad / BK;
    if (heuristic) ksplit = ring ? pick_ksplit_ring(tiles, nk, p.W.wtype != W_F16) : pick_ksplit(tiles, nk);
    if (ksplit < 1) ksplit = 1;
    if (!gemm_tile_splits_k(bm, bn) || !p.sk_ws || !p.sk_cnt || tiles > p.sk_cnt_n || p.no_splitk) ksplit = 1;
    if (ksplit > nk / 2) ksplit = nk / 2 > 0 ? nk / 2 : 1;
    while (ksplit > 1 && (size_t)tiles * ksplit * bm * bn > p.sk_ws_floats) ksplit--;
    p.ksplit = ksplit;
    if (ring) {
        launch_gemm_ring(p, epilogue, bn, stream);
        return;
    }
    switch (p.W.wtype) {
    case W_F16: launch_gemm_wt0(p, epilogue, tile, stream); break;
    case W_Q4_0: launch_gemm_wt1(p, epilogue, tile, stream); break;
    case W_Q4_1: launch_gemm_wt2(p, epilogue, tile, stream); break;
    case W_Q5_0: launch_gemm_wt3(p, epilogue, tile, stream); break;
    case W_Q5_1: launch_gemm_wt4(p, epilogue, tile, stream); break;
    case W_Q8_0: launch_gemm_wt5(p, epilogue, tile, stream); break;
    }
}
#endif

}  // namespace clipamd
