// k_gemm8.hip — the large-M GEMM of the hot path for gfx950 (CDNA4): 8-wave "ping-pong" workgroups on 256-wide tiles.
//
//   out[M][N] = epilogue( X[M][K] (fp16) · W16[N][K]^T (fp16) + bias ),   M >= ~2000 rows (batch >= ~64 images / ~2000 tokens)
//
// Same contraction, operands, MFMA instruction (v_mfma_f32_16x16x32_f16) and k order as k_gemm.hip — the results are
// bit-identical to every tile of that kernel (tests/test_gpu_kernels.py) — but a different machine mapping, built for the
// regime where the 4-wave 160x128 kernel is LDS-pipe / L2->LDS bound (r01: 0.29 of the MFMA peak):
//
//   * tile (32 TM) x 256 (TM = 5: 160 x 256), BK = 64, 512 threads = 8 waves in 2 (M) x 4 (N); each wave owns an 80 x 64
//     sub-tile = 5 x 4 MFMA fragments.  Per MFMA the workgroup streams 166 B from L2 into LDS (160x128 4-wave tile: 230 B).
//   * the two M-halves of the workgroup ("groups") run one barrier apart: while group 0 executes the 20 MFMAs of a k-slice,
//     group 1 (its SIMD partner: wave w and wave w+4 share a SIMD) reads the next fragments from LDS and issues the LDS-DMA
//     of a later tile, and vice versa.  The matrix pipe of every SIMD always has one wave in its MFMA segment
//     (s_setprio 1 there), the other wave's LDS / VMEM issue hides under it (cdna_hip_programming.md T3+T4 / T5).
//   * 3-stage LDS ring of whole K-tiles (3 x 52 KB = 156 KB of the 160 KB), filled by global_load_lds_dwordx4 only.  Tile t+2
//     is requested while tile t is multiplied; the only waits are COUNTED (s_waitcnt vmcnt(4): everything but the 4 newest
//     requests has landed) and sit one barrier before the first read of the tile they retire, so a request has ~6 segments
//     (~2000 cycles) to land: HBM misses included.  Barriers are raw s_barrier (no vmcnt(0) drain).
//   * hazards.  WAR: stage (t+2)%3 held tile t-1; every read of tile t-1 is retired (lgkmcnt(0)) BEFORE the barrier that ends
//     its segment, so by the time any wave is in tile t's first segment no read of tile t-1 is outstanding in either group.
//     RAW: a wave's vmcnt wait for tile t+1 is in the load segment of tile t's SECOND k-slice; the lagging group executes it
//     one segment later, and one more barrier separates it from the leading group's first read of tile t+1.
//   * block-quantised weights: dequantised ONCE per layer by dequant_kernel into an fp16 scratch panel ([N][K] row-major,
//     L2 / Infinity-Cache resident: 14 MB per ViT-B/32 layer) with exactly the packed-fp16 arithmetic of the fused kernel, then
//     multiplied from there.  At M = 12800 the fused kernel dequantises every weight tile 80 times (once per M-tile); here the
//     VALU / ds_write work of the dequantisation leaves the K loop entirely.  The persistent copy in HBM stays block-quantised.
//   * epilogues shared with k_gemm.hip (gemm_common.h); fp16 outputs are staged through the (now free) LDS ring so that global
//     stores are full 128-byte lines; residual rows are fetched 5 at a time.
//
// Reference ops replaced: ggml_mul_mat with a weight operand, clip.cpp:1360-1380,1392,1407,1416 (vision) and
// :1079-1095,1112,1127,1136 (text), at batch sizes the reference cannot reach (its arenas cap B at ~13).

#include "gemm_common.h"

#ifndef CLIPAMD_G8_PHASES
#define CLIPAMD_G8_PHASES 2   // MFMA segments per K-tile: 2 = one per 32-wide k-slice, 1 = one per K-tile (tuning: scripts/build_variant.sh)
#endif

namespace clipamd {

namespace {

constexpr int NT8 = 512;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raw_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#define GLDS16(src_, dst_)                                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_),                      \
                                     (__attribute__((address_space(3))) void *)(dst_), 16, 0, 0)

template <int TM, int EPI>
__global__ void __launch_bounds__(NT8, 2) gemm8_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * TM, BN = 256, TN = 4;
    constexpr int XB = BM * 128;                      // bytes of the X tile of one stage ([BM][64] fp16)
    constexpr int STAGE = XB + BN * 128;              // + the W tile ([256][64] fp16)
    constexpr int RPW = BM / 8;                       // X rows staged per wave per K-tile
    constexpr int XF = RPW / 8, XR = RPW % 8;         // full 8-row (1 KB) pieces + remainder rows (a half piece when XR == 4)
    static_assert(XR == 0 || XR == 4, "X rows per wave must be a multiple of 4");
    constexpr int NX = XF + (XR ? 1 : 0);             // X requests per wave per K-tile
    constexpr int NW = 4;                             // W requests per wave per K-tile (32 rows)
    // ring depth: three whole K-tiles where they fit the 160 KB LDS (BM <= 160), else two (BM = 256: 2 x 64 KB) with the requests of
    // tile t+1 all issued in the first load segment of tile t and retired in its second
    constexpr int S = 3 * STAGE <= 160 * 1024 ? 3 : 2;
    static_assert(S * STAGE <= 160 * 1024, "ring does not fit the LDS");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fgrp = lane >> 4;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-contiguous chunks, n fastest: workgroups sharing an activation row-panel run back to back on one XCD / L2
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int T = p.W.Kpad / BK;

    // LDS-DMA sources: the LDS image of a piece is lane-linear (row l>>3, 16-byte position l&7), so the XOR swizzle of the tile
    // (chunk ^= row & 7) is applied to the per-lane SOURCE chunk
    const int prow = lane >> 3;
    const half_t * xsrc[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int tr = wave * RPW + 8 * i + prow;
        int gm = m0 + tr;
        gm = gm < p.M ? gm : p.M - 1;
        xsrc[i] = p.A + (size_t)gm * p.lda + (((lane & 7) ^ (tr & 7)) << 3);
    }
    const half_t * wsrc[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) {
        const int tr = wave * 32 + 8 * j + prow;
        int gn = n0 + tr;
        gn = gn < p.W.Npad ? gn : p.W.Npad - 1;
        wsrc[j] = (const half_t *)p.W.w16 + (size_t)gn * p.W.Kpad + (((lane & 7) ^ (tr & 7)) << 3);
    }
#define ISSUE_W(st_, kt_)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NW; j++)                                                            \
            GLDS16(wsrc[j] + (size_t)(kt_) * BK, smem + (st_) * STAGE + XB + (wave * 32 + 8 * j) * 128);          \
    }
#define ISSUE_X(st_, kt_)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < XF; i++)                                                            \
            GLDS16(xsrc[i] + (size_t)(kt_) * BK, smem + (st_) * STAGE + (wave * RPW + 8 * i) * 128);              \
        if constexpr (XR != 0) {                                                                                  \
            if (lane < 32) GLDS16(xsrc[XF] + (size_t)(kt_) * BK, smem + (st_) * STAGE + (wave * RPW + 8 * XF) * 128); \
        }                                                                                                         \
    }

    // fragment read addresses (bytes inside a stage): k-slice 0 reads chunk fgrp, k-slice 1 chunk 4 + fgrp -> offset ^ 64
    const int sw = (fgrp ^ (lane & 7)) << 4;
    const int lw = XB + (wn * 64 + frow) * 128;       // + a * 2048
    const int lx = (wm * TM * 16 + frow) * 128;       // + b * 2048

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // one k-slice: [load segment: fragment reads (+ the caller's DMA / wait work), retired before the barrier] barrier
    //              [MFMA segment at raised priority] barrier
#ifdef CLIPAMD_ABLATION   // tuning builds (scripts/build_variant.sh NAME -DCLIPAMD_ABLATION): p.debug bit 0 no DMA in the loop, bit 1 no MFMA, bit 2 no fragment reads
    const bool ab_nodma = p.debug & 1, ab_nomfma = p.debug & 2, ab_noread = p.debug & 4;
#define AB_DMA(x_) if (!ab_nodma) x_
#define AB_READ if (!ab_noread)
#define AB_MFMA if (!ab_nomfma)
#define AB_INIT _Pragma("unroll") for (int a = 0; a < TN; a++) wf[a] = (h8)(_Float16)1.0f; _Pragma("unroll") for (int b = 0; b < TM; b++) xf[b] = (h8)(_Float16)1.0f;
#else
#define AB_INIT
#define AB_DMA(x_) x_
#define AB_READ
#define AB_MFMA
#endif
#define READ_FRAGS(st_, kk_)                                                                                      \
    AB_READ {                                                                                                     \
        const unsigned char * sb = smem + (st_) * STAGE;                                                          \
        const int so = (kk_) ? (sw ^ 64) : sw;                                                                    \
        _Pragma("unroll") for (int a = 0; a < TN; a++) wf[a] = *(const h8 *)(sb + lw + a * 2048 + so);            \
        _Pragma("unroll") for (int b = 0; b < TM; b++) xf[b] = *(const h8 *)(sb + lx + b * 2048 + so);            \
    }
#define MFMA_SEGMENT()                                                                                            \
    {                                                                                                             \
        wait_lgkm0();                                                                                             \
        raw_barrier();                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        AB_MFMA _Pragma("unroll") for (int a = 0; a < TN; a++)                                                    \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                                        \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);             \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        raw_barrier();                                                                                            \
    }
    // K-tile t lives in stage ST; its segments request tile t + 2 into stage (ST + 2) % 3 (the stage of tile t - 1)
#if CLIPAMD_G8_PHASES == 1
    // one load segment + one 2 x TN x TM MFMA segment per K-tile: half the barriers, twice the cover per segment
#define KTILE(ST, t_)                                                                                             \
    {                                                                                                             \
        h8 wf[TN], xf[TM], wf1[TN], xf1[TM];                                                                      \
        AB_INIT                                                                                                   \
        const bool more = (t_) + 2 < T;                                                                           \
        if (more) {                                                                                               \
            AB_DMA(ISSUE_W((ST + 2) % 3, (t_) + 2));                                                              \
            AB_DMA(ISSUE_X((ST + 2) % 3, (t_) + 2));                                                              \
        }                                                                                                         \
        READ_FRAGS(ST, 0);                                                                                        \
        AB_READ {                                                                                                 \
            const unsigned char * sb = smem + (ST) * STAGE;                                                       \
            _Pragma("unroll") for (int a = 0; a < TN; a++) wf1[a] = *(const h8 *)(sb + lw + a * 2048 + (sw ^ 64)); \
            _Pragma("unroll") for (int b = 0; b < TM; b++) xf1[b] = *(const h8 *)(sb + lx + b * 2048 + (sw ^ 64)); \
        }                                                                                                         \
        if (more) wait_vmcnt<NW + NX>(); else wait_vmcnt<0>();   /* tile t + 1 has landed (this wave's part) */   \
        wait_lgkm0();                                                                                             \
        raw_barrier();                                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        AB_MFMA {                                                                                                 \
            _Pragma("unroll") for (int a = 0; a < TN; a++)                                                        \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                                    \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);         \
            _Pragma("unroll") for (int a = 0; a < TN; a++)                                                        \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                                    \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[a], xf1[b], acc[a][b], 0, 0, 0);       \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        raw_barrier();                                                                                            \
    }
#else
#define KTILE(ST, t_)                                                                                             \
    {                                                                                                             \
        h8 wf[TN], xf[TM];                                                                                        \
        AB_INIT                                                                                                   \
        const bool more = (t_) + (S - 1) < T;                                                                     \
        if (more) {                                                                                               \
            AB_DMA(ISSUE_W((ST + S - 1) % S, (t_) + S - 1));                                                      \
            if constexpr (S == 2) AB_DMA(ISSUE_X((ST + 1) % S, (t_) + 1));                                        \
        }                                                                                                         \
        READ_FRAGS(ST, 0);                                                                                        \
        MFMA_SEGMENT();                                                                                           \
        if constexpr (S == 3) {                                                                                   \
            if (more) wait_vmcnt<NW>();   /* all but the NW requests just made: tile t + 1 has landed (this wave's part) */ \
            else wait_vmcnt<0>();                                                                                 \
            if (more) AB_DMA(ISSUE_X((ST + 2) % S, (t_) + 2));                                                    \
        } else {                                                                                                  \
            wait_vmcnt<0>();              /* two-deep ring: tile t + 1, requested one segment ago, has landed */   \
        }                                                                                                         \
        READ_FRAGS(ST, 1);                                                                                        \
        MFMA_SEGMENT();                                                                                           \
    }
#endif

#ifdef CLIPAMD_G8_TIMING   // tuning builds: per-workgroup phase stamps (shader clock) into the split-K workspace: start, loop, epilogue, end
    unsigned long long * stamp = (unsigned long long *)p.sk_ws + (size_t)blockIdx.x * 8;
    const bool stamper = p.sk_ws && tid == 0;
    if (stamper) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // LayerNorm folded into this GEMM (gemm_common.h): thread t < BM reduces the statistics of row m0 + t to (mean, rstd) before the
    // first request (the counted waits below then see only tile requests); the epilogue picks the values up through LDS
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;
    const bool ln = LNE && p.ln_c != nullptr;
    float2 ln_mine = make_float2(0.f, 1.f);
    if constexpr (LNE) {
        if (ln && tid < BM) ln_mine = ln_row_centred(p, m0 + tid < p.M ? m0 + tid : p.M - 1, n0 == 0);
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0): said with the builtin so that hipcc's own counting restarts from zero
    }
    // Residual epilogue (FFN-down of a ViT-B/32-class batch: 160 x 256 tiles): the tile's f32 residual rows are requested HERE, before the first
    // operand request — older than every LDS-DMA request, so the counted waits of the loop mean what they say — and arrive under the K loop.
    // 80 registers beside the kernel's 172 (two waves per SIMD: 256); the 256-row tile has none to spare and keeps the one-strip-ahead fetch.
#ifdef CLIPAMD_G8_NO_RPRE
    constexpr bool RPRE = false;
#else
    constexpr bool RPRE = EPI == EPI_RESID_F32 && TM <= 5;
#endif
    f4 rpre[RPRE ? TN : 1][RPRE ? TM : 1];
    if constexpr (RPRE) {
#pragma unroll
        for (int a = 0; a < TN; a++) {
            int n = n0 + wn * 64 + a * 16 + fgrp * 4;
            n = n < p.W.N ? n : 0;
#pragma unroll
            for (int b = 0; b < TM; b++) {
                const int m = m0 + wm * TM * 16 + b * 16 + frow;
                rpre[a][b] = *(const f4 *)(p.resid + (size_t)(m < p.M ? m : p.M - 1) * p.ldc + n);
            }
        }
    }
    ISSUE_W(0, 0);
    ISSUE_X(0, 0);
    if (S == 3 && T > 1) {
        ISSUE_W(1, 1);
        ISSUE_X(1, 1);
        wait_vmcnt<NW + NX>();
    } else {
        wait_vmcnt<0>();
    }
    raw_barrier();
#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
    if (wm == 1) raw_barrier();                        // stagger: group 1 runs one segment behind group 0
    for (int t = 0; t < T; t += S) {
        KTILE(0, t);
        if (t + 1 < T) KTILE(1, t + 1);
        if constexpr (S == 3) {
            if (t + 2 < T) KTILE(2, t + 2);
        }
    }
    if (wm == 0) raw_barrier();                        // equalise the barrier count; after it no wave reads the ring any more
#undef KTILE
#undef MFMA_SEGMENT
#undef READ_FRAGS
#undef ISSUE_X
#undef ISSUE_W

    const int nb = n0 + wn * 64, mb = m0 + wm * TM * 16;
    // the lane coordinates are derived again for the epilogue (v_mbcnt, opaque to CSE): the 256 x 256 tile's residual + fold epilogue is at
    // the register limit, and a value kept across the K loop only for the epilogue's sake was spilled to scratch (12 bytes, r04 table)
    unsigned lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int frow_e = (int)(lane_e & 15u), fgrp_e = (int)(lane_e >> 4), lane_i = (int)lane_e;
#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[2] = __builtin_amdgcn_s_memtime();
#endif
    // LDS after the K loop: 8 fp16 staging areas of (TM * 16) rows x 136 bytes, then BM float2 of row statistics
    float2 * ln_rs = (float2 *)(smem + 8 * (TM * 16) * 136);
    if constexpr (LNE) {
        if (ln) ln_rows_publish(ln_rs, ln_mine, tid, BM, [] { __syncthreads(); });
    }
    const float2 * rs_lane = ln_rs + wm * TM * 16 + frow_e;
    half_t * stage = (half_t *)smem + wave * (TM * 16) * 68;
    bool done = false;
    if constexpr (EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16) {
        if (nb + 64 <= p.W.N && (p.ldc & 7) == 0) {    // uniform per wave
            gemm_epilogue_f16_staged<EPI, TN, TM>(p, acc, nb, mb, frow_e, fgrp_e, stage, lane_i, ln, rs_lane);
            done = true;
        }
    }
    if constexpr (RPRE) {
        gemm_epilogue_resid_pre<TN, TM>(p, acc, rpre, nb, mb, frow_e, fgrp_e, stage, lane_i);
        done = true;
    }
    if (!done) gemm_epilogue<EPI, TN, TM>(p, acc, nb, mb, frow_e, fgrp_e, ln, rs_lane, stage, lane_i);
#ifdef CLIPAMD_G8_TIMING
    if (stamper) {
        stamp[3] = __builtin_amdgcn_s_memtime();       // stores issued (not necessarily landed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[5] = __builtin_amdgcn_s_memtime();       // this wave's stores acknowledged
        stamp[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int TM, int EPI>
void launch8(const GemmParams & p, hipStream_t stream) {
    constexpr int BM = 32 * TM;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + 255) / 256;
    constexpr size_t ring = (size_t)(3 * (BM + 256) * 128 <= 160 * 1024 ? 3 : 2) * (BM + 256) * 128;
    // fp16 epilogues stage the tile through LDS; behind the 8 staging areas sit the BM float2 of row statistics of the LayerNorm fold
    constexpr size_t stage_out = (size_t)8 * (TM * 16) * 68 * sizeof(half_t) + (size_t)BM * sizeof(float2);
    constexpr size_t smem = ring > stage_out ? ring : stage_out;
    static_assert(smem <= 160 * 1024, "tile does not fit the LDS");
    static unsigned long long lds_ok = 0;
    opt_in_dynamic_lds(gemm8_kernel<TM, EPI>, smem, lds_ok);
    hipLaunchKernelGGL((gemm8_kernel<TM, EPI>), dim3(tiles_m * tiles_n), dim3(NT8), smem, stream, p);
}

template <int TM>
void launch8_epi(const GemmParams & p, int epi, hipStream_t stream) {
    switch (epi) {
    case EPI_F32: launch8<TM, EPI_F32>(p, stream); break;
    case EPI_F16: launch8<TM, EPI_F16>(p, stream); break;
    case EPI_GELU_F16: launch8<TM, EPI_GELU_F16>(p, stream); break;
    case EPI_QGELU_F16: launch8<TM, EPI_QGELU_F16>(p, stream); break;
    case EPI_RESID_F32: launch8<TM, EPI_RESID_F32>(p, stream); break;
    case EPI_PATCH_F32: launch8<TM, EPI_PATCH_F32>(p, stream); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Weight dequantisation into an fp16 [Npad][Kpad] row-major panel: (q - zero) * d  /  q * d + m in packed fp16, one rounding —
// dequant_wfrag of gemm_common.h, i.e. the very values the fused kernel feeds the MFMA.  A wave covers 8 rows x 64 k:
// lane -> (word j = l & 3, k-block l >> 2 & 1, row l >> 3); it reads one 32-bit word of packed quants (4 lanes = the 16 B
// of a block, 8 rows contiguous in the block-column-major plane) and writes 16 B, so every store instruction writes
// 8 full 128-byte lines.  HBM-bound: reads 0.56-1.06 B, writes 2 B per weight.
// ---------------------------------------------------------------------------------------------
template <int WT>
__global__ void __launch_bounds__(256) dequant_kernel(const DequantJobs jobs) {
    int b = blockIdx.x, ji = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
        if (i + 1 < jobs.n && b >= jobs.blk_end[i]) ji = i + 1;
    if (ji > 0) b -= jobs.blk_end[ji - 1];
    const DevWeight W = jobs.W[ji];
    half_t * out = jobs.out[ji];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblk32 = W.Npad / 32;
    const int kb2 = b / nblk32, nb = b % nblk32;
    const int j = lane & 3, kb = kb2 * 2 + ((lane >> 2) & 1), n = nb * 32 + wave * 8 + (lane >> 3);
    const size_t idx = (size_t)kb * W.Npad + n;
    WFrag<WT> f;
    if constexpr (WT == W_Q8_0) {
        const uint2 q = ((const uint2 *)W.qs)[idx * 4 + j];
        f.q = q.x;
        f.q1 = q.y;
    } else {
        f.q = ((const uint32_t *)W.qs)[idx * 4 + j];
    }
    if constexpr (WT == W_Q5_0 || WT == W_Q5_1) f.h = ((const uint32_t *)W.qh)[idx];
    if constexpr (WT == W_Q4_1 || WT == W_Q5_1) f.dm = ((const h2 *)W.dm)[idx];
    else f.d = ((const half_t *)W.dm)[idx];
    *(h8 *)(out + (size_t)n * W.Kpad + kb * 32 + j * 8) = dequant_wfrag<WT>(f, j);
}

template <int WT>
void launch_dequant_wt(const DequantJobs & jobs, hipStream_t stream) {
    hipLaunchKernelGGL(dequant_kernel<WT>, dim3(jobs.blk_end[jobs.n - 1]), dim3(256), 0, stream, jobs);
}

}  // namespace

// tm: fragments of 16 rows per wave in M (tile = 32 tm x 256): 3, 4, 5 or 8
void launch_gemm8(const GemmParams & p, int epilogue, int tm, hipStream_t stream) {
    switch (tm) {
    case 3: launch8_epi<3>(p, epilogue, stream); break;
    case 4: launch8_epi<4>(p, epilogue, stream); break;
    case 8: launch8_epi<8>(p, epilogue, stream); break;
    default: launch8_epi<5>(p, epilogue, stream); break;
    }
}

// Dequantise up to 4 weights (one layer: q/k/v, out, FFN up, FFN down) of the SAME quantised type with one launch.
void launch_dequant(const DevWeight * const * ws, half_t * const * outs, int n, hipStream_t stream) {
    int i = 0;
    while (i < n) {
        DequantJobs jobs;
        const int wt = ws[i]->wtype;
        int cum = 0;
        while (i < n && jobs.n < 4 && ws[i]->wtype == wt) {
            jobs.W[jobs.n] = *ws[i];
            jobs.out[jobs.n] = outs[i];
            cum += (ws[i]->Kpad / 64) * (ws[i]->Npad / 32);
            jobs.blk_end[jobs.n] = cum;
            jobs.n++;
            i++;
        }
        switch (wt) {
        case W_Q4_0: launch_dequant_wt<W_Q4_0>(jobs, stream); break;
        case W_Q4_1: launch_dequant_wt<W_Q4_1>(jobs, stream); break;
        case W_Q5_0: launch_dequant_wt<W_Q5_0>(jobs, stream); break;
        case W_Q5_1: launch_dequant_wt<W_Q5_1>(jobs, stream); break;
        case W_Q8_0: launch_dequant_wt<W_Q8_0>(jobs, stream); break;
        default: break;   // f16 weights need no panel
        }
    }
}

}  // namespace clipamd
