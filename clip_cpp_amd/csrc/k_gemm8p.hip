// k_gemm8p.hip — PERSISTENT form of the 8-wave large-M GEMM (k_gemm8.hip) for the fp16-output epilogues: the q/k/v projection and
// FFN-up of a batch (reference clip.cpp:1360-1380, 1407-1411; text :1079-1095, 1127-1131), where the one-workgroup-per-CU kernel
// exposed a prologue (first K-tiles in flight: 2.7-3.6 k cycles) and an epilogue (7.7 k cycles of fp16 stores that all 256 CUs hit
// together) around 12 K-tiles of 1.85 k cycles each (profiles/HISTORY.md, r02 phase stamps): a third of every tile.
//
// Same contraction, MFMA instruction, k order and epilogue arithmetic as k_gemm8.hip / k_gemm.hip — bit-identical outputs (tests) —
// but a different life cycle of the workgroup (the epilogue overlapped with the next tile's K loop):
//   * one workgroup per CU walks tiles v, v + G, v + 2G, ... of the XCD-contiguous, n-fastest tile order (G = grid size, a multiple
//     of 8, so a workgroup stays on its XCD's chunk and the 32 workgroups of an XCD work on 32 neighbouring tiles at any time);
//   * the K-tiles of consecutive output tiles form ONE stream through the 3-stage LDS ring: the last two K-tiles of tile i request
//     the first two of tile i + 1, so the ring never drains and there is no prologue after the first tile;
//   * the epilogue is split in two.  Arithmetic (bias / LayerNorm-fold apply / Q scale / GELU, fp16 rounding) runs right behind the
//     tile's last MFMA and leaves the tile as 40 registers of packed fp16 per lane; the accumulators are free again at once.  The
//     STORES of those registers are issued two per K-tile inside the K loop of the NEXT tile, straight from the accumulator layout
//     (8 bytes per lane; the four stores that complete a row's 128-byte line are issued in consecutive slots and merge in L2):
//     stores are fire-and-forget on gfx950 (data is read at issue), so the HBM write of tile i is spread over tile i + 1's MFMA
//     time instead of being a burst between two K loops;
//   * everything that returns data to registers (row statistics of the LayerNorm fold, bias, c vector) is loaded in the arithmetic
//     part, outside the K loop, so the loop's counted vmcnt waits see only LDS-DMA requests and the (older) stores.  The row
//     statistics are reduced per WAVE for its own 80 rows (4 x redundant loads of a few KB) and handed round with ds_bpermute —
//     no workgroup barrier in the epilogue, the two M-groups keep their one-barrier stagger across tiles.
// Needs the weight as an fp16 panel (k_gemm8.hip dequant_kernel; forward.cpp keeps the panels of these two weights resident).  The K
// loop is fully unrolled per output tile (the store slots and the switch of sources are then compile-time positions), so it is
// instantiated per depth: K = 512, 768, 1024, 1280 (the hidden sizes of the CLIP towers); other depths stay on k_gemm8.hip / k_gemm.hip.

#include "gemm_common.h"

namespace clipamd {

namespace {

constexpr int NT8P = 512;
constexpr int GEMM8P_MAX_TILES = 4;

template <int N> __device__ __forceinline__ void p_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void p_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void p_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#define PGLDS16(src_, dst_)                                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_),                      \
                                     (__attribute__((address_space(3))) void *)(dst_), 16, 0, 0)

template <int TM, int EPI, int KT>
__global__ void __launch_bounds__(NT8P, 2) gemm8p_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * TM, BN = 256, TN = 4;
    constexpr int XB = BM * 128;                      // bytes of the X tile of one stage ([BM][64] fp16)
    constexpr int STAGE = XB + BN * 128;              // + the W tile ([256][64] fp16)
    constexpr int RPW = BM / 8;                       // X rows staged per wave per K-tile
    constexpr int XF = RPW / 8, XR = RPW % 8;
    static_assert(XR == 0 || XR == 4, "X rows per wave must be a multiple of 4");
    constexpr int NX = XF + (XR ? 1 : 0);             // X requests per wave per K-tile
    constexpr int NW = 4;                             // W requests per wave per K-tile (32 rows)
    static_assert(3 * STAGE <= 160 * 1024, "three-stage ring does not fit the LDS");
    static_assert(KT >= 6, "at least two K-tiles of slack for the deferred stores");
    static_assert(EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16, "fp16-output epilogues only");
    constexpr int NFR = TN * TM;                      // fragments (packed fp16 register pairs) per lane and tile
    constexpr int FPK = (NFR + KT - 3) / (KT - 2);    // stores per K-tile: the tile is out before the next tile's last two K-tiles
    constexpr int QP = (TM * 16 + 63) / 64;           // passes of 64 rows over the wave's TM * 16 rows (row statistics)
    constexpr int MAXT = GEMM8P_MAX_TILES;            // tiles per workgroup the statistics queue holds (launcher: ceil(tiles / grid) <= MAXT)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fgrp = lane >> 4;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int G = (int)gridDim.x;                     // a multiple of 8 (launcher), or nwg when there are fewer tiles than that
    auto tile_of = [&](int v) {                       // XCD-contiguous chunks, n fastest (as k_gemm8.hip); v % 8 is the same for every tile of a workgroup
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = v & 7, idx = v >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    const int prow = lane >> 3;
    // LDS-DMA sources as 32-bit byte offsets from the two uniform base pointers (7 registers per tile instead of 14 pointers: the tile's
    // packed output of the previous tile lives in registers beside the accumulators).  The LDS image of a piece is lane-linear (row
    // l >> 3, 16-byte position l & 7); the XOR swizzle of the tile (chunk ^= row & 7) is applied to the per-lane SOURCE chunk.
    struct Src { uint32_t x[NX]; uint32_t w[NW]; };
    const char * const Ab = (const char *)p.A;
    const char * const Wb = (const char *)p.W.w16;
    auto sources = [&](int m0, int n0, Src & s) {
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const int tr = wave * RPW + 8 * i + prow;
            int gm = m0 + tr;
            gm = gm < p.M ? gm : p.M - 1;
            s.x[i] = ((uint32_t)gm * (uint32_t)p.lda + (uint32_t)(((lane & 7) ^ (tr & 7)) << 3)) * 2u;
        }
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int tr = wave * 32 + 8 * j + prow;
            int gn = n0 + tr;
            gn = gn < p.W.Npad ? gn : p.W.Npad - 1;
            s.w[j] = ((uint32_t)gn * (uint32_t)p.W.Kpad + (uint32_t)(((lane & 7) ^ (tr & 7)) << 3)) * 2u;
        }
    };
#define P_ISSUE_W(st_, S_, kt_)                                                                                   \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NW; j++)                                                            \
            PGLDS16(Wb + (S_).w[j] + (uint32_t)((kt_) * BK * 2), smem + so[st_] + XB + (wave * 32 + 8 * j) * 128); \
    }
#define P_ISSUE_X(st_, S_, kt_)                                                                                   \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < XF; i++)                                                            \
            PGLDS16(Ab + (S_).x[i] + (uint32_t)((kt_) * BK * 2), smem + so[st_] + (wave * RPW + 8 * i) * 128); \
        if constexpr (XR != 0) {                                                                                  \
            if (lane < 32) PGLDS16(Ab + (S_).x[XF] + (uint32_t)((kt_) * BK * 2), smem + so[st_] + (wave * RPW + 8 * XF) * 128); \
        }                                                                                                         \
    }
    // fragment read addresses (bytes inside a stage): k-slice 0 reads chunk fgrp, k-slice 1 chunk 4 + fgrp -> offset ^ 64
    const int sw = (fgrp ^ (lane & 7)) << 4;
    const int lw = XB + (wn * 64 + frow) * 128;       // + a * 2048
    const int lx = (wm * TM * 16 + frow) * 128;       // + b * 2048

    // byte offsets of the three ring stages as the CURRENT output tile sees them: K-tile t of a tile lives in so[t % 3].  The K-tiles of
    // successive tiles are one stream, so for KT % 3 != 0 the assignment rotates by KT % 3 from tile to tile (uniform values: SGPRs)
    int so[3] = {0, STAGE, 2 * STAGE};
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    uint2 outp[TN][TM];                               // the previous tile, packed fp16, waiting for its store slots
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) outp[a][b] = make_uint2(0u, 0u);

#ifdef CLIPAMD_ABLATION   // tuning builds (scripts/build_variant.sh NAME -DCLIPAMD_ABLATION): p.debug bit 0 no LDS-DMA in the loop, 1 no MFMA, 2 no fragment reads, 3 no stores, 4 no epilogue arithmetic
    const bool ab_nodma = p.debug & 1, ab_nomfma = p.debug & 2, ab_noread = p.debug & 4, ab_nostore = p.debug & 8, ab_noepi = p.debug & 16;
#define PAB_DMA if (!ab_nodma)
#define PAB_READ if (!ab_noread)
#define PAB_MFMA if (!ab_nomfma)
#define PAB_STORE if (!ab_nostore)
#define PAB_EPI if (!ab_noepi)
#define PAB_INIT _Pragma("unroll") for (int a = 0; a < TN; a++) wf[a] = (h8)(_Float16)1.0f; _Pragma("unroll") for (int b = 0; b < TM; b++) xf[b] = (h8)(_Float16)1.0f;
#else
#define PAB_DMA
#define PAB_READ
#define PAB_MFMA
#define PAB_STORE
#define PAB_EPI
#define PAB_INIT
#endif
#define P_READ_FRAGS(st_, kk_)                                                                                    \
    PAB_READ {                                                                                                    \
        const unsigned char * sb = smem + so[st_];                                                                \
        const int so = (kk_) ? (sw ^ 64) : sw;                                                                    \
        _Pragma("unroll") for (int a = 0; a < TN; a++) wf[a] = *(const h8 *)(sb + lw + a * 2048 + so);            \
        _Pragma("unroll") for (int b = 0; b < TM; b++) xf[b] = *(const h8 *)(sb + lx + b * 2048 + so);            \
    }
#define P_MFMA_SEGMENT()                                                                                          \
    {                                                                                                             \
        p_wait_lgkm0();                                                                                           \
        p_barrier();                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        PAB_MFMA _Pragma("unroll") for (int a = 0; a < TN; a++)                                                   \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                                        \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);             \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        p_barrier();                                                                                              \
    }

#ifdef P8_NOLN
    constexpr bool ln = false;
#else
    const bool ln = p.ln_c != nullptr;
#endif
    const int N = p.W.N;
    half_t * const outb = (half_t *)p.out;

    int v = (int)blockIdx.x;
    int tile = tile_of(v);
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    Src cur;
    sources(m0, n0, cur);
    // stores of the previous tile: row pointer of fragment row 0 (this lane's output row mb + frow, columns nb + fgrp * 4 ...) and its coordinates
    half_t * pout = outb;
    int prow_m = p.M, pnb = 0;                        // this lane's output row of the row block being stored (M: nothing to store yet)
    const size_t ld16 = (size_t)16 * p.ldc;
    bool have_prev = false;

    // Row statistics of the LayerNorm fold for EVERY tile of this workgroup (<= MAXT), reduced once, up front, while no accumulator is
    // live and the ring is still empty: the 512 threads share the <= MAXT * BM rows (one or two rows each, sequential Chan merge per row in
    // the canonical order: ln_row_final), park (mean - mu, rstd) in LDS, and every wave then picks up the pairs of its TM * 16 rows per
    // tile — lane l <-> row (q * 64 + l) of pass q: 4 registers per tile carried through the K loops (a queue shifted by one tile after each
    // epilogue: static register indexing).  Reduced inside each epilogue instead, the loads and merge temporaries spilled 76 registers.
    float2 stq[MAXT][QP];
#pragma unroll
    for (int k = 0; k < MAXT; k++)
#pragma unroll
        for (int q = 0; q < QP; q++) stq[k][q] = make_float2(0.f, 1.f);
    if (ln) {
        float2 * const park = (float2 *)smem;         // [MAXT][BM]
#pragma unroll
        for (int pass = 0; pass < (MAXT * BM + NT8P - 1) / NT8P; pass++) {
            const int j = pass * NT8P + tid, k = j / BM, r = j - k * BM;
            const int vk = v + k * G;
            if (k < MAXT && vk < nwg) {
                const int tk = tile_of(vk);
                const int m = (tk / tiles_n) * BM + r;
                park[j] = ln_row_centred(p, m < p.M ? m : p.M - 1, (tk % tiles_n) == 0 && m < p.M);   // the first column tile leaves the row means for the next producer (mu_out)
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXT; k++) {
            if (v + k * G < nwg) {
#pragma unroll
                for (int q = 0; q < QP; q++) {
                    int r = q * 64 + lane;
                    r = r < TM * 16 ? r : (q * 64 + (lane & 15) < TM * 16 ? q * 64 + (lane & 15) : TM * 16 - 1);   // a partial last pass repeats its rows in every 16-lane group
                    stq[k][q] = park[k * BM + wm * TM * 16 + r];
                }
            }
        }
        __syncthreads();                              // (the ring is about to be filled)
    }
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0), said with the builtin: hipcc's own counting restarts from zero

    // first tile only: its first two K-tiles, then the one-barrier stagger of the two M-groups
    P_ISSUE_W(0, cur, 0);
    P_ISSUE_X(0, cur, 0);
    P_ISSUE_W(1, cur, 1);
    P_ISSUE_X(1, cur, 1);
    p_wait_vmcnt<NW + NX>();
    p_barrier();
    if (wm == 1) p_barrier();

    for (;;) {
        const int vn = v + G;
        const bool has_next = vn < nwg;               // (uniform)
        int m0n = m0, n0n = n0;
#pragma unroll
        for (int t = 0; t < KT; t++) {
            const int ST = t % 3, SR = (t + 2) % 3;   // this K-tile's stage; the stage K-tile t + 2 of the stream is requested into
            if (t == KT - 2) {                         // K-tile KT - 1 was requested in the previous step: from here on the sources are the NEXT tile's
                if (has_next) {
                    const int tn_ = tile_of(vn);
                    m0n = (tn_ / tiles_n) * BM; n0n = (tn_ % tiles_n) * BN;
                    sources(m0n, n0n, cur);
                }
            }
            const bool more = (t + 2 < KT) || has_next;
            h8 wf[TN], xf[TM];
            PAB_INIT
            if (more) {
                PAB_DMA P_ISSUE_W(SR, cur, t + 2 < KT ? t + 2 : t + 2 - KT);
            }
            P_READ_FRAGS(ST, 0);
            P_MFMA_SEGMENT();
            // everything but the NW requests just made has landed: K-tile t + 1 of the stream (this wave's part) — and the stores of the
            // previous K-tile's slot, which are older than those requests (issued BEFORE the X requests below for exactly this reason:
            // the immediate of this wait must not depend on whether a slot had something to store)
            if (more) p_wait_vmcnt<NW>(); else p_wait_vmcnt<0>();
            PAB_STORE if (have_prev) {
#pragma unroll
                for (int f = t * FPK; f < (t + 1) * FPK && f < NFR; f++) {
                    const int b = f / TN, a = f % TN;  // the four column strips of a row block in consecutive slots: full 128-byte lines meet in L2
                    if (prow_m < p.M && pnb + a * 16 + fgrp * 4 < N) *(uint2 *)(pout + a * 16) = outp[a][b];
                    if (a == TN - 1) { pout += ld16; prow_m += 16; }      // next row block (running pointer: no per-fragment address products)
                }
            }
            if (more) {
                PAB_DMA P_ISSUE_X(SR, cur, t + 2 < KT ? t + 2 : t + 2 - KT);
            }
            P_READ_FRAGS(ST, 1);
            P_MFMA_SEGMENT();
        }
        // ---- arithmetic half of the epilogue: acc -> packed fp16 registers.  The only place with loads that return to registers. ----
        const int nb = n0 + wn * 64, mb = m0 + wm * TM * 16;
        PAB_EPI {
            f4 biasv[TN], cv[TN];                      // requested first: in flight under the statistics passes
#pragma unroll
            for (int a = 0; a < TN; a++) {
                int n = nb + a * 16 + fgrp * 4;
                n = n < N ? n : 0;                     // (clamped: columns past N are never stored)
                biasv[a] = p.bias ? *(const f4 *)(p.bias + n) : (f4){0.f, 0.f, 0.f, 0.f};
                cv[a] = ln ? *(const f4 *)(p.ln_c + n) : (f4){0.f, 0.f, 0.f, 0.f};
            }
            float2 mr[TM];
#pragma unroll
            for (int b = 0; b < TM; b++) {
                if (ln) {
                    const int q = (b * 16) / 64, src = (b * 16) % 64 + frow;
                    mr[b] = make_float2(__shfl(stq[0][q].x, src), __shfl(stq[0][q].y, src));
                } else {
                    mr[b] = make_float2(0.f, 1.f);
                }
            }
#pragma unroll
            for (int b = 0; b < TM; b++) {
#pragma unroll
                for (int a = 0; a < TN; a++) {
                    const int n = nb + a * 16 + fgrp * 4;
                    f4 val = ln_apply(ln, mr[b], acc[a][b], cv[a], biasv[a]);
                    if constexpr (EPI == EPI_F16) {
                        if (n < p.qcols) val = val * p.qscale;
                    } else if constexpr (EPI == EPI_GELU_F16) {
#pragma unroll
                        for (int r = 0; r < 4; r++) val[r] = gelu_tanh(val[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) val[r] = gelu_quick(val[r]);
                    }
                    const h2 lo = (h2){(_Float16)val[0], (_Float16)val[1]};
                    const h2 hi = (h2){(_Float16)val[2], (_Float16)val[3]};
                    outp[a][b] = make_uint2(h2u(lo), h2u(hi));
                    acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        prow_m = mb + frow; pnb = nb;
        pout = outb + (size_t)(mb + frow) * p.ldc + nb + fgrp * 4;
        have_prev = true;
        if (!has_next) break;
#pragma unroll
        for (int k = 0; k + 1 < MAXT; k++)
#pragma unroll
            for (int q = 0; q < QP; q++) stq[k][q] = stq[k + 1][q];
        v = vn; m0 = m0n; n0 = n0n;
        if constexpr (KT % 3 == 1) { const int t0 = so[0]; so[0] = so[1]; so[1] = so[2]; so[2] = t0; }
        if constexpr (KT % 3 == 2) { const int t0 = so[0]; so[0] = so[2]; so[2] = so[1]; so[1] = t0; }
    }
    // the last tile of this workgroup: all of its stores at once
    PAB_STORE
#pragma unroll
    for (int b = 0; b < TM; b++) {
#pragma unroll
        for (int a = 0; a < TN; a++)
            if (prow_m < p.M && pnb + a * 16 + fgrp * 4 < N) *(uint2 *)(pout + a * 16) = outp[a][b];
        pout += ld16; prow_m += 16;
    }
    if (wm == 0) p_barrier();                          // equalise the barrier count of the two M-groups
#undef P_MFMA_SEGMENT
#undef P_READ_FRAGS
#undef P_ISSUE_X
#undef P_ISSUE_W
}

template <int TM, int EPI, int KT>
bool launch8p(const GemmParams & p, hipStream_t stream) {
    constexpr int BM = 32 * TM;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + 255) / 256;
    constexpr size_t smem = (size_t)3 * (BM + 256) * 128;
    static_assert(smem <= 160 * 1024, "tile does not fit the LDS");
    static unsigned long long lds_ok = 0;
    opt_in_dynamic_lds(gemm8p_kernel<TM, EPI, KT>, smem, lds_ok);
    int ncu = 256;
    {
        static int cus[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!cus[dev & 63]) {
            hipDeviceProp_t prop;
            cus[dev & 63] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        ncu = cus[dev & 63];
    }
    const int nwg = tiles_m * tiles_n;
    int grid = ncu & ~7;                               // one workgroup per CU; a multiple of 8 keeps a workgroup on its XCD's chunk of the tile order
    if (grid < 8) grid = 8;
    if (nwg <= grid) grid = nwg;
    if ((nwg + grid - 1) / grid > GEMM8P_MAX_TILES) return false;    // more tiles per workgroup than the statistics queue holds: the caller's other kernels
    hipLaunchKernelGGL((gemm8p_kernel<TM, EPI, KT>), dim3(grid), dim3(NT8P), smem, stream, p);
    return true;
}

template <int TM, int KT>
bool launch8p_epi(const GemmParams & p, int epi, hipStream_t stream) {
    switch (epi) {
    case EPI_F16: return launch8p<TM, EPI_F16, KT>(p, stream);
    case EPI_GELU_F16: return launch8p<TM, EPI_GELU_F16, KT>(p, stream);
    case EPI_QGELU_F16: return launch8p<TM, EPI_QGELU_F16, KT>(p, stream);
    }
    return false;
}

}  // namespace

static bool launch8p_depth(const GemmParams & p, int epilogue, hipStream_t stream);

// the persistent kernel takes this launch: fp16-output epilogue, fp16 panel in p.W.w16, K depth one of the instantiated ones
bool gemm8p_supported(int Kpad, int epilogue) {
    const bool f16out = epilogue == EPI_F16 || epilogue == EPI_GELU_F16 || epilogue == EPI_QGELU_F16;
    return f16out && (Kpad == 512 || Kpad == 768 || Kpad == 1024 || Kpad == 1280);
}

static unsigned long long g_gemm8p_launches = 0;      // (test hook: proves that a launch really took this kernel and not its fallback)
unsigned long long gemm8p_launch_count() { return g_gemm8p_launches; }

bool launch_gemm8p(const GemmParams & p, int epilogue, hipStream_t stream) {
    if (!gemm8p_supported(p.W.Kpad, epilogue) || p.W.wtype != W_F16 || !p.W.w16) return false;
    if ((p.ldc & 3) != 0) return false;               // 8-byte stores
    if ((unsigned long long)p.M * (unsigned)p.lda * 2ull >= (1ull << 32) || (unsigned long long)p.W.Npad * (unsigned)p.W.Kpad * 2ull >= (1ull << 32)) return false;   // 32-bit source offsets
    const bool ok = launch8p_depth(p, epilogue, stream);
    if (ok) g_gemm8p_launches++;
    return ok;
}

static bool launch8p_depth(const GemmParams & p, int epilogue, hipStream_t stream) {
    switch (p.W.Kpad / BK) {
    case 8: return launch8p_epi<5, 8>(p, epilogue, stream);        // K = 512: ViT-B/32 text tower
    case 12: return launch8p_epi<5, 12>(p, epilogue, stream);      // K = 768: ViT-B/32 vision, ViT-L/14 text
    case 16: return launch8p_epi<5, 16>(p, epilogue, stream);      // K = 1024: ViT-L/14 vision, ViT-H/14 text
    case 20: return launch8p_epi<5, 20>(p, epilogue, stream);      // K = 1280: ViT-H/14 vision
    }
    return false;
}

}  // namespace clipamd
