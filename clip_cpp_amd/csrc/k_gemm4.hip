// k_gemm4.hip — the largest-M GEMM of the hot path for gfx950 (CDNA4): 256 x 256 tiles, FOUR waves of 128 x 128.
//
//   out[M][N] = epilogue( X[M][K] (fp16) · W16[N][K]^T (fp16) + bias ),   tens of thousands of rows (ViT-L/14 at batch 256: 65792)
//
// Same contraction, operands, MFMA instruction (v_mfma_f32_16x16x32_f16) and k order as k_gemm.hip / k_gemm8.hip — bit-identical
// results (tests/test_gpu_kernels.py) — mapped for the regime where those kernels are bound by the LDS pipe, not by the matrix pipe:
//
//   * what a wave reads from LDS per MFMA depends only on its own sub-tile: (rows + cols) / (rows x cols) fragments.  The 8-wave
//     kernel's 128 x 64 (k_gemm8.hip, 256 x 256 tile) reads 12 fragments per 32 MFMAs; 128 x 128 reads 16 per 64: a third less.
//     Per K-tile (256 x 256 x 64) the CU then moves 128 KB LDS -> registers and 64 KB DMA -> LDS against 2062 cycles of MFMA:
//     ~0.75 of the LDS pipe's 128 B/clk instead of ~1.0 (profiles/r02_gemm8_experiments.txt, section 11).
//   * 8 x 8 fragments = 256 fp32 accumulators per lane: one wave per SIMD, the accumulators live in the AGPR half of the 512-entry
//     register file, fragments and addresses in the VGPR half (__launch_bounds__(256, 1)).
//   * no partner wave to hide behind: each wave software-pipelines itself.  A K-tile is two phases of 64 MFMAs; the 16 fragment
//     reads of the NEXT phase (and, in the second phase, the 16 LDS-DMA requests of the tile after next) are interleaved one per
//     4 MFMAs (sched_group_barrier), so the matrix pipe always has ~60 cycles of queued work while the LDS pipe serves the reads.
//   * two-deep ring of whole K-tiles (2 x 64 KB) filled by global_load_lds_dwordx4.  One barrier per K-tile, in the middle: by then
//     every wave has read all of tile t, so tile t+2 may overwrite it, and tile t+1 — requested one whole K-tile (~2000 cycles)
//     earlier — is complete (vmcnt(0): the tile's 16 requests are the only memory operations in flight).
//   * one workgroup per CU, so tiles come in rounds of 256: launch_gemm gives this kernel the tile rows that fill whole rounds and
//     the rest of the rows to a second launch (k_gemm.hip, tile code 256260).
//   * epilogues shared with the other GEMM kernels (gemm_common.h); fp16 outputs staged through the freed ring, 64 columns at a time.
//
// Reference ops replaced: ggml_mul_mat with a weight operand, clip.cpp:1360-1380,1392,1407,1416 (vision) and
// :1079-1095,1112,1127,1136 (text), at batch sizes the reference cannot reach (its arenas cap B at ~13).

#include "gemm_common.h"

namespace clipamd {

namespace {

constexpr int NT4 = 256;

template <int N> __device__ __forceinline__ void wait_vmcnt4() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0_4() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raw_barrier4() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#define GLDS16_4(src_, dst_)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_),                      \
                                     (__attribute__((address_space(3))) void *)(dst_), 16, 0, 0)

template <int EPI>
__global__ void __launch_bounds__(NT4, 1) gemm4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 256, BN = 256, TN = 8, TM = 8;
    constexpr int XB = BM * 128;                      // bytes of the X tile of one stage ([256][64] fp16)
    constexpr int STAGE = XB + BN * 128;              // + the W tile
    constexpr int NR = 8;                             // 1 KB requests (8 rows) per wave per operand per K-tile: 64 rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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

    // LDS-DMA sources: the LDS image of a request is lane-linear (row l>>3, 16-byte position l&7), so the XOR swizzle of the tile
    // (chunk ^= row & 7) is applied to the per-lane SOURCE chunk
    const int prow = lane >> 3;
    const half_t * xsrc[NR];
    const half_t * wsrc[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const int tr = wave * 64 + 8 * i + prow;
        int gm = m0 + tr;
        gm = gm < p.M ? gm : p.M - 1;
        xsrc[i] = p.A + (size_t)gm * p.lda + (((lane & 7) ^ (tr & 7)) << 3);
        int gn = n0 + tr;
        gn = gn < p.W.Npad ? gn : p.W.Npad - 1;
        wsrc[i] = (const half_t *)p.W.w16 + (size_t)gn * p.W.Kpad + (((lane & 7) ^ (tr & 7)) << 3);
    }
#define G4_ISSUE(st_, kt_)                                                                                        \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NR; i++) {                                                          \
            GLDS16_4(wsrc[i] + (size_t)(kt_) * BK, smem + (st_) + XB + (wave * 64 + 8 * i) * 128);                \
            GLDS16_4(xsrc[i] + (size_t)(kt_) * BK, smem + (st_) + (wave * 64 + 8 * i) * 128);                     \
        }                                                                                                         \
    }

    // fragment read addresses (bytes inside a stage): k-slice 0 reads chunk fgrp, k-slice 1 chunk 4 + fgrp -> offset ^ 64
    const int sw = (fgrp ^ (lane & 7)) << 4;
    const int lw = XB + (wn * 128 + frow) * 128;      // + a * 2048
    const int lx = (wm * 128 + frow) * 128;           // + b * 2048
    // LDS-DMA request inside the K loop, as inline asm: hipcc's waitcnt pass treats the builtin (a FLAT-encoded instruction touching two
    // address spaces) as "pending flat" and from then on turns every counted lgkmcnt wait into lgkmcnt(0) — ~100 idle cycles per phase.
    // The loop's only vmcnt wait is the explicit one in front of the barrier.
    const int lds0 = __builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + wave * 64 * 128);
#ifdef G4_ABL_NODMA   // tuning builds (scripts/build_variant.sh): K loop without its LDS-DMA requests / MFMAs / fragment reads (wrong results)
#define G4_DMA(src_, ldsaddr_) asm volatile("" ::"v"(src_), "s"(ldsaddr_) : "memory")
#else
#define G4_DMA(src_, ldsaddr_)                                                                                    \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_), "s"(ldsaddr_) : "memory", "m0")
#endif
#ifdef G4_ABL_NOREAD
#define G4_DO_READ false
#else
#define G4_DO_READ true
#endif
#ifdef G4_ABL_NOMFMA
#define G4_MFMA_TEXT "s_nop 0"
#else
#define G4_MFMA_TEXT "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
#endif

    // One phase = the 64 MFMAs of a k-slice in 16 groups of 4, each group preceded by ONE fragment read for the next phase (and, in a
    // K-tile's second phase, one W + one X LDS-DMA request of the tile after next).  The MFMAs are inline asm with the accumulator
    // tied in place in an AGPR ("+a"): hipcc's own allocation of 256 accumulators un-ties a third of them and shuffles ~200
    // v_accvgpr moves per K-tile through the loop.  The first MFMA of a group carries a memory clobber, which pins the group's read /
    // requests in program order between the groups (the reads stay ordinary loads, so the counted lgkmcnt waits are hipcc's).
    // Read order x[0..7], w[0..7]: the next phase consumes w[a] at its group 2a, i.e. 8+ groups after the read was issued.
#define G4_GROUP(WF_, XF_, g_)                                                                                    \
    {                                                                                                             \
        asm volatile(G4_MFMA_TEXT : "+a"(acc[(4 * (g_)) / 8][(4 * (g_)) % 8])        \
                     : "v"(WF_[(4 * (g_)) / 8]), "v"(XF_[(4 * (g_)) % 8]) : "memory");                           \
        asm volatile(G4_MFMA_TEXT : "+a"(acc[(4 * (g_) + 1) / 8][(4 * (g_) + 1) % 8]) \
                     : "v"(WF_[(4 * (g_) + 1) / 8]), "v"(XF_[(4 * (g_) + 1) % 8]));                              \
        asm volatile(G4_MFMA_TEXT : "+a"(acc[(4 * (g_) + 2) / 8][(4 * (g_) + 2) % 8]) \
                     : "v"(WF_[(4 * (g_) + 2) / 8]), "v"(XF_[(4 * (g_) + 2) % 8]));                              \
        asm volatile(G4_MFMA_TEXT : "+a"(acc[(4 * (g_) + 3) / 8][(4 * (g_) + 3) % 8]) \
                     : "v"(WF_[(4 * (g_) + 3) / 8]), "v"(XF_[(4 * (g_) + 3) % 8]));                              \
    }
    // PHASE(MW, MX: fragments multiplied; RW, RX: fragments read from stage offset rst_ k-slice rkk_; REQ_: request K-tile kt_ into qst_)
#define G4_PHASE(MW_, MX_, RW_, RX_, rst_, rkk_, REQ_, qst_, kt_)                                                 \
    {                                                                                                             \
        const unsigned char * sb = smem + (rst_);                                                                 \
        const int swz_ = (rkk_) ? (sw ^ 64) : sw;                                                                     \
        _Pragma("unroll") for (int g = 0; g < 16; g++) {                                                          \
            if (G4_DO_READ) {                                                                                     \
                if (g < 8) RX_[g] = *(const h8 *)(sb + lx + g * 2048 + swz_);                                       \
                else RW_[g - 8] = *(const h8 *)(sb + lw + (g - 8) * 2048 + swz_);                                   \
            }                                                                                                     \
            if (REQ_) {                                                                                           \
                if (g & 1) G4_DMA(xsrc[g >> 1] + (size_t)(kt_) * BK, lds0 + (qst_) + 8 * (g >> 1) * 128);                           \
                else G4_DMA(wsrc[g >> 1] + (size_t)(kt_) * BK, lds0 + (qst_) + XB + 8 * (g >> 1) * 128);                            \
            }                                                                                                     \
            G4_GROUP(MW_, MX_, g);                                                                                \
        }                                                                                                         \
    }
#define G4_READ(WF_, XF_, st_, kk_)                                                                               \
    {                                                                                                             \
        const unsigned char * sb = smem + (st_);                                                                  \
        const int swz_ = (kk_) ? (sw ^ 64) : sw;                                                                    \
        _Pragma("unroll") for (int a = 0; a < TN; a++) WF_[a] = *(const h8 *)(sb + lw + a * 2048 + swz_);           \
        _Pragma("unroll") for (int b = 0; b < TM; b++) XF_[b] = *(const h8 *)(sb + lx + b * 2048 + swz_);           \
    }

    // CONSUMER half of the LayerNorm fold (round 5; gemm_common.h): thread t reduces the statistics of row m0 + t to (mean - mu, rstd) before
    // the first request and parks the pair in LDS BEHIND the ring (2 KB at 128 KB) — nothing is carried in registers through the K loop, the
    // fp16 epilogues pick the rows' pairs up from there (any of the loop's barriers orders the two).  The producer half (xg + statistics out of
    // the residual epilogue) stays out of this kernel: beside 256 accumulators it spilled 196 registers (profiles/HISTORY.md, r04).
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;
    float2 * const ln_rs = (float2 *)(smem + 2 * STAGE);
    const bool ln = LNE && p.ln_c != nullptr;
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    // the zeroes must be IN the AGPRs here: left alone, hipcc sinks each v_accvgpr_mov to just before the accumulator's first (asm) MFMA,
    // whose SrcC read then races the VALU write — its hazard recogniser inserts the wait states only for MFMAs it can see
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) asm volatile("" : "+a"(acc[a][b]));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    h8 w0[TN], x0[TM], w1[TN], x1[TM];                // fragments of k-slice 0 / k-slice 1

#ifdef CLIPAMD_G8_TIMING   // tuning builds: per-workgroup phase stamps (shader clock) into the split-K workspace: start, loop, epilogue, end
    unsigned long long * stamp = (unsigned long long *)p.sk_ws + (size_t)blockIdx.x * 8;
    const bool stamper = p.sk_ws && tid == 0;
    if (stamper) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // prologue: tiles 0 and 1 requested, tile 0 awaited, its first k-slice read
    G4_ISSUE(0, 0);
    if (T > 1) G4_ISSUE(STAGE, 1);
    if constexpr (LNE) {
        // the statistics round trip runs under the landing of the first K-tiles (in front of the requests it cost every tile ~1 us of nothing)
        if (ln) {
            ln_rs[tid] = ln_row_centred(p, m0 + tid < p.M ? m0 + tid : p.M - 1, n0 == 0 && m0 + tid < p.M);
            __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0), as a builtin: hipcc's own counting restarts from zero (both K-tiles have landed too)
        }
    }
    if (T > 1) wait_vmcnt4<2 * NR>(); else wait_vmcnt4<0>();
    raw_barrier4();
    G4_READ(w0, x0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): the loop is entered with nothing pending (counted waits inside)

    // K-tile t in stage ST:  phase A = MFMAs of k-slice 0 | reads of k-slice 1;  boundary: tile t+1 landed, tile t fully read;
    //                        phase B = MFMAs of k-slice 1 | requests of tile t+2 (into ST) | reads of tile t+1's k-slice 0
    // loop body = second half of K-tile t + first half of K-tile t+1, so that the loop header sits at the full wait (a header in front
    // of a phase makes hipcc's waitcnt pass merge the pending-read state of two paths and wait for everything: ~100 idle cycles)
#define G4_BODY(t_, REQ_)                                                                                         \
    {                                                                                                             \
        __builtin_amdgcn_s_waitcnt(0x0070);            /* vmcnt(0) lgkmcnt(0), as a builtin: hipcc's waitcnt pass sees it */ \
        raw_barrier4();                                /* K-tile t+1 landed; every wave has read all of K-tile t */ \
        G4_PHASE(w1, x1, w0, x0, so ^ STAGE, 0, REQ_, so, (t_) + 2);                                              \
        so ^= STAGE;                                                                                              \
        G4_PHASE(w0, x0, w1, x1, so, 1, false, 0, 0);                                                             \
    }
#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
    int so = 0;                                        // byte offset of the current K-tile's stage
    G4_PHASE(w0, x0, w1, x1, so, 1, false, 0, 0);      // first half of K-tile 0
    int t = 0;
    for (; t + 2 < T; t++) G4_BODY(t, true);           // steady state: K-tile t+2 requested into the stage K-tile t leaves
    for (; t + 1 < T; t++) G4_BODY(t, false);
    // second half of the last K-tile (its fragments are in registers)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int g = 0; g < 16; g++) G4_GROUP(w1, x1, g);
#undef G4_BODY
#undef G4_DMA
#undef G4_PHASE
#undef G4_GROUP
#undef G4_READ
#undef G4_ISSUE
    // (the asm MFMAs are opaque to hipcc's hazard recogniser: let the last ones retire before the accumulators are read)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) asm volatile("" : "+a"(acc[a][b]));   // ... and keep hipcc from reading one earlier than this point

#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[2] = __builtin_amdgcn_s_memtime();
#endif
    const int nb = n0 + wn * 128, mb = m0 + wm * 128;
    // the lane coordinates are derived again for the epilogue (v_mbcnt, opaque to CSE): the K loop uses all 256 VGPRs, and a value kept
    // across it only for the epilogue's sake is spilled to scratch
    unsigned lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int frow_e = (int)(lane_e & 15u), fgrp_e = (int)(lane_e >> 4), lane_i = (int)lane_e;
    // (LayerNorm fold: the consumer half only — rs_lane: this lane's rows of the pairs parked in the prologue.  The producer half stays out of this
    //  kernel, launch_gemm sends producer launches to the 8-wave 256 x 256 tile: in registers it spills 196 beside the accumulators (r04), and re-reading
    //  the stored rows — round 5, commit in profiles/r05_experiments.txt section 11, bit-identical — costs 130-160 us per launch: a workgroup that has its
    //  CU to itself must first wait out the drain of its own 256 KB of stores, which otherwise hides under the next tile's prologue)
    const float2 * rs_lane = ln_rs + wm * 128 + frow_e;
    bool done = false;
    if constexpr (LNE) {
        if (nb + 128 <= p.W.N && (p.ldc & 7) == 0) {
            raw_barrier4();                            // every wave is done with the ring: it becomes the staging area
            half_t * stage = (half_t *)smem + wave * (TM * 16) * 68;
            typedef f4 half_acc_t[4][TM];
            gemm_epilogue_f16_staged<EPI, 4, TM, true>(p, *(half_acc_t *)&acc[0], nb, mb, frow_e, fgrp_e, stage, lane_i, ln, rs_lane);
            gemm_epilogue_f16_staged<EPI, 4, TM, true>(p, *(half_acc_t *)&acc[4], nb + 64, mb, frow_e, fgrp_e, stage, lane_i, ln, rs_lane);
            done = true;
        }
    }
    if (!done) {
        if constexpr (LNE) gemm_epilogue<EPI, TN, TM, true>(p, acc, nb, mb, frow_e, fgrp_e, ln, rs_lane);
        else gemm_epilogue<EPI, TN, TM, false>(p, acc, nb, mb, frow_e, fgrp_e, false, nullptr);
    }
#ifdef CLIPAMD_G8_TIMING
    if (stamper) {
        stamp[3] = __builtin_amdgcn_s_memtime();       // stores issued (not necessarily landed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[5] = __builtin_amdgcn_s_memtime();       // this wave's stores acknowledged
        stamp[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int EPI>
void launch4(const GemmParams & p, hipStream_t stream) {
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.W.N + 255) / 256;
    constexpr size_t smem = (size_t)2 * 512 * 128 + 256 * sizeof(float2);   // the ring (128 KB; the fp16 staging, 4 x 17 KB, lies inside it) + the row statistics of the LayerNorm fold
    static unsigned long long lds_ok = 0;
    opt_in_dynamic_lds(gemm4_kernel<EPI>, smem, lds_ok);
    hipLaunchKernelGGL((gemm4_kernel<EPI>), dim3(tiles_m * tiles_n), dim3(NT4), smem, stream, p);
}

}  // namespace

void launch_gemm4(const GemmParams & p, int epilogue, hipStream_t stream) {
    switch (epilogue) {
    case EPI_F32: launch4<EPI_F32>(p, stream); break;
    case EPI_F16: launch4<EPI_F16>(p, stream); break;
    case EPI_GELU_F16: launch4<EPI_GELU_F16>(p, stream); break;
    case EPI_QGELU_F16: launch4<EPI_QGELU_F16>(p, stream); break;
    case EPI_RESID_F32: launch4<EPI_RESID_F32>(p, stream); break;
    case EPI_PATCH_F32: launch4<EPI_PATCH_F32>(p, stream); break;
    }
}

}  // namespace clipamd
