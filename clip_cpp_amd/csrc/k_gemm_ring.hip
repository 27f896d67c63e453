// k_gemm_ring.hip — the mid-M weight GEMM of the hot path for gfx950 (CDNA4): 64 activation rows per workgroup, a RING of K-tiles.
//
//   out[M][N] = epilogue( X[M][K] (fp16) · W[N][K]^T (f16 | q4_0 | q4_1 | q5_0 | q5_1 | q8_0) + bias ),   ~65 … ~2500 rows
//   (a batch of 2-48 ViT-B/32 images, a batch of texts, one ViT-L/14 image)
//
// Why a third tiled kernel.  With a few hundred to a few thousand rows there are too few 128/160-row tiles to fill 256 CUs, so
// k_gemm.hip runs these problems on 64 x 64 tiles (+ split-K) — and there its K-step is LATENCY bound: two LDS buffers, the
// LDS-DMA of tile k+1 is requested at the top of step k and awaited (vmcnt(0)) at its end, and a step of a 64 x 64 tile holds only
// 8 MFMAs per wave (~130 cycles) against ~1 us from request to landing.  Measured (r02): 0.5-1 us per K-step, 11-29 us per GEMM at
// 1600 rows (ViT-B/32 batch 32: 9-12 % of the MFMA peak).  What this kernel changes:
//
//   * a ring of NS = 3-4 whole K-tiles in LDS; tile k+NS-1 is requested while tile k is multiplied, and the only wait is a
//     COUNTED s_waitcnt vmcnt((NS-2) x requests-per-tile) in front of the step's single raw s_barrier: a request has NS-2 whole
//     steps to land.  Every request is an LDS-DMA (global_load_lds_dword / _dwordx4) issued from inline asm with an SGPR base
//     and a per-lane 32-bit offset, so the loop has no VGPR-returning global load at all and hipcc's own wait insertion is not
//     involved (its counting breaks down around LDS-DMA: k_gemm4.hip);
//   * block-quantised weights are staged RAW: the tile's slab of the block-column-major planes (packed quants, fifth bits,
//     scales; kernels.h) is copied plane by plane into the stage — 4.5-8.5 bits per weight instead of 16 — and each wave builds
//     its MFMA A fragments straight from it (one ds_read_b32 of packed quants + the block scale per fragment, dequant_wfrag in
//     registers: the arithmetic of every other GEMM kernel here, so the fp16 operand values are identical);
//   * the activation tile (64 rows x 64 k, 8 KB) is then most of a stage: a 64 x 128 tile of q4_0 weights moves 12.6 KB per K-step
//     where the 64 x 128 tile of k_gemm.hip moved 24 KB;
//   * same MFMA (v_mfma_f32_16x16x32_f16), same k order, same epilogues (gemm_common.h) and the same deterministic split-K
//     hand-off as k_gemm.hip: without split-K the results are bit-identical to the other tiled kernels (tests).
//
// Tile codes (kernels.h launch_gemm): 65064 / 65128 = 64 rows x 64 / 128 weight rows (optionally ksplit * 1000000 + code).
// Reference ops replaced: ggml_mul_mat with a weight operand, clip.cpp:1360-1380,1392,1407,1416 (vision), :1079-1095,1112,1127,1136 (text).

#include "gemm_common.h"

namespace clipamd {

namespace {

constexpr int RBM = 64;                  // activation rows per tile

template <int N> __device__ __forceinline__ void ring_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// LDS-DMA request: LDS address = m0 + lane * size, global address = SGPR base + per-lane VGPR byte offset
#define RING_DMA16(voff_, base_, lds_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_), "s"(base_), "s"(lds_) : "memory", "m0")
#define RING_DMA4(voff_, base_, lds_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff_), "s"(base_), "s"(lds_) : "memory", "m0")

// Stage layout (bytes).  X: [64][64] fp16, 16-byte chunks XOR-swizzled (chunk ^= row & 7) as in k_gemm.hip.
// Quantised W: the two k-blocks of the K-tile, plane by plane, each plane [2][BN][E] exactly as it lies in HBM (E = bytes per block).
template <int WT, int BN, int NW> struct RingLayout {
    static constexpr int QE = WT == W_F16 ? 0 : (WT == W_Q8_0 ? 32 : 16);                    // packed quants per block
    static constexpr int HE = (WT == W_Q5_0 || WT == W_Q5_1) ? 4 : 0;                        // fifth bits
    static constexpr int DE = WT == W_F16 ? 0 : ((WT == W_Q4_1 || WT == W_Q5_1) ? 4 : 2);    // scale (+ min)
    static constexpr int XOFF = 0;
    static constexpr int QOFF = RBM * 128;
    static constexpr int HOFF = QOFF + (WT == W_F16 ? BN * 128 : 2 * BN * QE);               // f16: the [BN][64] fp16 tile sits at QOFF
    static constexpr int DOFF = HOFF + 2 * BN * HE;
    static constexpr int PADOFF = ((DOFF + 2 * BN * DE) + 255) & ~255;                       // 1 KB scratch: where padding requests land
    static constexpr int STAGE = PADOFF + 1024;
    // ring depth: 3-4 stages, so that two or three workgroups share a CU (a deeper ring with one workgroup per CU was slower)
    static constexpr int NS = 4 * STAGE <= 72 * 1024 ? 4 : 3;
    static_assert(NS * STAGE <= 80 * 1024, "ring does not fit two workgroups per CU");
    // requests (one wave instruction each) per K-tile
    static constexpr int JX = RBM / 8;                                                       // 1 KB pieces of the X tile
    static constexpr int JQ = WT == W_F16 ? BN / 8 : 2 * BN * QE / 1024;                     // dwordx4
    static constexpr int JH = 2 * BN * HE / 256;                                             // dword
    static constexpr int JD = 2 * BN * DE / 256;                                             // dword
    static constexpr int NJ = JX + JQ + JH + JD;
    // requests per wave per K-tile (every plane padded to whole rounds of the NW waves)
    static constexpr int COUNT = (JX + NW - 1) / NW + (JQ + NW - 1) / NW + (JH + NW - 1) / NW + (JD + NW - 1) / NW;
};

// NW waves (4 or 8) as WN (along the weight rows) x NW / WN (along the activation rows).  One wave per SIMD issues at most one
// instruction every ~4-5 cycles whatever its kind (a 64 x 256 step is ~330 instructions per wave with 4 waves: ~1900 cycles, measured,
// against 512 cycles of MFMA); with 8 waves a SIMD holds two waves of the workgroup and their VALU / LDS / MFMA / SALU issue overlaps.
template <int WT, int BN, int NW, int WN, int EPI>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) gemm_ring_kernel(const GemmParams p) {
    using LY = RingLayout<WT, BN, NW>;
    constexpr int NTR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WM = NW / WN;
    constexpr int TN = BN / WN / 16, TM = RBM / WM / 16;
    constexpr int NS = LY::NS, STAGE = LY::STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int frow = lane & 15, fgrp = lane >> 4;

    const int tiles_m = (p.M + RBM - 1) / RBM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    const int ksplit = p.ksplit;
    const int nwg = tiles_m * tiles_n * ksplit;
    int bid = blockIdx.x;
    {   // XCD-contiguous chunks, (split, n) fastest: the workgroups that share an activation row-panel sit on one XCD / L2
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_id = bid / ksplit;
    const int split = bid - tile_id * ksplit;
    const int tile_n = tile_id % tiles_n, tile_m = tile_id / tiles_n;
    const int m0 = tile_m * RBM, n0 = tile_n * BN;
    const int nk_all = p.W.Kpad / BK;
    const int kbeg = (split * nk_all) / ksplit;
    const int nk = ((split + 1) * nk_all) / ksplit - kbeg;
    const int last = nk - 1;

    // ---- this wave's requests of a K-tile.  Plane by plane (X pieces, packed quants, fifth bits, scales) job j = slot * NW + wave;
    // a plane whose job count is not a multiple of 4 is padded with copies of its own jobs that land in a scratch area of the stage
    // (PADOFF), so that EVERY wave issues the same number of requests per K-tile and nothing in the K loop branches on the wave id:
    // a taken scalar branch costs 50-70 cycles here, and a 4-way switch on the wave in front of the barrier made a K-step ~250 cycles
    // longer than the barrier itself (scripts/ubench/barrier_bench.hip: 49 vs 261 cycles per iteration).
    // Per request: SGPR base (advanced by `stride` bytes per K-tile), per-lane byte offset, LDS offset inside the stage.
    constexpr int SX = (LY::JX + NW - 1) / NW, SQ = (LY::JQ + NW - 1) / NW, SH = (LY::JH + NW - 1) / NW, SD = (LY::JD + NW - 1) / NW;
    static_assert(LY::JX % NW == 0, "the X pieces divide evenly over the waves");
    const unsigned char * xb = (const unsigned char *)(p.A + (size_t)m0 * p.lda + (size_t)kbeg * BK);
    unsigned xv[SX];
#pragma unroll
    for (int i = 0; i < SX; i++) {
        // X piece J: tile rows 8J .. 8J+7; lane l fetches row l>>3, source chunk (l&7) ^ (row&7) (the LDS image is lane-linear)
        const int J = i * NW + wave, prow = lane >> 3;
        int gm = m0 + J * 8 + prow;
        gm = gm < p.M ? gm : p.M - 1;
        xv[i] = (unsigned)(gm - m0) * (unsigned)p.lda * 2u + (unsigned)(((lane & 7) ^ prow) << 4);
    }
    const unsigned char * qb;
    unsigned qv[SQ > 0 ? SQ : 1], qstride;
    int ql[SQ > 0 ? SQ : 1];
    if constexpr (WT == W_F16) {
        qb = (const unsigned char *)((const half_t *)p.W.w16 + (size_t)n0 * p.W.Kpad + (size_t)kbeg * BK);
        qstride = BK * 2;
#pragma unroll
        for (int i = 0; i < SQ; i++) {
            const int j0 = i * NW + wave, j = j0 < LY::JQ ? j0 : j0 % LY::JQ, prow = lane >> 3;
            int gn = n0 + j * 8 + prow;
            gn = gn < p.W.Npad ? gn : p.W.Npad - 1;
            qv[i] = (unsigned)(gn - n0) * (unsigned)p.W.Kpad * 2u + (unsigned)(((lane & 7) ^ prow) << 4);
            ql[i] = j0 < LY::JQ ? LY::QOFF + j * 1024 : LY::PADOFF;
        }
    } else {
        constexpr int E = LY::QE;
        qb = (const unsigned char *)p.W.qs + ((size_t)kbeg * 2 * p.W.Npad + n0) * E;
        qstride = (unsigned)(2 * p.W.Npad * E);
#pragma unroll
        for (int i = 0; i < SQ; i++) {
            const int j0 = i * NW + wave, j = j0 < LY::JQ ? j0 : j0 % LY::JQ;
            const int o = (j * 64 + lane) * 16;          // byte of the [2][BN][E] stage image this lane fills
            const int kb = o / (BN * E), w = o % (BN * E);
            int row = n0 + w / E;
            row = row < p.W.Npad ? row : p.W.Npad - 1;
            qv[i] = (unsigned)((kb * p.W.Npad + (row - n0)) * E + w % E);
            ql[i] = j0 < LY::JQ ? LY::QOFF + j * 1024 : LY::PADOFF;
        }
    }
    const unsigned char * hb = nullptr;
    unsigned hv[SH > 0 ? SH : 1], hstride = 0;
    int hl[SH > 0 ? SH : 1];
    if constexpr (LY::HE != 0) {
        hb = (const unsigned char *)p.W.qh + ((size_t)kbeg * 2 * p.W.Npad + n0) * 4;
        hstride = (unsigned)(2 * p.W.Npad * 4);
#pragma unroll
        for (int i = 0; i < SH; i++) {
            const int j0 = i * NW + wave, j = j0 < LY::JH ? j0 : j0 % LY::JH;
            const int o = (j * 64 + lane) * 4;
            const int kb = o / (BN * 4), w = o % (BN * 4);
            int row = n0 + w / 4;
            row = row < p.W.Npad ? row : p.W.Npad - 1;
            hv[i] = (unsigned)((kb * p.W.Npad + (row - n0)) * 4);
            hl[i] = j0 < LY::JH ? LY::HOFF + j * 256 : LY::PADOFF;
        }
    }
    const unsigned char * db = nullptr;
    unsigned dv[SD > 0 ? SD : 1], dstride = 0;
    int dl[SD > 0 ? SD : 1];
    if constexpr (LY::DE != 0) {
        constexpr int E = LY::DE;
        db = (const unsigned char *)p.W.dm + ((size_t)kbeg * 2 * p.W.Npad + n0) * E;
        dstride = (unsigned)(2 * p.W.Npad * E);
#pragma unroll
        for (int i = 0; i < SD; i++) {
            const int j0 = i * NW + wave, j = j0 < LY::JD ? j0 : j0 % LY::JD;
            const int o = (j * 64 + lane) * 4;
            const int kb = o / (BN * E), w = o % (BN * E);
            int row = n0 + w / E;                        // (E = 2: this lane carries rows row, row + 1)
            row = row + 4 / E <= p.W.Npad ? row : p.W.Npad - 4 / E;
            dv[i] = (unsigned)((kb * p.W.Npad + (row - n0)) * E);
            dl[i] = j0 < LY::JD ? LY::DOFF + j * 256 : LY::PADOFF;
        }
    }
    const int lds0 = __builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
#define RING_ISSUE(st_, kt_)                                                                                      \
    {                                                                                                             \
        const int sb_ = lds0 + (st_) * STAGE;                                                                     \
        const unsigned t_ = (unsigned)(kt_);                                                                      \
        {                                                                                                         \
            const unsigned char * b_ = xb + (size_t)(t_ * (unsigned)(BK * 2));                                    \
            _Pragma("unroll") for (int i = 0; i < SX; i++) RING_DMA16(xv[i], b_, sb_ + LY::XOFF + (i * NW + wave) * 1024); \
        }                                                                                                         \
        {                                                                                                         \
            const unsigned char * b_ = qb + (size_t)(t_ * qstride);                                               \
            _Pragma("unroll") for (int i = 0; i < SQ; i++) RING_DMA16(qv[i], b_, sb_ + ql[i]);                    \
        }                                                                                                         \
        if constexpr (LY::HE != 0) {                                                                              \
            const unsigned char * b_ = hb + (size_t)(t_ * hstride);                                               \
            _Pragma("unroll") for (int i = 0; i < SH; i++) RING_DMA4(hv[i], b_, sb_ + hl[i]);                     \
        }                                                                                                         \
        if constexpr (LY::DE != 0) {                                                                              \
            const unsigned char * b_ = db + (size_t)(t_ * dstride);                                               \
            _Pragma("unroll") for (int i = 0; i < SD; i++) RING_DMA4(dv[i], b_, sb_ + dl[i]);                     \
        }                                                                                                         \
    }

    // ---- fragment addresses (bytes inside a stage)
    const int xrow0 = wm * (RBM / WM) + frow;          // + b * 16
    const int wrow0 = wn * (BN / WN) + frow;           // + a * 16
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // One K-tile.  A wave's MFMAs of one weight fragment (TM of them) are each followed, in program order and fenced by
    // sched_barrier(0), by a share of the NEXT fragment's dequantisation (dequant_wpair: 5-7 packed-fp16 / integer VALU ops per pair
    // of weights), so the VALU work issues while the matrix pipe is busy: written as "dequantise a fragment, then multiply it" the
    // wave alternated ~80 cycles of VALU with ~64 cycles of MFMA issue and a 64 x 256 step took ~1900 cycles, one wave per SIMD
    // (profiles/r02_second_session_experiments.txt section 3).  Operands of k-slice 1 are read from LDS under k-slice 0's MFMAs.
#define RING_RAW(kk_, a_, dst_)                                                                                   \
    {                                                                                                             \
        const int r_ = wrow0 + (a_) * 16;                                                                         \
        if constexpr (WT == W_Q8_0) {                                                                             \
            const uint2 q_ = *(const uint2 *)(sb + LY::QOFF + ((kk_) * BN + r_) * 32 + fgrp * 8);                 \
            dst_.q = q_.x; dst_.q1 = q_.y;                                                                        \
        } else {                                                                                                  \
            dst_.q = *(const uint32_t *)(sb + LY::QOFF + ((kk_) * BN + r_) * 16 + fgrp * 4);                      \
        }                                                                                                         \
        if constexpr (WT == W_Q5_0 || WT == W_Q5_1) dst_.h = *(const uint32_t *)(sb + LY::HOFF + ((kk_) * BN + r_) * 4); \
        if constexpr (WT == W_Q4_1 || WT == W_Q5_1) dst_.dm = *(const h2 *)(sb + LY::DOFF + ((kk_) * BN + r_) * 4); \
        else dst_.d = *(const half_t *)(sb + LY::DOFF + ((kk_) * BN + r_) * 2);                                   \
    }
#define RING_COMPUTE(st_)                                                                                         \
    {                                                                                                             \
        const unsigned char * sb = smem + (st_) * STAGE;                                                          \
        h8 xf[2][TM];                                                                                             \
        if constexpr (WT == W_F16) {                                                                              \
            h8 wf[2][TN];                                                                                         \
            _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                                    \
                _Pragma("unroll") for (int a = 0; a < TN; a++) wf[kk][a] = *(const h8 *)(sb + LY::QOFF + 2 * lds_off(wrow0 + a * 16, kk * 4 + fgrp)); \
                _Pragma("unroll") for (int b = 0; b < TM; b++) xf[kk][b] = *(const h8 *)(sb + LY::XOFF + 2 * lds_off(xrow0 + b * 16, kk * 4 + fgrp)); \
            }                                                                                                     \
            _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                                      \
                _Pragma("unroll") for (int a = 0; a < TN; a++)                                                    \
                    _Pragma("unroll") for (int b = 0; b < TM; b++)                                                \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][a], xf[kk][b], acc[a][b], 0, 0, 0); \
        } else {                                                                                                  \
            WFrag<WT> rf[2][TN];                                                                                  \
            _Pragma("unroll") for (int a = 0; a < TN; a++) RING_RAW(0, a, rf[0][a]);                              \
            _Pragma("unroll") for (int b = 0; b < TM; b++) xf[0][b] = *(const h8 *)(sb + LY::XOFF + 2 * lds_off(xrow0 + b * 16, fgrp)); \
            uint32_t cur[4], nxt[4];                                                                              \
            {                                                                                                     \
                const uint32_t hb_ = dequant_hbits<WT>(rf[0][0], fgrp);                                           \
                _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++) cur[s_] = dequant_wpair<WT>(rf[0][0], hb_, s_);  \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            _Pragma("unroll") for (int f = 0; f < 2 * TN; f++) {                                                  \
                const int kk = f / TN, a = f % TN;                                                                \
                const int fn = f + 1 < 2 * TN ? f + 1 : f;                                                        \
                const uint32_t hb_ = dequant_hbits<WT>(rf[fn / TN][fn % TN], fgrp);                               \
                const h8 wf = __builtin_bit_cast(h8, (u32x4){cur[0], cur[1], cur[2], cur[3]});                    \
                _Pragma("unroll") for (int b = 0; b < TM; b++) {                                                  \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[kk][b], acc[a][b], 0, 0, 0);        \
                    if (f == 0 && b == 0) {           /* operands of k-slice 1, under the first MFMAs */          \
                        _Pragma("unroll") for (int a2 = 0; a2 < TN; a2++) RING_RAW(1, a2, rf[1][a2]);             \
                        _Pragma("unroll") for (int b2 = 0; b2 < TM; b2++) xf[1][b2] = *(const h8 *)(sb + LY::XOFF + 2 * lds_off(xrow0 + b2 * 16, 4 + fgrp)); \
                    }                                                                                             \
                    if (f + 1 < 2 * TN) {                                                                         \
                        _Pragma("unroll") for (int s_ = b * (4 / TM); s_ < (b + 1) * (4 / TM); s_++)              \
                            nxt[s_] = dequant_wpair<WT>(rf[fn / TN][fn % TN], hb_, s_);                           \
                    }                                                                                             \
                    __builtin_amdgcn_sched_barrier(0);                                                            \
                }                                                                                                 \
                _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++) cur[s_] = nxt[s_];                               \
            }                                                                                                     \
        }                                                                                                         \
    }

#define RING_WAIT(tiles_) ring_wait_vmcnt<(tiles_) * LY::COUNT>()

    // epilogue operands, requested before the K loop (older than every LDS-DMA request, so the counted waits below still mean what
    // they say: vmcnt retires in order).  Without split-K only: with it, the one workgroup that runs the epilogue is not known yet.
    const int nb_w = n0 + wn * (BN / WN), mb_w = m0 + wm * (RBM / WM);
    f4 bias_pre[TN], resid_pre[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++) {
        bias_pre[a] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < TM; b++) resid_pre[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
    }
    if (ksplit == 1) {
#pragma unroll
        for (int a = 0; a < TN; a++) {
            int n = nb_w + a * 16 + fgrp * 4;
            n = n < p.W.N ? n : 0;                     // (clamped: columns past N are never stored)
            if (EPI != EPI_PATCH_F32 && p.bias) bias_pre[a] = *(const f4 *)(p.bias + n);
            if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
                for (int b = 0; b < TM; b++) {
                    const int m = mb_w + b * 16 + frow;
                    resid_pre[a][b] = *(const f4 *)(p.resid + (size_t)(m < p.M ? m : p.M - 1) * p.ldc + n);
                }
            }
        }
    }
#ifdef CLIPAMD_G8_TIMING   // tuning builds: per-workgroup phase stamps (shader clock + 100 MHz real-time clock) into the split-K workspace
    unsigned long long * stamp = (unsigned long long *)p.sk_ws + (size_t)blockIdx.x * 8;
    const bool stamper = p.sk_ws && tid == 0 && ksplit == 1;
    if (stamper) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // prologue: tiles 0 .. NS-2 in flight (tile indices past the end are clamped: the request counts stay uniform, the extra copies
    // land in stages nobody reads any more)
#ifdef CLIPAMD_ABLATION
    if (!(p.debug & 4))
#endif
#pragma unroll
    for (int s = 0; s < NS - 1; s++) RING_ISSUE(s, (s < last ? s : last));
    // LayerNorm folded into this GEMM (gemm_common.h): thread t < 64 reduces the statistics of row m0 + t to (mean, rstd), and the c
    // vector is fetched, behind the prologue requests; hipcc's wait for these loads (before the loop) also lands the prologue tiles,
    // so the counted waits of the loop see only the loop's own requests
    constexpr bool LNE = EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16;
    const bool ln = LNE && p.ln_c != nullptr;
    f4 c_pre[TN];
    float2 ln_mine = make_float2(0.f, 1.f);
#pragma unroll
    for (int a = 0; a < TN; a++) c_pre[a] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (LNE) {
        if (ln) {
#pragma unroll
            for (int a = 0; a < TN; a++) {
                int n = nb_w + a * 16 + fgrp * 4;
                n = n < p.W.N ? n : 0;
                c_pre[a] = *(const f4 *)(p.ln_c + n);
            }
            if (tid < RBM) ln_mine = ln_row_centred(p, m0 + tid < p.M ? m0 + tid : p.M - 1, n0 == 0 && split == 0);
        }
    }
    int st = 0;                // stage of tile kt
    int sq = NS - 1;           // stage the next request goes to
    for (int kt = 0; kt < nk; kt++) {
        RING_WAIT(NS - 2);     // this wave's requests of tile kt have landed (the NS-2 newer tiles may still fly)
        ring_barrier();        // ... and everybody's; every wave is past its reads of tile kt-1 (consumed by MFMAs already issued)
#ifdef CLIPAMD_G8_TIMING
        if (stamper && kt == 0) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
#ifdef CLIPAMD_ABLATION   // kernel-tuning builds only (scripts/build_variant.sh): p.debug bit 0 no requests in the loop, bit 1 no fragment reads / MFMAs, bit 2 no requests in the prologue
        if (!(p.debug & 1)) {
            const int tq = kt + NS - 1;
            RING_ISSUE(sq, (tq < last ? tq : last));
        }
        if (!(p.debug & 2)) RING_COMPUTE(st);
#else
        {
            const int tq = kt + NS - 1;
            RING_ISSUE(sq, (tq < last ? tq : last));
        }
        RING_COMPUTE(st);
#endif
        st = st + 1 == NS ? 0 : st + 1;
        sq = sq + 1 == NS ? 0 : sq + 1;
    }
#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[2] = __builtin_amdgcn_s_memtime();     // (before the final wait: the "epilogue" span includes it)
#endif
    ring_wait_vmcnt<0>();      // nothing may land in this LDS allocation after the workgroup has left it
    if constexpr (LNE) {
        if (ln) ln_rows_publish((float2 *)smem, ln_mine, tid, RBM, [] { __syncthreads(); });
    }
    const float2 * rs_lane = (const float2 *)smem + wm * (RBM / WM) + frow;
#undef RING_ISSUE
#undef RING_COMPUTE
#undef RING_RAW
#undef RING_WAIT
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

    if (ksplit > 1) {
        // deterministic split-K fix-up, as in k_gemm.hip: partials parked with agent-scope (write-through) stores, per-tile ticket,
        // the last arriver sums all partials in split order and runs the epilogue
        __shared__ int is_last;
        typedef unsigned long long u64;
        u64 * part = (u64 *)(p.sk_ws + ((size_t)tile_id * ksplit) * (RBM * BN));
        constexpr int PER_SPLIT = RBM * BN / 2;
#pragma unroll
        for (int a = 0; a < TN; a++)
#pragma unroll
            for (int b = 0; b < TM; b++) {
                const float2 lo = make_float2(acc[a][b][0], acc[a][b][1]), hi2 = make_float2(acc[a][b][2], acc[a][b][3]);
                u64 * dst = part + (size_t)split * PER_SPLIT + ((a * TM + b) * 2) * NTR + tid;
                __hip_atomic_store(dst, __builtin_bit_cast(u64, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + NTR, __builtin_bit_cast(u64, hi2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(p.sk_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            is_last = (ticket == (unsigned)ksplit - 1u);
            if (is_last) __hip_atomic_store(p.sk_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    const u64 * src = part + (size_t)sp * PER_SPLIT + ((a * TM + b) * 2) * NTR + tid;
                    const float2 lo = __builtin_bit_cast(float2, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const float2 hi2 = __builtin_bit_cast(float2, __hip_atomic_load(src + NTR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    acc[a][b] += (f4){lo.x, lo.y, hi2.x, hi2.y};
                }
        }
    }
    if (ksplit == 1) gemm_epilogue_pre<EPI, TN, TM>(p, acc, bias_pre, resid_pre, nb_w, mb_w, frow, fgrp, ln, c_pre, rs_lane);
    else gemm_epilogue<EPI, TN, TM>(p, acc, nb_w, mb_w, frow, fgrp, ln, rs_lane);
#ifdef CLIPAMD_G8_TIMING
    if (stamper) {
        stamp[3] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[5] = __builtin_amdgcn_s_memtime();
        stamp[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int WT, int BN, int NW, int WN, int EPI>
void launch_ring_t(const GemmParams & p, hipStream_t stream) {
    using LY = RingLayout<WT, BN, NW>;
    const int tiles_m = (p.M + RBM - 1) / RBM, tiles_n = (p.W.N + BN - 1) / BN;
    constexpr size_t smem = (size_t)LY::NS * LY::STAGE;
    static unsigned long long lds_ok = 0;
    if (smem > 64 * 1024) opt_in_dynamic_lds(gemm_ring_kernel<WT, BN, NW, WN, EPI>, smem, lds_ok);
    hipLaunchKernelGGL((gemm_ring_kernel<WT, BN, NW, WN, EPI>), dim3(tiles_m * tiles_n * p.ksplit), dim3(NW * 64), smem, stream, p);
}

// bn: 64 (waves 2 x 2) or 128 (4 x 1).  Eight-wave workgroups and 256 weight rows per tile were built and measured too
// (profiles/r02_second_session_experiments.txt section 3): never the fastest form of any shape, not instantiated.
template <int WT, int EPI>
void launch_ring_bn(const GemmParams & p, int bn, hipStream_t stream) {
    if (bn >= 128) launch_ring_t<WT, 128, 4, 4, EPI>(p, stream);
    else launch_ring_t<WT, 64, 4, 2, EPI>(p, stream);
}

template <int WT>
void launch_ring_epi(const GemmParams & p, int epi, int bn, hipStream_t stream) {
    switch (epi) {
    case EPI_F32: launch_ring_bn<WT, EPI_F32>(p, bn, stream); break;
    case EPI_F16: launch_ring_bn<WT, EPI_F16>(p, bn, stream); break;
    case EPI_GELU_F16: launch_ring_bn<WT, EPI_GELU_F16>(p, bn, stream); break;
    case EPI_QGELU_F16: launch_ring_bn<WT, EPI_QGELU_F16>(p, bn, stream); break;
    case EPI_RESID_F32: launch_ring_bn<WT, EPI_RESID_F32>(p, bn, stream); break;
    case EPI_PATCH_F32: launch_ring_bn<WT, EPI_PATCH_F32>(p, bn, stream); break;
    }
}

}  // namespace

// One translation unit per weight type (-DCLIPAMD_RING_WT=<n>), the dispatcher compiled once with no define (as k_gemm.hip).
#ifdef CLIPAMD_RING_WT
#define CLIPAMD_RCAT2(a, b) a##b
#define CLIPAMD_RCAT(a, b) CLIPAMD_RCAT2(a, b)
void CLIPAMD_RCAT(launch_gemm_ring_wt, CLIPAMD_RING_WT)(const GemmParams & p, int epilogue, int bn, hipStream_t stream) {
    launch_ring_epi<CLIPAMD_RING_WT>(p, epilogue, bn, stream);
}
#else
void launch_gemm_ring_wt0(const GemmParams &, int, int, hipStream_t);
void launch_gemm_ring_wt1(const GemmParams &, int, int, hipStream_t);
void launch_gemm_ring_wt2(const GemmParams &, int, int, hipStream_t);
void launch_gemm_ring_wt3(const GemmParams &, int, int, hipStream_t);
void launch_gemm_ring_wt4(const GemmParams &, int, int, hipStream_t);
void launch_gemm_ring_wt5(const GemmParams &, int, int, hipStream_t);

// p.ksplit already validated by launch_gemm (workspace, ticket counters); bn in {64, 128, 256}
void launch_gemm_ring(const GemmParams & p, int epilogue, int bn, hipStream_t stream) {
    switch (p.W.wtype) {
    case W_F16: launch_gemm_ring_wt0(p, epilogue, bn, stream); break;
    case W_Q4_0: launch_gemm_ring_wt1(p, epilogue, bn, stream); break;
    case W_Q4_1: launch_gemm_ring_wt2(p, epilogue, bn, stream); break;
    case W_Q5_0: launch_gemm_ring_wt3(p, epilogue, bn, stream); break;
    case W_Q5_1: launch_gemm_ring_wt4(p, epilogue, bn, stream); break;
    case W_Q8_0: launch_gemm_ring_wt5(p, epilogue, bn, stream); break;
    }
}
#endif

}  // namespace clipamd
