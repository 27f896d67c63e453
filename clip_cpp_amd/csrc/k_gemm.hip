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
//   * block-quantised weights stay quantised in HBM; each workgroup streams the [BN rows x 2 blocks]
//     slab of its tile with one coalesced 16 B load per lane (block-column-major planes, kernels.h),
//     dequantises in registers with packed-fp16 VALU (magic-number 0x6400|q trick, v_pk_add/v_pk_mul)
//     and stages the fp16 tile in LDS;
//   * activations are fp16 (the producing kernel's epilogue rounded them once), staged through LDS;
//   * both LDS tiles are [rows][64] fp16 with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) so the
//     ds_read_b128 fragment reads of v_mfma_f32_16x16x32_f16 are bank-conflict free;
//   * 256 threads = 4 waves in a 2x2 grid; each wave owns a (BN/2)x(BM/2) sub-tile as 16x16 MFMA
//     fragments, fp32 accumulation; the weight is the MFMA "A" operand so that each lane ends up with
//     4 consecutive output columns of one row -> 8/16-byte epilogue stores;
//   * register-prefetch double buffering: tile k+1 is loaded (quantised) into VGPRs before the MFMAs of
//     tile k are issued, dequantised + written to the other LDS buffer afterwards, one barrier per step;
//   * workgroup -> tile mapping is XCD-aware (blocks that share a weight slab land on the same L2).
//
// Roofline: compute (MFMA fp16, 2.5 PFLOP/s dense) for M >= ~256; HBM (weight bytes) for M <= 64.

#include "kernels.h"

namespace clipamd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 64;          // K step per iteration (two 32-wide quant blocks)
constexpr int NTHREADS = 256;

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

// idx = kb * Npad + n (block index in the planes), g = lane >> 4
template <int WT>
__device__ __forceinline__ void load_wfrag(WFrag<WT> & f, const DevWeight & W, size_t idx, int g) {
    if constexpr (WT == W_Q8_0) {
        const uint2 v = *(const uint2 *)((const uint8_t *)W.qs + idx * 32 + g * 8);
        f.q = v.x;
        f.q1 = v.y;
    } else {
        f.q = *(const uint32_t *)((const uint8_t *)W.qs + idx * 16 + g * 4);
    }
    if constexpr (WT == W_Q5_0 || WT == W_Q5_1) f.h = ((const uint32_t *)W.qh)[idx];
    if constexpr (WT == W_Q4_1 || WT == W_Q5_1) f.dm = ((const h2 *)W.dm)[idx];
    else f.d = ((const half_t *)W.dm)[idx];
}

template <int WT>
__device__ __forceinline__ h8 dequant_wfrag(const WFrag<WT> & f, int g) {
    uint32_t o[4] = {0, 0, 0, 0};
    if constexpr (WT == W_F16) return __builtin_bit_cast(h8, (u32x4){f.q, 0u, 0u, 0u});
    else
    if constexpr (WT == W_Q8_0) {
        const h2 scale = (h2){f.d, f.d};
        const h2 sub = splat(1152.0f);
        o[0] = h2u((u2h((f.q & 0x00FF00FFu) | 0x64006400u) - sub) * scale);
        o[1] = h2u((u2h(((f.q >> 8) & 0x00FF00FFu) | 0x64006400u) - sub) * scale);
        o[2] = h2u((u2h((f.q1 & 0x00FF00FFu) | 0x64006400u) - sub) * scale);
        o[3] = h2u((u2h(((f.q1 >> 8) & 0x00FF00FFu) | 0x64006400u) - sub) * scale);
    } else {
        h2 scale, sub, add;
        if constexpr (WT == W_Q4_0) { scale = (h2){f.d, f.d}; sub = splat(1032.0f); }
        if constexpr (WT == W_Q5_0) { scale = (h2){f.d, f.d}; sub = splat(1040.0f); }
        if constexpr (WT == W_Q4_1 || WT == W_Q5_1) {
            scale = (h2){f.dm[0], f.dm[0]};
            add = (h2){f.dm[1], f.dm[1]};
            sub = splat(1024.0f);
        }
        uint32_t hb = 0;
        if constexpr (WT == W_Q5_0 || WT == W_Q5_1) hb = (uint32_t)(((uint64_t)f.h << 4) >> (4 * g));  // pair bits of word g at 4+s / 20+s
#pragma unroll
        for (int s = 0; s < 4; s++) {
            uint32_t u = ((f.q >> (4 * s)) & 0x000F000Fu) | 0x64006400u;
            if constexpr (WT == W_Q5_0 || WT == W_Q5_1) u |= (hb >> s) & 0x00100010u;
            h2 v = u2h(u) - sub;  // exact small integer
            if constexpr (WT == W_Q4_1 || WT == W_Q5_1) v = __builtin_elementwise_fma(v, scale, add);  // q*d + m, one rounding
            else v = v * scale;
            o[s] = h2u(v);
        }
    }
    return __builtin_bit_cast(h8, (u32x4){o[0], o[1], o[2], o[3]});
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

__device__ __forceinline__ float gelu_tanh(float x) {
    // ggml_gelu_f32: 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)))
    const float u = 0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x);
    // tanh(u) = 1 - 2/(exp(2u)+1)
    const float e = __expf(2.0f * u);
    const float th = 1.0f - 2.0f / (e + 1.0f);
    return 0.5f * x * (1.0f + th);
}
__device__ __forceinline__ float gelu_quick(float x) { return x / (1.0f + __expf(-1.702f * x)); }

// ---- epilogue.  D[i][j]: i = weight row n (row = 4*(lane>>4)+reg), j = activation row m (col = lane&15).
// nbase / mbase: first weight row / activation row of this wave's sub-tile.
template <int EPI, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue(const GemmParams & p, f4 (&acc)[TN][TM], int nbase, int mbase, int frow, int fgrp) {
    const int N = p.W.N;
#pragma unroll
    for (int a = 0; a < TN; a++) {
        const int n = nbase + a * 16 + fgrp * 4;
        if (n >= N) continue;
        f4 bias = (f4){0.f, 0.f, 0.f, 0.f};
        if (EPI != EPI_PATCH_F32 && p.bias) bias = *(const f4 *)(p.bias + n);
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = mbase + b * 16 + frow;
            if (m >= p.M) continue;
            f4 v = acc[a][b] + bias;
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

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant: 512 threads = 4 COMPUTE waves (ds_read + MFMA only, 2x2 over the BMxBN tile) and
// 4 LOADER waves (global loads two K-steps ahead, dequantisation, ds_write), one of each kind per SIMD, so the
// matrix pipe keeps running while the next tile is fetched / dequantised / staged.  Two LDS buffers, one
// workgroup barrier per K-step: after it the compute waves read the buffer the loaders just filled and the
// loaders overwrite the one the compute waves just finished with.
// ---------------------------------------------------------------------------------------------
template <int WT, int BM, int BN, int EPI>
__global__ void __launch_bounds__(512, 2) gemm_ws_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t * Xs = (half_t *)smem_raw;                 // [2][BM*BK]
    half_t * Ws = Xs + 2 * BM * BK;                   // [2][BN*BK]
    constexpr int TN = BN / 32, TM = BM / 32;
    constexpr int XCH = BM * 8 / NTHREADS, WCH = BN * 8 / NTHREADS;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.W.Kpad / BK;
    const int last = nk - 1;

    if (wave >= 4) {
        // ================================ LOADER waves ================================
        const int lt = tid - 256;
        const int xrow0 = lt >> 3, xc = lt & 7;
        const int xlds0 = lds_off(xrow0, xc);
        size_t xgoff[XCH];
#pragma unroll
        for (int i = 0; i < XCH; i++) {
            int gm = m0 + xrow0 + 32 * i;
            gm = gm < p.M ? gm : p.M - 1;
            xgoff[i] = (size_t)gm * p.lda + xc * 8;
        }
        const int bnl = lt % BN, bkb = lt / BN;
        const bool bact = (BN * 2 >= NTHREADS) || lt < BN * 2;
        u32x4 X0[XCH], X1[XCH];
        u32x4 W0[WT == W_F16 ? WCH : 1], W1[WT == W_F16 ? WCH : 1];
        RawBlock<WT> B0, B1;
#define WS_LOAD(XR, WR, BR, kt_)                                                               \
        {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < XCH; i++) XR[i] = *(const u32x4 *)(p.A + xgoff[i] + (kt_) * BK); \
            if constexpr (WT == W_F16) {                                                       \
                _Pragma("unroll") for (int i = 0; i < WCH; i++)                                \
                    WR[i] = *(const u32x4 *)((const half_t *)p.W.w16 + (size_t)(n0 + xrow0 + 32 * i) * p.W.Kpad + (kt_) * BK + xc * 8); \
            } else {                                                                           \
                if (bact) load_block<WT>(BR, p.W, (size_t)((kt_) * 2 + bkb) * p.W.Npad + n0 + bnl); \
            }                                                                                  \
        }
#define WS_STORE(XR, WR, BR, buf_)                                                             \
        {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < XCH; i++) *(u32x4 *)(Xs + (buf_) * BM * BK + xlds0 + i * 32 * BK) = XR[i]; \
            if constexpr (WT == W_F16) {                                                       \
                _Pragma("unroll") for (int i = 0; i < WCH; i++) *(u32x4 *)(Ws + (buf_) * BN * BK + xlds0 + i * 32 * BK) = WR[i]; \
            } else {                                                                           \
                if (bact) {                                                                    \
                    half_t * wrow_ = Ws + (buf_) * BN * BK + bnl * BK;                         \
                    _Pragma("unroll") for (int j = 0; j < 4; j++)                              \
                        *(h8 *)(wrow_ + (((bkb * 4 + j) ^ (bnl & 7)) << 3)) = dequant_wfrag<WT>(block_word<WT>(BR, j), j); \
                }                                                                              \
            }                                                                                  \
        }
        WS_LOAD(X0, W0, B0, 0);
        { const int t1 = last < 1 ? last : 1; WS_LOAD(X1, W1, B1, t1); }
        WS_STORE(X0, W0, B0, 0);
        __syncthreads();                                   // tile 0 visible
        for (int kt = 0; kt < nk; kt += 2) {
            { const int t2 = kt + 2 < last ? kt + 2 : last; WS_LOAD(X0, W0, B0, t2); }
            WS_STORE(X1, W1, B1, 1);                       // tile kt+1 -> buf1 (compute is on buf0)
            __syncthreads();
            if (kt + 1 >= nk) break;
            { const int t3 = kt + 3 < last ? kt + 3 : last; WS_LOAD(X1, W1, B1, t3); }
            WS_STORE(X0, W0, B0, 0);                       // tile kt+2 -> buf0 (compute is on buf1)
            __syncthreads();
        }
#undef WS_LOAD
#undef WS_STORE
    } else {
    // ================================ COMPUTE waves ================================
    const int lane = tid & 63;
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, fgrp = lane >> 4;
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
#define WS_COMPUTE(buf_)                                                                       \
    {                                                                                          \
        const half_t * xs = Xs + (buf_) * BM * BK;                                             \
        const half_t * ws = Ws + (buf_) * BN * BK;                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                     \
            h8 xf[TM];                                                                         \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                     \
                xf[b] = *(const h8 *)(xs + lds_off(wm * (BM / 2) + b * 16 + frow, kk * 4 + fgrp)); \
            _Pragma("unroll") for (int a = 0; a < TN; a++) {                                   \
                const h8 wf = *(const h8 *)(ws + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fgrp)); \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[a][b], 0, 0, 0); \
            }                                                                                  \
        }                                                                                      \
    }
    __syncthreads();                                       // tile 0 visible
    for (int kt = 0; kt < nk; kt += 2) {
        WS_COMPUTE(0);
        __syncthreads();
        if (kt + 1 >= nk) break;
        WS_COMPUTE(1);
        __syncthreads();
    }
#undef WS_COMPUTE
    asm volatile("" ::: "memory");   // keep the epilogue's bias / residual loads below the main loop (register pressure)
    gemm_epilogue<EPI, TN, TM>(p, acc, n0 + wn * (BN / 2), m0 + wm * (BM / 2), frow, fgrp);
    }
}

// ---------------------------------------------------------------------------------------------
template <int WT, int BM, int BN, int EPI, bool DIRECT>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_kernel(const GemmParams p) {
    // WLDS: the weight tile is staged (dequantised) through LDS like X.  !WLDS (DIRECT, quantised types only): every wave
    // loads its own MFMA A-fragments as packed quants straight into registers and dequantises beside the MFMAs.
    constexpr bool WLDS = (WT == W_F16) || !DIRECT;
    constexpr bool WQLDS = WLDS && (WT != W_F16);     // quantised weights through LDS: one 32-weight block per thread per K-step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t * Xs = (half_t *)smem_raw;                 // [2][BM*BK]
    half_t * Ws = Xs + 2 * BM * BK;                   // [2][BN*BK]   (f16 weights only)

    constexpr int TN = BN / 32;   // 16-row MFMA fragments per wave along N (wave owns BN/2 rows)
    constexpr int TM = BM / 32;
    constexpr int XCH = BM * 8 / NTHREADS;            // 16 B chunks of the X tile per thread
    constexpr int WCH = BN * 8 / NTHREADS;            // (f16 weights) chunks per thread

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, fgrp = lane >> 4;

    // ---- XCD-aware tile mapping (bijective for any grid size) ----
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // n fastest: the tiles_n workgroups that share one X row-panel run back to back on one XCD, so the panel is
    // fetched into that L2 once; the (small) weight matrix stays L2-resident anyway.
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.W.Kpad / BK;

    // ---- per-thread global source coordinates (chunk q = tid + i*256 -> row (tid>>3) + 32 i, chunk tid&7) ----
    const int xrow0 = tid >> 3, xc = tid & 7;
    const int xlds0 = lds_off(xrow0, xc);     // rows advance by 32 per i: (row & 7) is unchanged -> offset += 32*BK
    size_t xgoff[XCH];
#pragma unroll
    for (int i = 0; i < XCH; i++) {
        int gm = m0 + xrow0 + 32 * i;
        gm = gm < p.M ? gm : p.M - 1;          // clamp: rows past M are computed but never stored
        xgoff[i] = (size_t)gm * p.lda + xc * 8;
    }
    const int wrow = n0 + wn * (BN / 2) + frow;   // this lane's weight row of fragment 0 (quantised path)

    u32x4 X0[XCH], X1[XCH];                   // two register stages of the X tile
    u32x4 W0[WT == W_F16 ? WCH : 1], W1[WT == W_F16 ? WCH : 1];
    WFrag<WT> F0[WLDS ? 1 : TN * 2], F1[WLDS ? 1 : TN * 2];   // DIRECT: two stages of raw weight fragments [a][kk]
    RawBlock<WT> B0, B1;                                      // WQLDS: two stages of one raw block per thread
    const int bnl = tid % BN, bkb = tid / BN;                 // block (row bnl, k-block bkb) of the tile handled by this thread
    const bool bact = (BN * 2 >= NTHREADS) || tid < BN * 2;

#define LOAD_X(R, kt_)                                                                         \
    _Pragma("unroll") for (int i = 0; i < XCH; i++) R[i] = *(const u32x4 *)(p.A + xgoff[i] + (kt_) * BK);
#define STORE_X(R, buf_)                                                                       \
    _Pragma("unroll") for (int i = 0; i < XCH; i++) *(u32x4 *)(Xs + (buf_) * BM * BK + xlds0 + i * 32 * BK) = R[i];
#define LOAD_W16(R, B, kt_)                                                                    \
    if constexpr (WT == W_F16) {                                                               \
        _Pragma("unroll") for (int i = 0; i < WCH; i++)                                        \
            R[i] = *(const u32x4 *)((const half_t *)p.W.w16 + (size_t)(n0 + xrow0 + 32 * i) * p.W.Kpad + (kt_) * BK + xc * 8); \
    } else {                                                                                   \
        if (bact) load_block<WT>(B, p.W, (size_t)((kt_) * 2 + bkb) * p.W.Npad + n0 + bnl);     \
    }
#define STORE_W16(R, B, buf_)                                                                  \
    if constexpr (WT == W_F16) {                                                               \
        _Pragma("unroll") for (int i = 0; i < WCH; i++) *(u32x4 *)(Ws + (buf_) * BN * BK + xlds0 + i * 32 * BK) = R[i]; \
    } else {                                                                                   \
        if (bact) {                                                                            \
            half_t * wrow_ = Ws + (buf_) * BN * BK + bnl * BK;                                 \
            _Pragma("unroll") for (int j = 0; j < 4; j++)                                      \
                *(h8 *)(wrow_ + (((bkb * 4 + j) ^ (bnl & 7)) << 3)) = dequant_wfrag<WT>(block_word<WT>(B, j), j); \
        }                                                                                      \
    }
#define LOAD_WQ(F, kt_)                                                                        \
    _Pragma("unroll") for (int a = 0; a < TN; a++)                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++)                                       \
            load_wfrag<WT>(F[a * 2 + kk], p.W, (size_t)((kt_) * 2 + kk) * p.W.Npad + wrow + a * 16, fgrp);

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

#define COMPUTE(buf_, F)                                                                       \
    {                                                                                          \
        const half_t * xs = Xs + (buf_) * BM * BK;                                             \
        const half_t * ws = Ws + (buf_) * BN * BK;                                             \
        (void)ws;                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                     \
            h8 xf[TM];                                                                         \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                     \
                xf[b] = *(const h8 *)(xs + lds_off(wm * (BM / 2) + b * 16 + frow, kk * 4 + fgrp)); \
            _Pragma("unroll") for (int a = 0; a < TN; a++) {                                   \
                h8 wf;                                                                         \
                if constexpr (WLDS) wf = *(const h8 *)(ws + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fgrp)); \
                else wf = dequant_wfrag<WT>(F[a * 2 + kk], fgrp);                              \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[a][b], 0, 0, 0); \
            }                                                                                  \
        }                                                                                      \
    }

    // ---- main loop.  X (and f16 W): two register stages (tiles k+1, k+2 in flight) + two LDS buffers.
    //      Quantised W: fragments of tile k+1 are loaded into registers while tile k is multiplied. One barrier per K-step.
    // Every prefetch is UNCONDITIONAL (tile index clamped to the last tile): a load inside an `if` makes hipcc's
    // s_waitcnt insertion assume it may not have been issued and fall back to vmcnt(0/1) for the older tile, which
    // exposes the full memory latency of the tile just requested on every K-step.
    const int last = nk - 1;
    LOAD_X(X0, 0);
    if constexpr (WLDS) { LOAD_W16(W0, B0, 0); } else { LOAD_WQ(F0, 0); }
    {
        const int t1 = last < 1 ? last : 1;
        LOAD_X(X1, t1);
        if constexpr (WLDS) { LOAD_W16(W1, B1, t1); }
    }
    STORE_X(X0, 0);
    if constexpr (WLDS) { STORE_W16(W0, B0, 0); }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        {
            const int t2 = kt + 2 < last ? kt + 2 : last;
            LOAD_X(X0, t2);
            if constexpr (WLDS) { LOAD_W16(W0, B0, t2); }
            if constexpr (!WLDS) { LOAD_WQ(F1, kt + 1); }
        }
        COMPUTE(0, F0);
        STORE_X(X1, 1);
        if constexpr (WLDS) { STORE_W16(W1, B1, 1); }
        __syncthreads();
        {
            const int t3 = kt + 3 < last ? kt + 3 : last;
            LOAD_X(X1, t3);
            if constexpr (WLDS) { LOAD_W16(W1, B1, t3); }
            if constexpr (!WLDS) { const int t2 = kt + 2 < last ? kt + 2 : last; LOAD_WQ(F0, t2); }
        }
        COMPUTE(1, F1);
        STORE_X(X0, 0);
        if constexpr (WLDS) { STORE_W16(W0, B0, 0); }
        __syncthreads();
    }
    if (kt < nk) COMPUTE(0, F0);   // odd tail (its X tile was stored by the last iteration / the prologue)
#undef LOAD_X
#undef STORE_X
#undef LOAD_W16
#undef STORE_W16
#undef LOAD_WQ
#undef COMPUTE

    gemm_epilogue<EPI, TN, TM>(p, acc, n0 + wn * (BN / 2), m0 + wm * (BM / 2), frow, fgrp);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant: fp16 tiles (X always, W when the weights are f16) go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write): one wave-instruction moves a 1 KB piece = 8 tile rows;
// the LDS image is lane-linear, so the 16-byte-chunk XOR swizzle is applied on the per-lane SOURCE address
// (lane l writes row l>>3, position l&7, and therefore fetches chunk (l&7)^(l>>3)); fragment reads use lds_off().
// Quantised weights: one raw 32-weight block per thread in registers (two stages), dequantised into LDS.
// Per K-step: issue DMA for tile k+1 + register loads for tile k+2, multiply tile k, dequant-store tile k+1, barrier
// (the barrier's vmcnt(0) is what lands the DMA, so nothing inside a step waits on memory).
// ---------------------------------------------------------------------------------------------
template <int WT, int BM, int BN, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_dma_kernel(const GemmParams p) {
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
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.W.Kpad / BK;
    const int last = nk - 1;

    // DMA source addresses: piece = wave*XPW + i covers tile rows piece*8 .. +7
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    const half_t * xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; i++) {
        int gm = m0 + (wave * XPW + i) * 8 + prow;
        gm = gm < p.M ? gm : p.M - 1;
        xsrc[i] = p.A + (size_t)gm * p.lda + pchunk * 8;
    }
    const half_t * wsrc[WT == W_F16 ? WPW : 1];
    if constexpr (WT == W_F16) {
#pragma unroll
        for (int i = 0; i < WPW; i++)
            wsrc[i] = (const half_t *)p.W.w16 + (size_t)(n0 + (wave * WPW + i) * 8 + prow) * p.W.Kpad + pchunk * 8;
    }
    RawBlock<WT> B0, B1;
    const int bnl = tid % BN, bkb = tid / BN;
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
        if (bact) load_block<WT>(B, p.W, (size_t)((kt_) * 2 + bkb) * p.W.Npad + n0 + bnl);     \
    }
#define STORE_B(B, buf_)                                                                       \
    if constexpr (WT != W_F16) {                                                               \
        if (bact) {                                                                            \
            half_t * wrow_ = Ws + (buf_) * BN * BK + bnl * BK;                                 \
            _Pragma("unroll") for (int j = 0; j < 4; j++)                                      \
                *(h8 *)(wrow_ + (((bkb * 4 + j) ^ (bnl & 7)) << 3)) = dequant_wfrag<WT>(block_word<WT>(B, j), j); \
        }                                                                                      \
    }
    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};
#define COMPUTE(buf_)                                                                          \
    {                                                                                          \
        const half_t * xs = Xs + (buf_) * BM * BK;                                             \
        const half_t * ws = Ws + (buf_) * BN * BK;                                             \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                     \
            h8 xf[TM];                                                                         \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                     \
                xf[b] = *(const h8 *)(xs + lds_off(wm * (BM / 2) + b * 16 + frow, kk * 4 + fgrp)); \
            _Pragma("unroll") for (int a = 0; a < TN; a++) {                                   \
                const h8 wf = *(const h8 *)(ws + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fgrp)); \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[a][b], 0, 0, 0); \
            }                                                                                  \
        }                                                                                      \
    }

    DMA_TILE(0, 0);
    LOAD_B(B0, 0);
    { const int t1 = last < 1 ? last : 1; LOAD_B(B1, t1); }
    STORE_B(B0, 0);
    __syncthreads();
    int kt = 0;
    if (p.debug == 0) {
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
    } else {   // ablation copy of the loop (kernel tuning only; results are garbage)
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
    }
    if (kt < nk) COMPUTE(0);
#undef DMA_TILE
#undef LOAD_B
#undef STORE_B
#undef COMPUTE
    asm volatile("" ::: "memory");
    gemm_epilogue<EPI, TN, TM>(p, acc, n0 + wn * (BN / 2), m0 + wm * (BM / 2), frow, fgrp);
}

template <int WT, int BM, int BN, int EPI>
void launch_dma(const GemmParams & p, hipStream_t stream) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + BN - 1) / BN;
    const size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(half_t);
    hipLaunchKernelGGL((gemm_dma_kernel<WT, BM, BN, EPI>), dim3(tiles_m * tiles_n), dim3(NTHREADS), smem, stream, p);
}

// ---------------------------------------------------------------------------------------------
// Ring variant (3-slot LDS ring, everything staged by LDS-DMA, two tiles in flight, ONE raw barrier per K-step):
//   * X tile (and the W tile of f16 weights): global_load_lds_dwordx4, swizzled via the source address as above;
//   * quantised W tile: the RAW packed blocks are DMA'd (16 B quants + 2/4 B scale (+4 B fifth bits) per block) —
//     4.5-8.5 bits per weight also inside LDS — and dequantised in registers when a wave builds its MFMA
//     A-fragment (ds_read_b32 of word g + ds_read_u16 of the scale -> 8 fp16).  No dequantised copy is ever written.
//   * no VGPR-destination global load exists in the loop, so the waits are hand-counted: after multiplying tile k
//     a wave waits `vmcnt(NI)` (= its own NI DMA instructions of tile k+2 may stay in flight, those of tile k+1
//     have landed), then s_barrier publishes every wave's part of tile k+1 and retires slot k%3 for reuse.
// BN is fixed at 128: each of the 4 waves DMAs exactly one 64-block piece of quants/scales per tile.
// ---------------------------------------------------------------------------------------------
template <int WT> struct RingFmt { static constexpr int QB = 16, DB = 2, HB = 0; };
template <> struct RingFmt<W_Q4_1> { static constexpr int QB = 16, DB = 4, HB = 0; };
template <> struct RingFmt<W_Q5_0> { static constexpr int QB = 16, DB = 2, HB = 4; };
template <> struct RingFmt<W_Q5_1> { static constexpr int QB = 16, DB = 4, HB = 4; };
template <> struct RingFmt<W_Q8_0> { static constexpr int QB = 32, DB = 2, HB = 0; };

#define CLIPAMD_GLDS(gptr_, lptr_, size_) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr_), (__attribute__((address_space(3))) void *)(lptr_), size_, 0, 0)

template <int WT, int BM, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2) gemm_ring_kernel(const GemmParams p) {
    constexpr int BN = 128;
    constexpr bool WF16 = (WT == W_F16);
    using RF = RingFmt<WT>;
    constexpr int NSLOT = 3;
    constexpr int XS_BYTES = BM * BK * 2;                                   // one X slot
    constexpr int WS_BYTES = WF16 ? BN * BK * 2 : 2 * BN * (RF::QB + RF::DB + RF::HB);   // one W slot
    constexpr int Q_OFF = 0, D_OFF = 2 * BN * RF::QB, H_OFF = D_OFF + 2 * BN * RF::DB;  // planes inside a raw W slot
    constexpr int TN = BN / 32, TM = BM / 32;
    constexpr int XPW = BM / 32;                                            // X pieces (1 KB) per wave
    constexpr int WPW = BN / 32;                                            // f16 W pieces per wave
    // DMA instructions one wave issues per tile (hand-counted for s_waitcnt vmcnt)
    constexpr int NI = XPW + (WF16 ? WPW : (RF::QB / 16) + 1 + (RF::HB ? 1 : 0));

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char * Xring = smem_raw;
    unsigned char * Wring = smem_raw + NSLOT * XS_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, fgrp = lane >> 4;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.W.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.W.Kpad / BK;
    const int last = nk - 1;

    // ---- DMA sources ----
    const int prow = lane >> 3;
    const int pchunk = (lane & 7) ^ prow;
    const half_t * xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; i++) {
        int gm = m0 + (wave * XPW + i) * 8 + prow;
        gm = gm < p.M ? gm : p.M - 1;
        xsrc[i] = p.A + (size_t)gm * p.lda + pchunk * 8;
    }
    const half_t * wsrc[WF16 ? WPW : 1];
    if constexpr (WF16) {
#pragma unroll
        for (int i = 0; i < WPW; i++)
            wsrc[i] = (const half_t *)p.W.w16 + (size_t)(n0 + (wave * WPW + i) * 8 + prow) * p.W.Kpad + pchunk * 8;
    }
    // quantised: wave w owns the 64-block piece (k-block w>>1, rows (w&1)*64 .. +63); lane l its block
    const int pkb = wave >> 1;
    const size_t pblk = (size_t)n0 + (wave & 1) * 64 + lane;     // + (kt*2+pkb)*Npad at issue time
    const int praw = (pkb * BN + (wave & 1) * 64) ;              // first block index of the piece inside a slot plane

#define RING_DMA(slot_, kt_)                                                                   \
    {                                                                                          \
        unsigned char * xs_ = Xring + (slot_) * XS_BYTES;                                      \
        unsigned char * ws_ = Wring + (slot_) * WS_BYTES;                                      \
        _Pragma("unroll") for (int i = 0; i < XPW; i++)                                        \
            CLIPAMD_GLDS(xsrc[i] + (kt_) * BK, xs_ + (wave * XPW + i) * 1024, 16);             \
        if constexpr (WF16) {                                                                  \
            _Pragma("unroll") for (int i = 0; i < WPW; i++)                                    \
                CLIPAMD_GLDS(wsrc[i] + (kt_) * BK, ws_ + (wave * WPW + i) * 1024, 16);         \
        } else {                                                                               \
            const size_t bi_ = (size_t)((kt_) * 2 + pkb) * p.W.Npad + pblk;                    \
            if constexpr (RF::QB == 32) {                                                      \
                CLIPAMD_GLDS((const uint8_t *)p.W.qs + bi_ * 32, ws_ + Q_OFF + praw * 16, 16);            \
                CLIPAMD_GLDS((const uint8_t *)p.W.qs + bi_ * 32 + 16, ws_ + Q_OFF + 2 * BN * 16 + praw * 16, 16); \
            } else {                                                                           \
                CLIPAMD_GLDS((const uint8_t *)p.W.qs + bi_ * 16, ws_ + Q_OFF + praw * 16, 16); \
            }                                                                                  \
            if constexpr (RF::DB == 2) { CLIPAMD_GLDS((const uint8_t *)p.W.dm + bi_ * 2, ws_ + D_OFF + praw * 2, 2); } \
            else { CLIPAMD_GLDS((const uint8_t *)p.W.dm + bi_ * 4, ws_ + D_OFF + praw * 4, 4); } \
            if constexpr (RF::HB != 0) { CLIPAMD_GLDS((const uint8_t *)p.W.qh + bi_ * 4, ws_ + H_OFF + praw * 4, 4); } \
        }                                                                                      \
    }

    f4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

#define RING_COMPUTE(slot_)                                                                    \
    {                                                                                          \
        const half_t * xs = (const half_t *)(Xring + (slot_) * XS_BYTES);                      \
        const unsigned char * ws = Wring + (slot_) * WS_BYTES;                                 \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++) {                                     \
            h8 xf[TM];                                                                         \
            _Pragma("unroll") for (int b = 0; b < TM; b++)                                     \
                xf[b] = *(const h8 *)(xs + lds_off(wm * (BM / 2) + b * 16 + frow, kk * 4 + fgrp)); \
            _Pragma("unroll") for (int a = 0; a < TN; a++) {                                   \
                h8 wf;                                                                         \
                if constexpr (WF16) {                                                          \
                    wf = *(const h8 *)((const half_t *)ws + lds_off(wn * (BN / 2) + a * 16 + frow, kk * 4 + fgrp)); \
                } else {                                                                       \
                    const int blk_ = kk * BN + wn * (BN / 2) + a * 16 + frow;                  \
                    WFrag<WT> f_;                                                              \
                    if constexpr (RF::QB == 32) {                                              \
                        const uint2 q2_ = *(const uint2 *)(ws + Q_OFF + (fgrp >> 1) * (2 * BN * 16) + blk_ * 16 + (fgrp & 1) * 8); \
                        f_.q = q2_.x; f_.q1 = q2_.y;                                           \
                    } else {                                                                   \
                        f_.q = *(const uint32_t *)(ws + Q_OFF + blk_ * 16 + fgrp * 4);         \
                    }                                                                          \
                    if constexpr (RF::HB != 0) f_.h = *(const uint32_t *)(ws + H_OFF + blk_ * 4); \
                    if constexpr (RF::DB == 4) f_.dm = *(const h2 *)(ws + D_OFF + blk_ * 4);   \
                    else f_.d = *(const half_t *)(ws + D_OFF + blk_ * 2);                      \
                    wf = dequant_wfrag<WT>(f_, fgrp);                                          \
                }                                                                              \
                _Pragma("unroll") for (int b = 0; b < TM; b++)                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[a][b], 0, 0, 0); \
            }                                                                                  \
        }                                                                                      \
    }
// "my DMA of the older tile has landed" (NI newer instructions may remain in flight), then publish / retire via barrier
#define RING_SYNC(n_)                                                                          \
    {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory");                              \
        __builtin_amdgcn_s_barrier();                                                          \
        asm volatile("" ::: "memory");                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }

    RING_DMA(0, 0);
    { const int t1 = last < 1 ? last : 1; RING_DMA(1, t1); }
    RING_SYNC(NI);                                   // tile 0 landed everywhere
    int slot = 0;                                    // slot of tile kt; tile kt+2 goes to slot+2 (mod 3)
    for (int kt = 0; kt < nk; kt++) {
        const int t2 = kt + 2 < last ? kt + 2 : last;
        const int s2 = slot == 0 ? 2 : slot - 1;     // (slot + 2) % 3
        RING_DMA(s2, t2);
        RING_COMPUTE(slot);
        RING_SYNC(NI);                               // tile kt+1 landed everywhere; slot `slot` free again
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the redundant tail prefetches before LDS is released
#undef RING_DMA
#undef RING_COMPUTE
#undef RING_SYNC
    gemm_epilogue<EPI, TN, TM>(p, acc, n0 + wn * (BN / 2), m0 + wm * (BM / 2), frow, fgrp);
}

template <int WT, int BM, int EPI>
void launch_ring(const GemmParams & p, hipStream_t stream) {
    constexpr int BN = 128;
    using RF = RingFmt<WT>;
    constexpr size_t smem = 3 * ((size_t)BM * BK * 2 + (WT == W_F16 ? (size_t)BN * BK * 2 : (size_t)2 * BN * (RF::QB + RF::DB + RF::HB)));
    static bool attr_set = false;
    if (smem > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute((const void *)gemm_ring_kernel<WT, BM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_ring_kernel<WT, BM, EPI>), dim3(tiles_m * tiles_n), dim3(NTHREADS), smem, stream, p);
}

template <int WT, int BM, int BN, int EPI, bool DIRECT>
void launch_one(const GemmParams & p, hipStream_t stream) {
    constexpr bool WLDS = (WT == W_F16) || !DIRECT;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + BN - 1) / BN;
    const size_t smem = (size_t)2 * (BM + (WLDS ? BN : 0)) * BK * sizeof(half_t);
    hipLaunchKernelGGL((gemm_kernel<WT, BM, BN, EPI, DIRECT>), dim3(tiles_m * tiles_n), dim3(NTHREADS), smem, stream, p);
}

// tile code: variant * 1000000 + BM * 1000 + BN; variant 0 = weights staged through LDS, 1 = per-wave register fragments
template <int WT, int BM, int BN, int EPI>
void launch_ws(const GemmParams & p, hipStream_t stream) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.W.N + BN - 1) / BN;
    const size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(half_t);
    hipLaunchKernelGGL((gemm_ws_kernel<WT, BM, BN, EPI>), dim3(tiles_m * tiles_n), dim3(512), smem, stream, p);
}

template <int WT, int EPI>
void launch_tile(const GemmParams & p, int tile, hipStream_t stream) {
    if (tile / 1000000 == 4) {   // 3-slot LDS-DMA ring
        switch (tile % 1000000) {
        case 64128: launch_ring<WT, 64, EPI>(p, stream); break;
        default: launch_ring<WT, 128, EPI>(p, stream); break;
        }
        return;
    }
    if (tile / 1000000 == 3) {   // LDS-DMA staging
        switch (tile % 1000000) {
        case 64128: launch_dma<WT, 64, 128, EPI>(p, stream); break;
        case 128064: launch_dma<WT, 128, 64, EPI>(p, stream); break;
        case 64064: launch_dma<WT, 64, 64, EPI>(p, stream); break;
        default: launch_dma<WT, 128, 128, EPI>(p, stream); break;
        }
        return;
    }
    if (tile / 1000000 == 2) {   // wave-specialised
        switch (tile % 1000000) {
        case 64128: launch_ws<WT, 64, 128, EPI>(p, stream); break;
        default: launch_ws<WT, 128, 128, EPI>(p, stream); break;
        }
        return;
    }
    const bool direct = (tile / 1000000) == 1 && WT != W_F16;
    switch (tile % 1000000) {
    case 128128: direct ? launch_one<WT, 128, 128, EPI, (WT != W_F16)>(p, stream) : launch_one<WT, 128, 128, EPI, false>(p, stream); break;
    case 64128: direct ? launch_one<WT, 64, 128, EPI, (WT != W_F16)>(p, stream) : launch_one<WT, 64, 128, EPI, false>(p, stream); break;
    case 128064: direct ? launch_one<WT, 128, 64, EPI, (WT != W_F16)>(p, stream) : launch_one<WT, 128, 64, EPI, false>(p, stream); break;
    default: direct ? launch_one<WT, 64, 64, EPI, (WT != W_F16)>(p, stream) : launch_one<WT, 64, 64, EPI, false>(p, stream); break;
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

// Tile heuristic: biggest tile that still yields >= ~1 workgroup per CU (256 CUs).
int pick_tile(int M, int N) {
    auto wgs = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    if (M <= 64) return N >= 2048 ? 64128 : 64064;
    if (wgs(128, 128) >= 384) return 128128;
    if (wgs(64, 128) >= 256) return 64128;
    if (wgs(128, 128) >= 200) return 128128;
    return 64064;
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

void launch_gemm(const GemmParams & p, int epilogue, int tile, hipStream_t stream) {
    if (p.M <= 0) return;
    if (tile == 0) tile = pick_tile(p.M, p.W.N);
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
