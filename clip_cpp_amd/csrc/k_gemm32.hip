// k_gemm32.hip — the fp16-output weight GEMMs of a large batch (q/k/v, FFN-up) for gfx950 (CDNA4) on v_mfma_f32_32x32x16_f16.
//
//   out[M][N] (fp16) = epilogue( X[M][K] (fp16) · W16[N][K]^T (fp16 panel) + bias )          M >= ~8000 rows, K >= 128
//
// Why another kernel (round 6; profiles/r06_experiments.txt): every other GEMM in the tree issues v_mfma_f32_16x16x32_f16 and spends most of
// its issue slots on the non-MFMA stream of a K step.  This one is built on the 32 x 32 x 16 instruction (half the MFMA instructions per
// FLOP, 32 cycles each on its SIMD: ~5 issue slots under every MFMA, 0.5 fragment reads + 0.25 LDS-DMA requests per MFMA to fill them):
//   * four waves, one per SIMD, each a (32 WM) x 128 sub-tile of a (64 WM) x 256 workgroup tile, WM in {4, 5}: 256 / 320 accumulator
//     registers per lane (AGPRs; the fifth row block of WM = 5 in VGPRs).  320 x 256 turns FFN-up of the BASELINE batch
//     (12800 x 3072: 600 tiles of 256 x 256 = 2.34 rounds of 256 CUs) into 480 tiles = 1.88 rounds;
//   * K-tiles of 64 in TWO LDS stages (2 x 64 / 72 KB) filled by global_load_lds_dwordx4 only, 8 rows x 128 B per 1 KB piece — whole cache
//     lines of the operands —, XOR swizzle (chunk ^= (row >> 1) & 7) on the per-lane SOURCE address -> conflict-free ds_read_b128 of
//     32-row fragments (SQ_LDS_BANK_CONFLICT = 0);
//   * the 16-18 requests of K-tile t + 1 are spread one per ~3 MFMAs over step 3 of tile t - 1 and steps 0-1 of tile t (a request stalls
//     the issuing wave ~75 cycles: bunched, they idle the matrix pipe); step 2 is their landing time;
//   * ONE barrier per K-tile (64-80 MFMAs per wave), between steps 2 and 3: every wave has read all of tile t, tile t + 1 has landed;
//   * the fragments of step s + 1 are read under the MFMAs of step s (two register sets);
//   * ONE loop body with the stage offsets in SGPRs (separately unrolled bodies per stage made hipcc move the accumulators through scratch);
//   * epilogue: the tile's bias / c columns and its rows' (mean, rstd) wait in LDS since the prologue; two FMAs per output (LayerNorm-fold
//     consumer rstd (acc - mean c) + b', gemm_common.h; Q scale folded into the coefficients) + GELU, fp16 through the freed ring so that
//     every global store writes full 128-byte lines.
// What it reaches and where it is used: in cycles its K loop runs at 84 % of the matrix pipe's issue rate, but the chip lowers its clock to
// 1.55-1.75 GHz while matrix work and operand traffic coincide (2.2 GHz without the traffic), and with one workgroup per CU nothing runs
// under a tile's prologue and epilogue (45 % of its cycles).  Per launch 5-13 % ahead of the two-per-CU kernels on the ViT-B/32 batch shapes
// (0.30 of the MFMA peak on FFN-up), level with k_gemm4.hip on ViT-L/14 shapes; in a two-tower step it LOSES 0.4-2.7 % because the other
// tower's workgroups can no longer share its CUs.  pick_tile (k_gemm.hip) therefore takes it only when the device is not shared.
// Numerics: same operands and f32 accumulation as the other kernels; the 32 x 32 x 16 instruction sums k in its own order, so results
// agree with the 16 x 16 x 32 kernels to f32 rounding of the sums (<= 1 fp16 ulp after the output rounding; tests/test_gpu_gemm32.py).
//
// Reference ops replaced: ggml_mul_mat with a weight operand + bias add (+ scale / gelu), clip.cpp:1360-1380 (q/k/v), :1407-1413 (FFN-up),
// text :1079-1095, :1127-1131.

#include <type_traits>

#include "gemm_common.h"

namespace clipamd {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int NT32 = 256;

// WM x WN fragments of 32 x 32 per wave, 2 x 2 waves: workgroup tile (64 WM) x (64 WN); two stages of K-tiles of 64
template <int WM, int WN> struct G32 {
    static constexpr int BM = 64 * WM, BN = 64 * WN;
    static constexpr int XB = BM * 128;                // bytes of the X tile of one stage: [BM][64] fp16
    static constexpr int WB = BN * 128;
    static constexpr int STAGE = XB + WB;
    static constexpr int NPX = BM / 32, NPW = BN / 32; // 1 KB LDS-DMA pieces (8 rows x 128 B) per wave per K-tile
    static constexpr int NP = NPX + NPW;
    static constexpr int RINGB = 2 * STAGE;
    static constexpr int LDS = RINGB + BM * 8 + BN * 8;   // + (mean, rstd) of the tile's rows (LayerNorm fold) + (bias, c) of its columns
    static constexpr int NM = WM * WN;                 // MFMAs per k16 step
    // requests of a K-tile: part A under step 3 of the tile two before it (right behind the barrier that frees the stage), parts B and C
    // under steps 0 and 1 of the tile before it; step 2 is the landing time of the last ones
    static constexpr int NA = (NP + 2) / 3, NB = NA + (NP - NA + 1) / 2;
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F && f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void raw_barrier32() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// accumulators: AGPRs, except the fifth row block of the 320-row tile (VGPRs: a wave addresses 256 of each)
template <bool AG>
__device__ __forceinline__ void mfma32(f16v & c, const h8 & a, const h8 & b) {
#ifdef G32_ABL_NOMFMA
    asm volatile("s_nop 0" : "+a"(c) : "v"(a), "v"(b) : "memory");
#else
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b) : "memory");
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "memory");
#endif
}

// one LDS-DMA piece: 64 lanes x 16 B from gbase + voff (per lane) to LDS [ldsbase + CST, + 1 KB) lane-linear
// (a function, not a macro: clang does not capture variables that appear only as asm operands inside a generic lambda)
template <int CST>
__device__ __forceinline__ void g32_dma(unsigned voff, const char * gbase, unsigned ldsbase) {
#ifdef G32_ABL_NODMA
    asm volatile("" ::"v"(voff), "s"(gbase), "s"(ldsbase) : "memory");
#else
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(ldsbase), "i"(CST) : "memory", "m0", "scc");
#endif
}

template <int WM, int WN>
struct G32Ctx {
    const unsigned char * smem;
    const char * xg, * wg;            // global bases of the operands (bytes), uniform
    unsigned xd0, wd0;                // LDS byte address of this wave's first X / W piece in stage 0
    unsigned xoff[G32<WM, WN>::NPX], woff[G32<WM, WN>::NPW];
    int xa[4], wa[4];                 // fragment read offsets (bytes inside a stage) of k16 steps 0 .. 3
};

// request i of K-tile kt (i < NPX: X piece i, else W piece i - NPX) into the stage at byte offset sbase
template <int WM, int WN, int I>
__device__ __forceinline__ void g32_piece(const G32Ctx<WM, WN> & c, int kt, unsigned sbase) {
    using G = G32<WM, WN>;
    if constexpr (I < G::NPX) g32_dma<I * 1024>(c.xoff[I], c.xg + (size_t)kt * 128, c.xd0 + sbase);
    else g32_dma<(I - G::NPX) * 1024>(c.woff[I - G::NPX], c.wg + (size_t)kt * 128, c.wd0 + sbase);
}
template <int WM, int WN, int I0, int I1>
__device__ __forceinline__ void g32_pieces(const G32Ctx<WM, WN> & c, int kt, unsigned sbase) {
    static_for<I0, I1>([&](auto I_) { g32_piece<WM, WN, decltype(I_)::value>(c, kt, sbase); });
}

// the request (0 .. nd - 1) issued in front of MFMA g of a step of nm MFMAs: request k sits at (k nm + nm / 2) / nd; -1 = none
constexpr int slot_req(int g, int nm, int nd) {
    for (int k = 0; k < nd; k++) if ((k * nm + nm / 2) / nd == g) return k;
    return -1;
}

// one k16 step: NM MFMAs on (wm, xm), the fragments (wr, xr) of the next step read under them (rbase: lane offsets of that step + its stage),
// requests [I0, I1) of K-tile kt into stage sbase spread over the step (dma: uniform condition)
template <int WM, int WN, int I0, int I1>
__device__ __forceinline__ void g32_step(const G32Ctx<WM, WN> & c, f16v (&acc)[WN][WM], const h8 (&wm_)[WN], const h8 (&xm_)[WM], h8 (&wr)[WN], h8 (&xr)[WM],
                                         const unsigned char * xrd, const unsigned char * wrd, bool dma, int kt, unsigned sbase) {
    constexpr int NM = WM * WN, NR = WM + WN, ND = I1 - I0;
    static_for<0, NM>([&](auto G_) {
        constexpr int g = decltype(G_)::value;
        constexpr int a = g / WM, b = g % WM;
#ifndef G32_ABL_NOREAD
        if constexpr (g < WM) xr[g] = *(const h8 *)(xrd + g * 4096);
        else if constexpr (g < NR) wr[g - WM] = *(const h8 *)(wrd + (g - WM) * 4096);
#endif
        if constexpr (slot_req(g, NM, ND) >= 0) {
            if (dma) g32_piece<WM, WN, I0 + slot_req(g, NM, ND)>(c, kt, sbase);
        }
        mfma32<(b < 4)>(acc[a][b], wm_[a], xm_[b]);
    });
}

template <int WM, int WN, int EPI>
__global__ void __launch_bounds__(NT32, 1) gemm32_kernel(const GemmParams p) {
    using G = G32<WM, WN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = G::BM, BN = G::BN;
    static_assert(EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_QGELU_F16, "fp16-output epilogues only");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

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
    const int T = p.W.Kpad / 64;                      // >= 2 (gemm32_supported)

    G32Ctx<WM, WN> c;
    c.smem = smem;
    c.xg = (const char *)p.A;
    c.wg = (const char *)p.W.w16;
    {
        const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
        c.xd0 = lds0 + wave * (G::NPX * 1024);
        c.wd0 = lds0 + G::XB + wave * (G::NPW * 1024);
        // LDS-DMA sources: lane l of a piece writes row l >> 3, 16-byte position l & 7 of an [8][128 B] block — whole cache lines of the
        // operand; it fetches the chunk that the swizzle (chunk ^= (row >> 1) & 7) puts there
        const int prow = lane >> 3;
#pragma unroll
        for (int i = 0; i < G::NPX; i++) {
            const int tr = (wave * G::NPX + i) * 8 + prow;
            int gm = m0 + tr;
            gm = gm < p.M ? gm : p.M - 1;
            c.xoff[i] = (unsigned)gm * (unsigned)(p.lda * 2) + (((lane & 7) ^ ((tr >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < G::NPW; i++) {
            const int tr = (wave * G::NPW + i) * 8 + prow;
            int gn = n0 + tr;
            gn = gn < p.W.Npad ? gn : p.W.Npad - 1;
            c.woff[i] = (unsigned)gn * (unsigned)(p.W.Kpad * 2) + (((lane & 7) ^ ((tr >> 1) & 7)) << 4);
        }
        // fragment reads: lane l holds row l & 31, k-chunk 2 s + (l >> 5) of k16 step s; 128-byte rows, chunk ^= (row >> 1) & 7:
        // the 16 lanes of every ds_read_b128 group fall on 16 different 16-byte slots of the 256-byte bank row
        const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            c.xa[s] = (wm * (WM * 32) + l31) * 128 + (((2 * s + hi) ^ sw) << 4);
            c.wa[s] = G::XB + (wn * (WN * 32) + l31) * 128 + (((2 * s + hi) ^ sw) << 4);
        }
    }

    float2 * const ln_rs = (float2 *)(smem + G::RINGB);
    float2 * const bc = (float2 *)(smem + G::RINGB + BM * 8);
    const bool ln = p.ln_c != nullptr;
#ifdef CLIPAMD_G8_TIMING   // tuning builds: per-workgroup phase stamps (shader clock / 100 MHz clock) into the split-K workspace: start, loop, epilogue, end
    unsigned long long * stamp = (unsigned long long *)p.sk_ws + (size_t)blockIdx.x * 8;
    const bool stamper = p.sk_ws && tid == 0;
    if (stamper) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
#endif

    f16v acc[WN][WM];
    h8 w0[WN], x0[WM], w1[WN], x1[WM];

    // prologue.  K-tile 0 is requested first and lands under the zeroing of the accumulators, the row statistics (LayerNorm fold, consumer
    // half — gemm_common.h: thread t reduces the statistics of row m0 + t (+ 256) to (mean - mu, rstd) and parks the pair behind the ring)
    // and the staging of the tile's bias / c columns in LDS (the epilogue then waits for no global load); then part A of K-tile 1.
    g32_pieces<WM, WN, 0, G::NP>(c, 0, 0);
    {
        int n = n0 + tid;
        n = n < p.W.N ? n : p.W.N - 1;
        if (tid < BN) bc[tid] = make_float2(p.bias ? p.bias[n] : 0.f, ln ? p.ln_c[n] : 0.f);
    }
#pragma unroll
    for (int a = 0; a < WN; a++)
#pragma unroll
        for (int b = 0; b < WM; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    // the zeroes must be IN the accumulator registers before the first (asm) MFMA reads them as SrcC (k_gemm4.hip)
#pragma unroll
    for (int a = 0; a < WN; a++)
#pragma unroll
        for (int b = 0; b < WM; b++) {
            if (b < 4) asm volatile("" : "+a"(acc[a][b]));
            else asm volatile("" : "+v"(acc[a][b]));
        }
    if (ln) {
        ln_rs[tid] = ln_row_centred(p, m0 + tid < p.M ? m0 + tid : p.M - 1, n0 == 0 && m0 + tid < p.M);
        if (BM > NT32 && tid < BM - NT32) {
            const int r2 = m0 + NT32 + tid;
            ln_rs[NT32 + tid] = ln_row_centred(p, r2 < p.M ? r2 : p.M - 1, n0 == 0 && r2 < p.M);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                // K-tile 0 and the loads above have landed (hipcc's own counting restarts from zero)
    if (T > 1) g32_pieces<WM, WN, 0, G::NA>(c, 1, G::STAGE);
    raw_barrier32();
#ifndef G32_ABL_NOREAD
#pragma unroll
    for (int b = 0; b < WM; b++) x0[b] = *(const h8 *)(smem + c.xa[0] + b * 4096);
#pragma unroll
    for (int a = 0; a < WN; a++) w0[a] = *(const h8 *)(smem + c.wa[0] + a * 4096);
#else
#pragma unroll
    for (int b = 0; b < WM; b++) { x0[b] = (h8)(_Float16)0.f; x1[b] = (h8)(_Float16)0.f; }
#pragma unroll
    for (int a = 0; a < WN; a++) { w0[a] = (h8)(_Float16)0.f; w1[a] = (h8)(_Float16)0.f; }
#endif

#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
    // K-tile t in stage cur (ONE body for every tile: separately unrolled bodies per stage / tail case made hipcc give the accumulators
    // different homes on different paths and copy them through scratch at the joins):
    //   step 0: MFMAs k 0-15  | reads of step 1 | requests part B of tile t + 1 (stage oth: free since the previous barrier)
    //   step 1: MFMAs k 16-31 | reads of step 2 | requests part C of tile t + 1
    //   step 2: MFMAs k 32-47 | reads of step 3
    //   wait (own requests of tile t + 1 landed, own reads of this stage complete) + barrier: tile t + 1 visible, stage cur free
    //   step 3: MFMAs k 48-63 | reads of tile t + 1's step 0 | requests part A of tile t + 2 (into cur)
    unsigned cur = 0, oth = G::STAGE;
    for (int t = 0; t < T; t++) {
        const bool nx1 = t + 1 < T, nx2 = t + 2 < T;       // (uniform)
        g32_step<WM, WN, G::NA, G::NB>(c, acc, w0, x0, w1, x1, smem + (c.xa[1] + (int)cur), smem + (c.wa[1] + (int)cur), nx1, t + 1, oth);
        g32_step<WM, WN, G::NB, G::NP>(c, acc, w1, x1, w0, x0, smem + (c.xa[2] + (int)cur), smem + (c.wa[2] + (int)cur), nx1, t + 1, oth);
        g32_step<WM, WN, 0, 0>(c, acc, w0, x0, w1, x1, smem + (c.xa[3] + (int)cur), smem + (c.wa[3] + (int)cur), false, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0070);
        raw_barrier32();
        g32_step<WM, WN, 0, G::NA>(c, acc, w1, x1, w0, x0, smem + (c.xa[0] + (int)oth), smem + (c.wa[0] + (int)oth), nx2, t + 2, cur);
        const unsigned r0 = cur; cur = oth; oth = r0;
    }
    // (the asm MFMAs are opaque to hipcc's hazard recogniser: let the last ones retire before the accumulators are read)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int a = 0; a < WN; a++)
#pragma unroll
        for (int b = 0; b < WM; b++) {
            if (b < 4) asm volatile("" : "+a"(acc[a][b]));
            else asm volatile("" : "+v"(acc[a][b]));
        }
    __builtin_amdgcn_s_waitcnt(0x0070);

    // ---- epilogue.  D[i][j] of a 32 x 32 fragment: i = weight row n = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3), j = activation row m = lane & 31.
    // 64 output columns at a time: arithmetic on the accumulators, fp16 into this wave's staging rows (136-byte pitch: conflict-free 8-byte
    // writes), re-read row-contiguous -> every global store instruction writes 8 full 128-byte lines.
#ifdef CLIPAMD_G8_TIMING
    if (stamper) stamp[2] = __builtin_amdgcn_s_memtime();
#endif
#ifdef G32_ABL_NOEPI
    if (p.M > 0) {
#ifdef CLIPAMD_G8_TIMING
        if (stamper) { stamp[3] = __builtin_amdgcn_s_memtime(); stamp[5] = stamp[3]; stamp[6] = __builtin_amdgcn_s_memrealtime(); }
#endif
        return;
    }
#endif
    raw_barrier32();                                   // every wave is done with the ring: it becomes the staging area
    unsigned lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int l31 = (int)(lane_e & 31u), hi = (int)(lane_e >> 5);
    constexpr int RS = 68;
    half_t * const stage = (half_t *)smem + wave * (WM * 32) * RS;
    const int mb = m0 + wm * (WM * 32);
    // v = rstd (acc - mean c) + b'  as  fma(acc, rstd, fma(-c, mean rstd, b'))  (plain: rstd = 1, mean = 0): two FMAs per output; the Q scale
    // (clip.cpp:1363: after the bias) multiplies rstd, c and b' of its columns instead of the outputs
    float rstd[WM], mrs[WM];
#pragma unroll
    for (int b = 0; b < WM; b++) {
        const float2 mr = ln ? ln_rs[wm * (WM * 32) + b * 32 + l31] : make_float2(0.f, 1.f);
        rstd[b] = mr.y;
        mrs[b] = mr.x * mr.y;
    }
    const int N = p.W.N;
    // output rows through a buffer descriptor: rows past M fall outside it and are dropped by the hardware (no exec masks in the store loop)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.M * p.ldc * 2), 0x00020000);
#ifdef G32_EPI_DIRECT   // measured alternative (profiles/r06_experiments.txt section 1): fewer cycles, 23 % more HBM write traffic; not the default
    // Straight from the registers: a lane holds 4 consecutive columns (8 j + 4 hi ...) of row l31 per j; v_permlane32_swap pairs j with j + 1
    // so that the lower half-wave holds the 8 columns 16 jp .. + 7 and the upper half-wave 16 jp + 8 .. + 15 of its row: one 16-byte store per
    // lane, 32 bytes per row per instruction, no LDS round trip and no wait inside the epilogue.
    (void)stage;
    const unsigned obase = (unsigned)(mb + l31) * (unsigned)(p.ldc * 2) + (unsigned)(n0 + wn * (WN * 32) + 8 * hi) * 2;
#pragma unroll
    for (int a = 0; a < WN; a++) {
        if (n0 + wn * (WN * 32) + a * 32 >= N) continue;      // (uniform; whole 64-column slabs: gemm32_supported)
#pragma unroll
        for (int jp = 0; jp < 2; jp++) {
            float bias[2][4], cc[2][4], qf[2];
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int nl = wn * (WN * 32) + a * 32 + 8 * (2 * jp + jj) + 4 * hi;        // column inside the tile
                const f4 b01 = *(const f4 *)(bc + nl), b23 = *(const f4 *)(bc + nl + 2);      // (bias, c) pairs of 4 columns
                qf[jj] = (EPI == EPI_F16 && n0 + nl < p.qcols) ? p.qscale : 1.0f;             // (qcols % 4 == 0: gemm32_supported)
                bias[jj][0] = b01[0] * qf[jj]; bias[jj][1] = b01[2] * qf[jj]; bias[jj][2] = b23[0] * qf[jj]; bias[jj][3] = b23[2] * qf[jj];
                cc[jj][0] = b01[1] * qf[jj]; cc[jj][1] = b01[3] * qf[jj]; cc[jj][2] = b23[1] * qf[jj]; cc[jj][3] = b23[3] * qf[jj];
            }
#pragma unroll
            for (int b = 0; b < WM; b++) {
                uint32_t pk[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const float rq = EPI == EPI_F16 ? rstd[b] * qf[jj] : rstd[b];
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        v[r] = __builtin_fmaf(acc[a][b][4 * (2 * jp + jj) + r], rq, __builtin_fmaf(-cc[jj][r], mrs[b], bias[jj][r]));
                        if constexpr (EPI == EPI_GELU_F16) v[r] = gelu_tanh(v[r]);
                        else if constexpr (EPI == EPI_QGELU_F16) v[r] = gelu_quick(v[r]);
                    }
                    pk[jj][0] = h2u((h2){(_Float16)v[0], (_Float16)v[1]});
                    pk[jj][1] = h2u((h2){(_Float16)v[2], (_Float16)v[3]});
                }
                asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(pk[0][0]), "+v"(pk[1][0]));
                asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(pk[0][1]), "+v"(pk[1][1]));
                const u32x4 o = (u32x4){pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, obase + (unsigned)(b * 32) * (unsigned)(p.ldc * 2) + (unsigned)(a * 32 + 16 * jp) * 2, 0, 0);
            }
        }
    }
#else
    const int rrow = (int)(lane_e >> 3), rchunk = (int)(lane_e & 7u);
#pragma unroll
    for (int hn = 0; hn < WN / 2; hn++) {
        const int nbh = n0 + wn * (WN * 32) + hn * 64;
        if (nbh >= N) continue;                        // (uniform; N % 64 == 0: gemm32_supported)
#pragma unroll
        for (int ai = 0; ai < 2; ai++) {
            const int a = 2 * hn + ai;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int nl = wn * (WN * 32) + hn * 64 + ai * 32 + 8 * j + 4 * hi;       // column inside the tile
                const f4 b01 = *(const f4 *)(bc + nl), b23 = *(const f4 *)(bc + nl + 2);    // (bias, c) pairs of 4 columns
                float bias[4] = {b01[0], b01[2], b23[0], b23[2]}, cc[4] = {b01[1], b01[3], b23[1], b23[3]};
                float rq[WM];
#pragma unroll
                for (int b = 0; b < WM; b++) rq[b] = rstd[b];
                if constexpr (EPI == EPI_F16) {
                    const float qf = n0 + nl < p.qcols ? p.qscale : 1.0f;                   // (qcols % 4 == 0: gemm32_supported)
#pragma unroll
                    for (int r = 0; r < 4; r++) { bias[r] *= qf; cc[r] *= qf; }
#pragma unroll
                    for (int b = 0; b < WM; b++) rq[b] *= qf;
                }
#pragma unroll
                for (int b = 0; b < WM; b++) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        v[r] = __builtin_fmaf(acc[a][b][4 * j + r], rq[b], __builtin_fmaf(-cc[r], mrs[b], bias[r]));
                        if constexpr (EPI == EPI_GELU_F16) v[r] = gelu_tanh(v[r]);
                        else if constexpr (EPI == EPI_QGELU_F16) v[r] = gelu_quick(v[r]);
                    }
                    const h2 lo = (h2){(_Float16)v[0], (_Float16)v[1]};
                    const h2 hh = (h2){(_Float16)v[2], (_Float16)v[3]};
                    *(uint2 *)(stage + (b * 32 + l31) * RS + ai * 32 + 8 * j + 4 * hi) = make_uint2(h2u(lo), h2u(hh));
                }
            }
        }
        const unsigned obase = (unsigned)(mb + rrow) * (unsigned)(p.ldc * 2) + (unsigned)(nbh + rchunk * 8) * 2;
#pragma unroll
        for (int i = 0; i < WM * 4; i++) {
            const u32x4 v = *(const u32x4 *)(stage + (i * 8 + rrow) * RS + rchunk * 8);
            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, obase + (unsigned)(i * 8) * (unsigned)(p.ldc * 2), 0, 0);
        }
    }
#endif
#ifdef CLIPAMD_G8_TIMING
    if (stamper) {
        stamp[3] = __builtin_amdgcn_s_memtime();       // stores issued (not necessarily landed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[5] = __builtin_amdgcn_s_memtime();       // this wave's stores acknowledged
        stamp[6] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int WM, int WN, int EPI>
void launch32(const GemmParams & p, hipStream_t stream) {
    using G = G32<WM, WN>;
    const int tiles_m = (p.M + G::BM - 1) / G::BM, tiles_n = (p.W.N + G::BN - 1) / G::BN;
    static unsigned long long lds_ok = 0;
    opt_in_dynamic_lds(gemm32_kernel<WM, WN, EPI>, (size_t)G::LDS, lds_ok);
    hipLaunchKernelGGL((gemm32_kernel<WM, WN, EPI>), dim3(tiles_m * tiles_n), dim3(NT32), (size_t)G::LDS, stream, p);
}

template <int WM, int WN>
void launch32_epi(const GemmParams & p, int epilogue, hipStream_t stream) {
    switch (epilogue) {
    case EPI_F16: launch32<WM, WN, EPI_F16>(p, stream); break;
    case EPI_GELU_F16: launch32<WM, WN, EPI_GELU_F16>(p, stream); break;
    case EPI_QGELU_F16: launch32<WM, WN, EPI_QGELU_F16>(p, stream); break;
    }
}

}  // namespace

// what the kernel takes: an fp16-output epilogue on an fp16 weight panel, K >= 128, whole 64-column slabs, 16-byte aligned rows,
// operands and output addressable with 32-bit byte offsets, Q-scale columns in groups of 4
bool gemm32_supported(const GemmParams & p, int epilogue) {
    if (epilogue != EPI_F16 && epilogue != EPI_GELU_F16 && epilogue != EPI_QGELU_F16) return false;
    if (p.W.wtype != W_F16 || !p.W.w16 || p.W.Kpad < 128 || (p.W.N & 63) || (p.ldc & 7) || (p.lda & 7) || (p.qcols & 3)) return false;
    if ((size_t)p.M * p.lda * 2 >= ((size_t)1 << 31) || (size_t)p.W.Npad * p.W.Kpad * 2 >= ((size_t)1 << 31) || (size_t)p.M * p.ldc * 2 >= ((size_t)1 << 31)) return false;
    return true;
}

// form: 4 = 256 x 256 tiles, 5 = 320 x 256 (one workgroup per CU either way)
void launch_gemm32(const GemmParams & p, int epilogue, int form, hipStream_t stream) {
    if (form == 5) launch32_epi<5, 4>(p, epilogue, stream);
    else launch32_epi<4, 4>(p, epilogue, stream);
}

}  // namespace clipamd
