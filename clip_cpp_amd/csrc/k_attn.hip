// k_attn.hip — fused multi-head attention softmax(Q K^T) V on fp16 MFMA for short sequences.
//
// Replaces, per (sequence, head), the ggml sub-graph of reference clip.cpp:1382-1388 (vision) and
// :1100-1108 (text, with ggml_diag_mask_inf causal mask :1101):
//     KQ = mul_mat(K, Q); [mask]; soft_max(KQ); KQV = mul_mat(V^T, KQ); permute/cont/cpy
// including the head split/merge permutes (the kernel reads Q/K/V straight out of the fused
// [rows][3h] projection output and writes the merged [rows][h] context).  Q arrives pre-scaled by
// 1/sqrt(d_head) (GEMM epilogue; the reference scales Q after the bias, clip.cpp:1363).
//
// One workgroup (4 waves) per (sequence, head).  K ([T][dh]) and V^T ([dh][T]) of the head are staged
// once in LDS (zero padded to the compile-time tile count); each wave then owns 16-query blocks:
//   S^T = K · Q^T        v_mfma_f32_16x16x32_f16, A = K rows (keys), B = Q rows (queries)
//                        -> every lane holds, for ONE query (lane & 15), 4 keys per 16-key tile: the
//                        whole score row lives in the registers of the 4 lanes {q, q+16, q+32, q+48}
//   softmax              in registers; row max / row sum = local reduce + 2 wave shuffles (xor 16, 32)
//   O = P · V            the exp()'d scores, packed to fp16, ARE the MFMA A operand of the second
//                        contraction (keys of two adjacent 16-key tiles form one K=32 slice; V^T is read
//                        with the same key permutation), so P never touches LDS or HBM.
// Scores are never materialised in HBM (the reference materialises [T,T,n_head*B] f32).
// Everything that indexes registers (key tiles NT, head-dim k-steps DKS, output d tiles DT) is a template
// parameter and the MFMA chains are unconditional: padded keys are masked to -inf / multiplied by zero.
// T <= 288 for every d_head in {32,64,80,88,96,104} (all 224-px models and every text length); T <= 592 for d_head <= 64
// (ViT-L/14 at 336 px: T = 577, 154.6 KB of LDS); longer sequences are rejected by the launcher.

#include <cstdlib>

#include "kernels.h"

namespace clipamd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifdef CLIPAMD_ATTN_ABL
#define ATTN_ABL(x_) (x_)
#else
#define ATTN_ABL(x_) false
#endif

struct AttnParams {
    int debug = 0;        // -DCLIPAMD_ATTN_ABL tuning builds (CLIP_AMD_ATTN_DEBUG): 1 no K / V staging loads, 2 no query blocks (staging only), 4 no output stores
    const half_t * qkv;   // [rows][3h]
    half_t * out;         // [rows][h]
    const int * seq_start;
    int T_uniform;
    int h, n_head;
    int causal;
};

// Everything a wave needs to process query blocks of one (sequence, head): LDS tiles, global Q / output rows.
template <int NT, int DKS, int DT>      // (DHR, the real head size when it is not DT * 16, is a parameter of the functions below)
struct AttnTile {
    const half_t * Ks;
    const half_t * Vt;
    const half_t * Qg;
    half_t * Og;          // output rows of this sequence, already offset to the head's columns
    int ld, h, len, causal, fq, fg;
    bool nostore = false;
};

// QB consecutive 16-query blocks starting at block qb0 (blocks beyond the sequence are computed on clamped rows and not stored).
template <int NT, int DKS, int DT, int QB, int DHR = DT * 16>
__device__ __forceinline__ void attn_blocks(const AttnTile<NT, DKS, DT> & t, int qb0) {
    constexpr int DKP = DKS * 32;
    constexpr bool SWZ = NT > 18;
    constexpr int KSTRIDE = SWZ ? DKP : DKP + 8;
    constexpr int NPR = (NT + 1) / 2;
    constexpr int VSTRIDE = NPR * 32 + 8;
    constexpr int DH = DHR;                          // real head size (a multiple of 8): 88 / 104 leave the last 16-wide output tile half empty
    const int fq = t.fq, fg = t.fg, len = t.len;
    // Q fragments (MFMA B operand): query row qb*16+fq, d = kk*32 + fg*8 .. +7
    h8 qf[QB][DKS];
    int qrow[QB];
#pragma unroll
    for (int j = 0; j < QB; j++) {
        qrow[j] = (qb0 + j) * 16 + fq;
        const int qclamped = qrow[j] < len ? qrow[j] : len - 1;
#pragma unroll
        for (int kk = 0; kk < DKS; kk++) {
            const int d0 = kk * 32 + fg * 8;
            u32x4 v = (u32x4){0u, 0u, 0u, 0u};
            if (d0 < DH) v = *(const u32x4 *)(t.Qg + (size_t)qclamped * t.ld + d0);
            qf[j][kk] = __builtin_bit_cast(h8, v);
        }
    }
    // ---- S^T tiles (unconditional MFMA chain); one K fragment read feeds QB MFMAs ----
    f4 s[QB][NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
#pragma unroll
        for (int j = 0; j < QB; j++) s[j][kt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DKS; kk++) {
            const int kch = kk * 4 + fg;
            const h8 kf = *(const h8 *)(t.Ks + (kt * 16 + fq) * KSTRIDE + (SWZ ? (kch ^ (fq & 7)) : kch) * 8);
#pragma unroll
            for (int j = 0; j < QB; j++) s[j][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[j][kk], s[j][kt], 0, 0, 0);
        }
        // keep at most 4 key tiles (8 fragment reads) in flight: fully hoisted, the reads of all NT tiles cost
        // 4*2*NT VGPRs (NT = 18: 488 registers, one workgroup per CU)
        if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- mask + softmax.  lane holds query fq, keys kt*16 + fg*4 + r ----
    float inv[QB];
    h8 pk[QB][NPR];
#pragma unroll
    for (int j = 0; j < QB; j++) {
        const int kmax = t.causal ? (qrow[j] < len - 1 ? qrow[j] : len - 1) : len - 1;   // last visible key
        const int kfull = t.causal ? 0 : (len >> 4);     // key tiles below kfull are entirely visible (uniform): no masking work
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            if (kt >= kfull) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = kt * 16 + fg * 4 + r;
                    s[j][kt][r] = key <= kmax ? s[j][kt][r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[j][kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // exp(s - mx) = exp2(s*log2(e) - mx*log2(e)): one fma + the hardware exp2 per score
        const float L2E = 1.44269504088896340736f;
        const float nmx = -mx * L2E;                     // key 0 is always visible -> mx is finite
        // the exp'd scores are packed to fp16 — the MFMA A operand of P V: keys of two adjacent 16-key tiles form one K = 32 slice — as
        // they are produced, so the f32 score registers die here instead of living through the second contraction (NT = 17 / 18 with
        // query-block pairs: 136 score registers + 32 output accumulators + fragments did not fit 256 and spilled)
        float sum = 0.f;
#pragma unroll
        for (int pr = 0; pr < NPR; pr++) {
            float e[8];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][2 * pr][r], L2E, nmx));
                e[4 + r] = (2 * pr + 1 < NT) ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][(2 * pr + 1 < NT) ? 2 * pr + 1 : 0][r], L2E, nmx)) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; r++) sum += e[r];
            if (2 * pr + 1 < NT) {
#pragma unroll
                for (int r = 0; r < 4; r++) sum += e[4 + r];
            }
#pragma unroll
            for (int r = 0; r < 8; r++) pk[j][pr][r] = (_Float16)e[r];
            // (bounded like the fragment prefetch above: hoisted, all 68 scaled differences of a block are computed before the first exp2)
            if (QB > 1 && NT >= 17 && (pr & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv[j] = 1.0f / sum;
        // one query block's softmax at a time: interleaved (hipcc's choice), both blocks' f32 scores AND both packed copies were live at once
        if (QB > 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- O = P V : pairs of key tiles form one K=32 slice; one V^T fragment read feeds QB MFMAs ----
    f4 o[QB][DT];
#pragma unroll
    for (int j = 0; j < QB; j++)
#pragma unroll
        for (int dt = 0; dt < DT; dt++) o[j][dt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < NPR; pr++) {
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const half_t * vrow = t.Vt + (dt * 16 + fq) * VSTRIDE + pr * 32 + fg * 4;
            const h4 v0 = *(const h4 *)(vrow);
            const h4 v1 = *(const h4 *)(vrow + 16);
            h8 vf;
            vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
            vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
#pragma unroll
            for (int j = 0; j < QB; j++) o[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pk[j][pr], vf, o[j][dt], 0, 0, 0);
        }
        if ((pr & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- normalise rows and store.  O layout: row (query) = fg*4 + r, col (d) = fq ----
#pragma unroll
    for (int j = 0; j < QB; j++) {
        float invr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) invr[r] = __shfl(inv[j], fg * 4 + r);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = (qb0 + j) * 16 + fg * 4 + r;
            if (q < len) {
                half_t * orow = t.Og + (size_t)q * t.h + fq;
#pragma unroll
                for (int dt = 0; dt < DT; dt++)
                    if ((DHR == DT * 16 || dt * 16 + fq < DHR) && !ATTN_ABL(t.nostore)) orow[dt * 16] = (_Float16)(o[j][dt][r] * invr[r]);
            }
        }
    }
}

// NT  = number of 16-key tiles (>= ceil(max_len/16)); DKS = 32-wide k-steps over the head dim (dh <= 32*DKS);
// DT = dh/16 output tiles.
template <int NT, int DKS, int DT, int DHR = DT * 16>
// occupancy: short sequences (T <= 80: ViT-B/32 images, every text) are latency-bound — stage, one barrier, a handful of MFMAs — so 4
// workgroups per CU (<= 128 VGPRs) instead of 2 hide twice the memory latency; long ones need the registers
__global__ void __launch_bounds__(256, (NT > 18 ? 1 : NT <= 5 ? 4 : 2)) attn_kernel(const AttnParams p) {
    static_assert(DHR % 8 == 0 && DHR <= DT * 16 && DHR > (DT - 1) * 16 && DHR <= DKS * 32, "head size: whole 16-byte chunks, DT output tiles, DKS k-steps");
    constexpr int DKP = DKS * 32;
    // Long sequences (NT > 18: 336-px models, T = 577) only fit the 160 KB LDS without the row padding: K rows are then
    // exactly 128 B with the 16-byte chunks XOR-swizzled by (key & 7) instead (same conflict-free ds_read_b128 pattern
    // as the GEMM tiles); needs DKP == 64.
    constexpr bool SWZ = NT > 18;
    static_assert(!SWZ || DKP == 64, "swizzled K layout is for d_head <= 64");
    constexpr int KSTRIDE = SWZ ? DKP : DKP + 8;     // halfs per K row  (+16 B pad: spreads ds_read_b128 over banks)
    constexpr int NPR = (NT + 1) / 2;                // key-tile pairs = K=32 slices of the P.V contraction
    constexpr int VSTRIDE = NPR * 32 + 8;            // halfs per V^T row; (VSTRIDE/2) = 4*odd -> conflict-free ds_read_b64
    constexpr int DH = DHR;                          // real head size; V^T rows DH .. DT*16-1 (88 / 104) are zeroed below and never stored
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t * Ks = (half_t *)smem_raw;                // [NT*16][KSTRIDE]
    half_t * Vt = Ks + NT * 16 * KSTRIDE;            // [DH][VSTRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: an SGPR)
    const int seq = blockIdx.x / p.n_head, head = blockIdx.x % p.n_head;
    int row0, len;
    if (p.seq_start) {
        row0 = p.seq_start[seq];
        len = p.seq_start[seq + 1] - row0;
    } else {
        row0 = seq * p.T_uniform;
        len = p.T_uniform;
    }
    const int ld = 3 * p.h;
    const half_t * Qg = p.qkv + (size_t)row0 * ld + head * DH;
    const half_t * Kg = Qg + p.h;
    const half_t * Vg = Qg + 2 * p.h;
    constexpr int DCH = DH / 8;                      // 16-byte chunks per head row

    // ---- stage K: Ks[key][0..DKP) (zero for key >= len and for columns >= DH) and V transposed: Vt[d][key], two keys per
    // thread so every LDS store is a full dword.  ALL global loads of the workgroup are issued before the first LDS store
    // (fully unrolled, unconditional clamped addresses + select): left as loops, each thread waited out one memory round
    // trip per 16-byte chunk (9 + 2x5 serial trips at T = 257 — two thirds of the kernel's time).
    {
        constexpr int KCH = DKP / 8;
        constexpr int KIT = (NT * 16 * KCH + 255) / 256;
        constexpr int NPAIR = NPR * 16;              // key pairs (covers NPR*32 keys, zero padded)
        constexpr int VIT = (NPAIR * DCH + 255) / 256;
        u32x4 kv[KIT], va[VIT], vb[VIT];
#pragma unroll
        for (int i = 0; i < KIT; i++) {
            const int it = tid + i * 256;
            const int key = it / KCH, c = it % KCH;
            const int kc = key < len ? key : len - 1, cc = c < DCH ? c : DCH - 1;
            kv[i] = ATTN_ABL(p.debug & 1) ? (u32x4){0u, 0u, 0u, 0u} : *(const u32x4 *)(Kg + (size_t)kc * ld + cc * 8);
        }
#pragma unroll
        for (int i = 0; i < VIT; i++) {
            const int it = tid + i * 256;
            const int kp = it % NPAIR, c = (it / NPAIR) < DCH ? (it / NPAIR) : DCH - 1;
            const int k0 = 2 * kp;
            va[i] = ATTN_ABL(p.debug & 1) ? (u32x4){0u, 0u, 0u, 0u} : *(const u32x4 *)(Vg + (size_t)(k0 < len ? k0 : len - 1) * ld + c * 8);
            vb[i] = ATTN_ABL(p.debug & 1) ? (u32x4){0u, 0u, 0u, 0u} : *(const u32x4 *)(Vg + (size_t)(k0 + 1 < len ? k0 + 1 : len - 1) * ld + c * 8);
        }
#pragma unroll
        for (int i = 0; i < KIT; i++) {
            const int it = tid + i * 256;
            const int key = it / KCH, c = it % KCH;
            if (it < NT * 16 * KCH) {
                const u32x4 v = (key < len && c < DCH) ? kv[i] : (u32x4){0u, 0u, 0u, 0u};
                *(u32x4 *)(Ks + key * KSTRIDE + (SWZ ? (c ^ (key & 7)) : c) * 8) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < VIT; i++) {
            const int it = tid + i * 256;
            const int kp = it % NPAIR, c = it / NPAIR;
            const int k0 = 2 * kp;
            if (it < NPAIR * DCH) {
                const u32x4 a = k0 < len ? va[i] : (u32x4){0u, 0u, 0u, 0u};
                const u32x4 b = k0 + 1 < len ? vb[i] : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint32_t av = (a[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                    const uint32_t bv = (b[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                    *(uint32_t *)(Vt + (c * 8 + e) * VSTRIDE + k0) = av | (bv << 16);
                }
            }
        }
        if constexpr (DHR < DT * 16) {               // the padding rows of V^T feed MFMAs whose outputs are dropped: keep them finite
            for (int i = tid; i < (DT * 16 - DHR) * (NPR * 16); i += 256)
                *(uint32_t *)(Vt + (DHR + i / (NPR * 16)) * VSTRIDE + 2 * (i % (NPR * 16))) = 0u;
        }
    }
    __syncthreads();

    const int fq = lane & 15, fg = lane >> 4;
    const int nqb = ATTN_ABL(p.debug & 2) ? 0 : (len + 15) >> 4;

    // Query blocks are processed in PAIRS where the register budget allows (QB = 2: every K / V^T fragment read from LDS
    // feeds two MFMAs, halving the LDS traffic that bounds this kernel at T = 257), the odd last block alone.
    constexpr int QB = (NT >= 7 && NT <= 18 && DT <= 6) ? 2 : 1;      // (d_head 104: 7 output tiles + 4 k-steps do not leave registers for block pairs)
    const AttnTile<NT, DKS, DT> t{Ks, Vt, Qg, p.out + (size_t)row0 * p.h + head * DH, ld, p.h, len, p.causal, fq, fg, (p.debug & 4) != 0};
    // gridDim.y workgroups share one (sequence, head): each stages K / V^T itself and takes every gridDim.y-th set of four work units
    // (few sequences x heads and many query blocks — one ViT-L/14 image is 16 workgroups of 17 query blocks otherwise)
    const int slot = wave + 4 * blockIdx.y, nslot = 4 * gridDim.y;
    if constexpr (QB == 2) {
        const int npair = nqb >> 1;
        for (int u = slot; u < npair; u += nslot) attn_blocks<NT, DKS, DT, 2, DHR>(t, 2 * u);
        // the odd last block: its wave rotates with the workgroup index — T = 257 is 8 pairs + 1 block, and with the extra block always on
        // wave npair % 4 the SIMD that hosts that wave of BOTH co-resident workgroups carries 10 blocks against 8 on the other three
        if ((nqb & 1) && slot == (npair + (int)blockIdx.x) % nslot) {
            // the lane coordinates are derived again (v_mbcnt, opaque to CSE) instead of being kept across the pair loop: at NT = 17 / 18 that
            // loop uses all 256 registers and the two values it had to keep for this call were spilled to scratch
            unsigned l2;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
            AttnTile<NT, DKS, DT> t1 = t;
            t1.fq = (int)(l2 & 15u);
            t1.fg = (int)(l2 >> 4);
            attn_blocks<NT, DKS, DT, 1, DHR>(t1, nqb - 1);
        }
    } else {
        for (int qb = slot; qb < nqb; qb += nslot) attn_blocks<NT, DKS, DT, 1, DHR>(t, qb);
    }
}

template <int NT, int DKS, int DT, int DHR = DT * 16>
void launch_inst(const AttnParams & p, int nseq, hipStream_t stream) {
    constexpr size_t smem = ((size_t)NT * 16 * (DKS * 32 + (NT > 18 ? 0 : 8)) + (size_t)DT * 16 * (((NT + 1) / 2) * 32 + 8)) * sizeof(half_t);
    static_assert(smem <= 160 * 1024, "attention tile does not fit the LDS");
    static unsigned long long lds_ok = 0;
    if (smem > 64 * 1024) opt_in_dynamic_lds(attn_kernel<NT, DKS, DT, DHR>, smem, lds_ok);
    // query split: only where (sequence, head) pairs alone leave most CUs idle and a pair has more work units than one workgroup's 4 waves
    constexpr int QB = (NT >= 7 && NT <= 18 && DT <= 6) ? 2 : 1;
    const int units = (NT + QB - 1) / QB;
    int qs = 1;
    if (nseq * p.n_head <= 64 && units > 4) qs = (units + 3) / 4 < 4 ? (units + 3) / 4 : 4;
#ifdef CLIPAMD_ATTN_ABL   // tuning builds: CLIP_AMD_ATTN_LDS_PAD=bytes of extra dynamic LDS per workgroup (e.g. 20000 at T = 257: ONE workgroup per CU instead of two)
    static const size_t pad = [] { const char * e = getenv("CLIP_AMD_ATTN_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
    if (pad && smem + pad <= 160 * 1024) {
        static unsigned long long lds_ok2 = 0;
        opt_in_dynamic_lds(attn_kernel<NT, DKS, DT, DHR>, smem + pad, lds_ok2);
        hipLaunchKernelGGL((attn_kernel<NT, DKS, DT, DHR>), dim3(nseq * p.n_head, qs), dim3(256), smem + pad, stream, p);
        return;
    }
#endif
    hipLaunchKernelGGL((attn_kernel<NT, DKS, DT, DHR>), dim3(nseq * p.n_head, qs), dim3(256), smem, stream, p);
}

template <int DKS, int DT, int DHR = DT * 16>
bool launch_nt(const AttnParams & p, int nseq, int nt, hipStream_t stream) {
    if (nt <= 1) launch_inst<1, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 2) launch_inst<2, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 3) launch_inst<3, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 4) launch_inst<4, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 5) launch_inst<5, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 7) launch_inst<7, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 10) launch_inst<10, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 14) launch_inst<14, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 17) launch_inst<17, DKS, DT, DHR>(p, nseq, stream);
    else if (nt <= 18) launch_inst<18, DKS, DT, DHR>(p, nseq, stream);
    else if constexpr (DKS == 2) {
        if (nt <= 37) launch_inst<37, DKS, DT, DHR>(p, nseq, stream);   // 336-px ViT-L/14: T = 577
        else return false;
    } else return false;
    return true;
}

}  // namespace

bool launch_attention(const half_t * qkv, half_t * out, int nseq, int T_uniform, const int * seq_start, int max_len,
                      int h, int n_head, bool causal, hipStream_t stream) {
    if (nseq <= 0) return true;
    const int dh = h / n_head;
    if (max_len > 592 || max_len <= 0) return false;
    AttnParams p;
    p.qkv = qkv;
    p.out = out;
    p.seq_start = seq_start;
    p.T_uniform = T_uniform;
    p.h = h;
    p.n_head = n_head;
    p.causal = causal ? 1 : 0;
#ifdef CLIPAMD_ATTN_ABL
    { const char * e = getenv("CLIP_AMD_ATTN_DEBUG"); p.debug = e ? atoi(e) : 0; }
#endif
    const int nt = (max_len + 15) / 16;
    switch (dh) {
    case 32: return launch_nt<1, 2>(p, nseq, nt, stream);
    case 64: return launch_nt<2, 4>(p, nseq, nt, stream);
    case 80: return launch_nt<3, 5>(p, nseq, nt, stream);
    case 88: return launch_nt<3, 6, 88>(p, nseq, nt, stream);      // ViT-g/14 (hidden 1408, 16 heads)
    case 96: return launch_nt<3, 6>(p, nseq, nt, stream);
    case 104: return launch_nt<4, 7, 104>(p, nseq, nt, stream);    // ViT-bigG/14 (hidden 1664, 16 heads)
    }
    return false;
}

}  // namespace clipamd
