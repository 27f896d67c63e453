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
// T <= 288 for every d_head in {32,64,80,96} (all 224-px models and every text length); T <= 592 for d_head <= 64
// (ViT-L/14 at 336 px: T = 577, 154.6 KB of LDS); longer sequences are rejected by the launcher.

#include "kernels.h"

namespace clipamd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct AttnParams {
    const half_t * qkv;   // [rows][3h]
    half_t * out;         // [rows][h]
    const int * seq_start;
    int T_uniform;
    int h, n_head;
    int causal;
};

// NT  = number of 16-key tiles (>= ceil(max_len/16)); DKS = 32-wide k-steps over the head dim (dh <= 32*DKS);
// DT = dh/16 output tiles.
template <int NT, int DKS, int DT>
__global__ void __launch_bounds__(256, (NT > 18 ? 1 : 2)) attn_kernel(const AttnParams p) {
    constexpr int DKP = DKS * 32;
    // Long sequences (NT > 18: 336-px models, T = 577) only fit the 160 KB LDS without the row padding: K rows are then
    // exactly 128 B with the 16-byte chunks XOR-swizzled by (key & 7) instead (same conflict-free ds_read_b128 pattern
    // as the GEMM tiles); needs DKP == 64.
    constexpr bool SWZ = NT > 18;
    static_assert(!SWZ || DKP == 64, "swizzled K layout is for d_head <= 64");
    constexpr int KSTRIDE = SWZ ? DKP : DKP + 8;     // halfs per K row  (+16 B pad: spreads ds_read_b128 over banks)
    constexpr int NPR = (NT + 1) / 2;                // key-tile pairs = K=32 slices of the P.V contraction
    constexpr int VSTRIDE = NPR * 32 + 8;            // halfs per V^T row; (VSTRIDE/2) = 4*odd -> conflict-free ds_read_b64
    constexpr int DH = DT * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t * Ks = (half_t *)smem_raw;                // [NT*16][KSTRIDE]
    half_t * Vt = Ks + NT * 16 * KSTRIDE;            // [DH][VSTRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // ---- stage K: Ks[key][0..DKP), zero for key >= len and for columns >= DH ----
    {
        constexpr int KCH = DKP / 8;
        for (int it = tid; it < NT * 16 * KCH; it += 256) {
            const int key = it / KCH, c = it % KCH;
            u32x4 v = (u32x4){0u, 0u, 0u, 0u};
            if (key < len && c < DCH) v = *(const u32x4 *)(Kg + (size_t)key * ld + c * 8);
            *(u32x4 *)(Ks + key * KSTRIDE + (SWZ ? (c ^ (key & 7)) : c) * 8) = v;
        }
    }
    // ---- stage V transposed: Vt[d][key]; two keys per thread so every LDS store is a full dword ----
    {
        constexpr int NPAIR = NPR * 16;              // key pairs (covers NPR*32 keys, zero padded)
        for (int it = tid; it < NPAIR * DCH; it += 256) {
            const int kp = it % NPAIR, c = it / NPAIR;
            const int k0 = 2 * kp;
            u32x4 a = (u32x4){0u, 0u, 0u, 0u}, b = (u32x4){0u, 0u, 0u, 0u};
            if (k0 < len) a = *(const u32x4 *)(Vg + (size_t)k0 * ld + c * 8);
            if (k0 + 1 < len) b = *(const u32x4 *)(Vg + (size_t)(k0 + 1) * ld + c * 8);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t av = (a[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                const uint32_t bv = (b[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                *(uint32_t *)(Vt + (c * 8 + e) * VSTRIDE + k0) = av | (bv << 16);
            }
        }
    }
    __syncthreads();

    const int fq = lane & 15, fg = lane >> 4;
    const int nqb = (len + 15) >> 4;

    for (int qb = wave; qb < nqb; qb += 4) {
        // Q fragment (MFMA B operand): query row qb*16+fq, d = kk*32 + fg*8 .. +7
        const int qrow = qb * 16 + fq;
        const int qclamped = qrow < len ? qrow : len - 1;
        h8 qf[DKS];
#pragma unroll
        for (int kk = 0; kk < DKS; kk++) {
            const int d0 = kk * 32 + fg * 8;
            u32x4 v = (u32x4){0u, 0u, 0u, 0u};
            if (d0 < DH) v = *(const u32x4 *)(Qg + (size_t)qclamped * ld + d0);
            qf[kk] = __builtin_bit_cast(h8, v);
        }
        // ---- S^T tiles (unconditional MFMA chain) ----
        f4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            s[kt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DKS; kk++) {
                const int kch = kk * 4 + fg;
                const h8 kf = *(const h8 *)(Ks + (kt * 16 + fq) * KSTRIDE + (SWZ ? (kch ^ (fq & 7)) : kch) * 8);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], s[kt], 0, 0, 0);
            }
            // keep at most 4 key tiles (8 fragment reads) in flight: fully hoisted, the reads of all NT tiles cost
            // 4*2*NT VGPRs (NT = 18: 488 registers, one workgroup per CU)
            if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- mask + row max.  lane holds query fq, keys kt*16 + fg*4 + r ----
        const int kmax = p.causal ? (qrow < len - 1 ? qrow : len - 1) : len - 1;   // last visible key
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = kt * 16 + fg * 4 + r;
                s[kt][r] = key <= kmax ? s[kt][r] : -INFINITY;
                mx = fmaxf(mx, s[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = __expf(s[kt][r] - mx);   // key 0 is always visible -> mx is finite
                s[kt][r] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        // ---- O = P V : pairs of key tiles form one K=32 slice ----
        f4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; dt++) o[dt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < NPR; pr++) {
            const f4 p0 = s[2 * pr];
            f4 p1 = (f4){0.f, 0.f, 0.f, 0.f};
            if constexpr (true) {
                if (2 * pr + 1 < NT) p1 = s[(2 * pr + 1 < NT) ? 2 * pr + 1 : 0];
            }
            h8 pf;
            pf[0] = (_Float16)p0[0]; pf[1] = (_Float16)p0[1]; pf[2] = (_Float16)p0[2]; pf[3] = (_Float16)p0[3];
            pf[4] = (_Float16)p1[0]; pf[5] = (_Float16)p1[1]; pf[6] = (_Float16)p1[2]; pf[7] = (_Float16)p1[3];
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const half_t * vrow = Vt + (dt * 16 + fq) * VSTRIDE + pr * 32 + fg * 4;
                const h4 v0 = *(const h4 *)(vrow);
                const h4 v1 = *(const h4 *)(vrow + 16);
                h8 vf;
                vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[dt], 0, 0, 0);
            }
            if ((pr & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- normalise rows and store.  O layout: row (query) = fg*4 + r, col (d) = fq ----
        float invr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) invr[r] = __shfl(inv, fg * 4 + r);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = qb * 16 + fg * 4 + r;
            if (q < len) {
                half_t * orow = p.out + (size_t)(row0 + q) * p.h + head * DH + fq;
#pragma unroll
                for (int dt = 0; dt < DT; dt++) orow[dt * 16] = (_Float16)(o[dt][r] * invr[r]);
            }
        }
    }
}

template <int NT, int DKS, int DT>
void launch_inst(const AttnParams & p, int nseq, hipStream_t stream) {
    constexpr size_t smem = ((size_t)NT * 16 * (DKS * 32 + (NT > 18 ? 0 : 8)) + (size_t)DT * 16 * (((NT + 1) / 2) * 32 + 8)) * sizeof(half_t);
    static_assert(smem <= 160 * 1024, "attention tile does not fit the LDS");
    static unsigned long long lds_ok = 0;
    if (smem > 64 * 1024) opt_in_dynamic_lds(attn_kernel<NT, DKS, DT>, smem, lds_ok);
    hipLaunchKernelGGL((attn_kernel<NT, DKS, DT>), dim3(nseq * p.n_head), dim3(256), smem, stream, p);
}

template <int DKS, int DT>
bool launch_nt(const AttnParams & p, int nseq, int nt, hipStream_t stream) {
    if (nt <= 1) launch_inst<1, DKS, DT>(p, nseq, stream);
    else if (nt <= 2) launch_inst<2, DKS, DT>(p, nseq, stream);
    else if (nt <= 3) launch_inst<3, DKS, DT>(p, nseq, stream);
    else if (nt <= 4) launch_inst<4, DKS, DT>(p, nseq, stream);
    else if (nt <= 5) launch_inst<5, DKS, DT>(p, nseq, stream);
    else if (nt <= 7) launch_inst<7, DKS, DT>(p, nseq, stream);
    else if (nt <= 10) launch_inst<10, DKS, DT>(p, nseq, stream);
    else if (nt <= 14) launch_inst<14, DKS, DT>(p, nseq, stream);
    else if (nt <= 17) launch_inst<17, DKS, DT>(p, nseq, stream);
    else if (nt <= 18) launch_inst<18, DKS, DT>(p, nseq, stream);
    else if constexpr (DKS == 2) {
        if (nt <= 37) launch_inst<37, DKS, DT>(p, nseq, stream);   // 336-px ViT-L/14: T = 577
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
    const int nt = (max_len + 15) / 16;
    switch (dh) {
    case 32: return launch_nt<1, 2>(p, nseq, nt, stream);
    case 64: return launch_nt<2, 4>(p, nseq, nt, stream);
    case 80: return launch_nt<3, 5>(p, nseq, nt, stream);
    case 96: return launch_nt<3, 6>(p, nseq, nt, stream);
    }
    return false;
}

}  // namespace clipamd
