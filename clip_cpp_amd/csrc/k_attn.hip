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
// once in LDS; each wave then owns 16-query blocks:
//   S^T = K · Q^T        v_mfma_f32_16x16x32_f16, A = K rows (keys), B = Q rows (queries)
//                        -> every lane holds, for ONE query (lane & 15), 4 keys per 16-key tile: the
//                        whole score row lives in the registers of the 4 lanes {q, q+16, q+32, q+48}
//   softmax              in registers; row max / row sum = local reduce + 2 wave shuffles (xor 16, 32)
//   O = P · V            the exp()'d scores, packed to fp16, ARE the MFMA A operand of the second
//                        contraction (keys of two adjacent 16-key tiles form one K=32 slice; V^T is read
//                        with the same key permutation), so P never touches LDS or HBM.
// Scores are never materialised in HBM (the reference materialises [T,T,n_head*B] f32).
// T <= 288 (all 224-px models and every text length); longer sequences are rejected by the launcher.

#include "kernels.h"

namespace clipamd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

struct AttnParams {
    const half_t * qkv;   // [rows][3h]
    half_t * out;         // [rows][h]
    const int * seq_start;
    int T_uniform;
    int h, n_head, dh;
    int causal;
    int kstride;          // halfs per K row in LDS   (DKP + 8)
    int vstride;          // halfs per V^T row in LDS (Tp32 + 8)
    int tp16, tp32;       // max_len rounded up to 16 / 32
};

// NT  = max number of 16-key tiles (compile-time so the score registers are statically indexed)
// DKS = number of 32-wide k-steps over the head dimension (dh padded to 32*DKS)
template <int NT, int DKS>
__global__ void __launch_bounds__(256) attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t * Ks = (half_t *)smem_raw;                       // [tp16][kstride]
    half_t * Vt = Ks + (size_t)p.tp16 * p.kstride;          // [dh][vstride]

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
    const int dh = p.dh, ld = 3 * p.h;
    const half_t * Qg = p.qkv + (size_t)row0 * ld + head * dh;
    const half_t * Kg = Qg + p.h;
    const half_t * Vg = Qg + 2 * p.h;
    const int nt = (len + 15) >> 4;          // key tiles in use
    const int dch = dh >> 3;                 // 16-byte chunks per head row
    constexpr int DKP = DKS * 32;

    // ---- stage K: Ks[key][0..DKP) (zero padded rows >= len and columns >= dh) ----
    {
        const int kch = DKP >> 3;
        const int total = nt * 16 * kch;
        for (int it = tid; it < total; it += 256) {
            const int key = it / kch, c = it % kch;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key < len && c < dch) v = *(const uint4 *)(Kg + (size_t)key * ld + c * 8);
            *(uint4 *)(Ks + key * p.kstride + c * 8) = v;
        }
    }
    // ---- stage V transposed: Vt[d][key], keys in pairs so every LDS store is a full dword ----
    {
        const int npair = ((nt + 1) >> 1) * 16;   // covers ceil(nt/2)*32 keys, zero padded
        const int total = npair * dch;
        for (int it = tid; it < total; it += 256) {
            const int kp = it % npair, c = it / npair;
            const int k0 = 2 * kp;
            uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
            if (k0 < len) a = *(const uint4 *)(Vg + (size_t)k0 * ld + c * 8);
            if (k0 + 1 < len) b = *(const uint4 *)(Vg + (size_t)(k0 + 1) * ld + c * 8);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t av = (aw[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                const uint32_t bv = (bw[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                *(uint32_t *)(Vt + (size_t)(c * 8 + e) * p.vstride + k0) = av | (bv << 16);
            }
        }
    }
    __syncthreads();

    const int fq = lane & 15, fg = lane >> 4;
    const int nqb = (len + 15) >> 4;
    const int ndt = dh >> 4;  // output d tiles

    for (int qb = wave; qb < nqb; qb += 4) {
        // Q fragment (MFMA B operand): query row qb*16+fq, d = kk*32 + fg*8 .. +7
        int qrow = qb * 16 + fq;
        const int qclamped = qrow < len ? qrow : len - 1;
        h8 qf[DKS];
#pragma unroll
        for (int kk = 0; kk < DKS; kk++) {
            const int d0 = kk * 32 + fg * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (d0 < dh) v = *(const uint4 *)(Qg + (size_t)qclamped * ld + d0);
            qf[kk] = __builtin_bit_cast(h8, v);
        }
        // ---- S^T tiles ----
        f4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            s[kt] = (f4){0.f, 0.f, 0.f, 0.f};
            if (kt < nt) {
#pragma unroll
                for (int kk = 0; kk < DKS; kk++) {
                    const h8 kf = *(const h8 *)(Ks + (kt * 16 + fq) * p.kstride + kk * 32 + fg * 8);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], s[kt], 0, 0, 0);
                }
            }
        }
        // ---- mask + row max.  lane holds query fq, keys kt*16 + fg*4 + r ----
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = kt * 16 + fg * 4 + r;
                const bool valid = (kt < nt) && key < len && (!p.causal || key <= qrow);
                s[kt][r] = valid ? s[kt][r] : -INFINITY;
                mx = fmaxf(mx, s[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (mx == -INFINITY) mx = 0.f;  // fully masked (padding query): keep exp() finite
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = __expf(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;

        // ---- O = P V : pairs of key tiles form one K=32 slice ----
        f4 o[6];
#pragma unroll
        for (int dt = 0; dt < 6; dt++) o[dt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < (NT + 1) / 2; pr++) {
            if (2 * pr < nt) {
                const f4 p0 = s[2 * pr];
                const f4 p1 = (2 * pr + 1 < NT) ? s[(2 * pr + 1 < NT) ? 2 * pr + 1 : 0] : (f4){0.f, 0.f, 0.f, 0.f};
                h8 pf;
                pf[0] = (_Float16)p0[0]; pf[1] = (_Float16)p0[1]; pf[2] = (_Float16)p0[2]; pf[3] = (_Float16)p0[3];
                pf[4] = (_Float16)p1[0]; pf[5] = (_Float16)p1[1]; pf[6] = (_Float16)p1[2]; pf[7] = (_Float16)p1[3];
#pragma unroll
                for (int dt = 0; dt < 6; dt++) {
                    if (dt < ndt) {
                        const half_t * vrow = Vt + (size_t)(dt * 16 + fq) * p.vstride + pr * 32 + fg * 4;
                        const h4 v0 = *(const h4 *)(vrow);
                        const h4 v1 = *(const h4 *)(vrow + 16);
                        h8 vf;
                        vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                        vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[dt], 0, 0, 0);
                    }
                }
            }
        }
        // ---- normalise rows and store.  O layout: row (query) = fg*4 + r, col (d) = fq ----
        float invr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) invr[r] = __shfl(inv, fg * 4 + r);
#pragma unroll
        for (int dt = 0; dt < 6; dt++) {
            if (dt < ndt) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int q = qb * 16 + fg * 4 + r;
                    if (q < len)
                        p.out[(size_t)(row0 + q) * p.h + head * dh + dt * 16 + fq] = (_Float16)(o[dt][r] * invr[r]);
                }
            }
        }
    }
}

template <int NT>
bool launch_nt(const AttnParams & p, int nseq, size_t smem, int dks, hipStream_t stream) {
    dim3 grid(nseq * p.n_head), block(256);
    switch (dks) {
    case 1:
        hipFuncSetAttribute((const void *)attn_kernel<NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((attn_kernel<NT, 1>), grid, block, smem, stream, p);
        return true;
    case 2:
        hipFuncSetAttribute((const void *)attn_kernel<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((attn_kernel<NT, 2>), grid, block, smem, stream, p);
        return true;
    case 3:
        hipFuncSetAttribute((const void *)attn_kernel<NT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((attn_kernel<NT, 3>), grid, block, smem, stream, p);
        return true;
    }
    return false;
}

}  // namespace

bool launch_attention(const half_t * qkv, half_t * out, int nseq, int T_uniform, const int * seq_start, int max_len,
                      int h, int n_head, bool causal, hipStream_t stream) {
    if (nseq <= 0) return true;
    const int dh = h / n_head;
    if (dh % 16 != 0 || dh > 96 || max_len > 288 || max_len <= 0) return false;
    AttnParams p;
    p.qkv = qkv;
    p.out = out;
    p.seq_start = seq_start;
    p.T_uniform = T_uniform;
    p.h = h;
    p.n_head = n_head;
    p.dh = dh;
    p.causal = causal ? 1 : 0;
    const int dks = (dh + 31) / 32;
    p.tp16 = (max_len + 15) / 16 * 16;
    p.tp32 = (max_len + 31) / 32 * 32;
    p.kstride = dks * 32 + 8;
    p.vstride = p.tp32 + 8;
    const size_t smem = ((size_t)p.tp16 * p.kstride + (size_t)dh * p.vstride) * sizeof(half_t);
    const int nt = p.tp16 / 16;
    if (nt <= 4) return launch_nt<4>(p, nseq, smem, dks, stream);
    if (nt <= 8) return launch_nt<8>(p, nseq, smem, dks, stream);
    return launch_nt<18>(p, nseq, smem, dks, stream);
}

}  // namespace clipamd
