// k_attn.hip — fused multi-head attention softmax(Q K^T) V on fp16 MFMA for short sequences.
//
// Replaces, per (sequence, head), the ggml sub-graph of reference clip.cpp:1382-1388 (vision) and
// :1100-1108 (text, with ggml_diag_mask_inf causal mask :1101):
//     KQ = mul_mat(K, Q); [mask]; soft_max(KQ); KQV = mul_mat(V^T, KQ); permute/cont/cpy
// including the head split/merge permutes (the kernel reads Q/K/V straight out of the fused
// [rows][3h] projection output and writes the merged [rows][h] context).  Q arrives pre-scaled by
// 1/sqrt(d_head) (GEMM epilogue; the reference scales Q after the bias, clip.cpp:1363).
//
// (Device code in attn_body.h: also the second phase of the fused small-M kernel, k_qkv_attn.hip.)
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

#include "attn_body.h"

namespace clipamd {

namespace {

// NT  = number of 16-key tiles (>= ceil(max_len/16)); DKS = 32-wide k-steps over the head dim (dh <= 32*DKS);
// DT = dh/16 output tiles.
// occupancy: short sequences (T <= 80: ViT-B/32 images, every text) are latency-bound — stage, one barrier, a handful of MFMAs — so 4
// workgroups per CU (<= 128 VGPRs) instead of 2 hide twice the memory latency; long ones need the registers
template <int NT, int DKS, int DT>
__global__ void __launch_bounds__(256, (NT > 18 ? 1 : NT <= 5 ? 4 : 2)) attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    attn_body<NT, DKS, DT, 256>(p, blockIdx.x / p.n_head, blockIdx.x % p.n_head, blockIdx.y, gridDim.y, smem_raw);
}

template <int NT, int DKS, int DT>
void launch_inst(const AttnParams & p, int nseq, hipStream_t stream) {
    constexpr size_t smem = ((size_t)NT * 16 * (DKS * 32 + (NT > 18 ? 0 : 8)) + (size_t)DT * 16 * (((NT + 1) / 2) * 32 + 8)) * sizeof(half_t);
    static_assert(smem <= 160 * 1024, "attention tile does not fit the LDS");
    static unsigned long long lds_ok = 0;
    if (smem > 64 * 1024) opt_in_dynamic_lds(attn_kernel<NT, DKS, DT>, smem, lds_ok);
    // query split: only where (sequence, head) pairs alone leave most CUs idle and a pair has more work units than one workgroup's 4 waves
    constexpr int QB = (NT >= 7 && NT <= 18) ? 2 : 1;
    const int units = (NT + QB - 1) / QB;
    int qs = 1;
    if (nseq * p.n_head <= 64 && units > 4) qs = (units + 3) / 4 < 4 ? (units + 3) / 4 : 4;
    hipLaunchKernelGGL((attn_kernel<NT, DKS, DT>), dim3(nseq * p.n_head, qs), dim3(256), smem, stream, p);
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
