// EXPERIMENT, NOT BUILT (round 3): built, verified bit-identical to launch_attention + the out-projection GEMM on 24
// test configurations, measured SLOWER and taken out of the library again — profiles/r03_lnfold_and_text_tiles.txt section 4:
//   ViT-B/32 vision, 256 sequences x T = 50:  91 us per layer  vs  18 (attention) + 36 (out-projection GEMM with the fold tail) = 54 us
//   text tower, 256 sequences x T <= 77:      73 us            vs  16 + 25 = 41 us          bench.py: 110.2 k -> 96.6 k embeddings/s
// To build it again: add the file to clip_cpp_amd/build.py HIP_SOURCES, declare attn_out_supported / launch_attn_out in kernels.h and
// call launch_attn_out from run_layers_fold in place of launch_attention + the "gemm_out" GEMM.
//
// k_attn_out.hip — attention + out-projection + residual of ONE sequence per workgroup, for short sequences (T <= 80, d_head = 64):
// the ViT-B/32 vision tower (T = 50) and every text (T <= 77) at large batch.
//
// Replaces, per layer, the attention launch AND the out-projection GEMM launch (reference clip.cpp:1382-1397, text :1100-1117):
//     KQ = mul_mat(K, Q); [causal mask]; soft_max; KQV = mul_mat(V^T, KQ); merge heads; x += W_o . ctx + b_o
// Why: at batch 256 the out-projection is the worst GEMM of the step (12800 x 768 x 768: 470 TFLOP/s — at K = 768
// a tile is 12 K-steps between a prologue and a 98 MB residual epilogue) and the attention launch writes 19.7 MB of context rows only
// for that GEMM to read them back.  Here a workgroup owns a whole sequence: it walks the heads, keeps the [T][h] fp32 out-projection
// accumulator in registers (h / 64 column fragments per wave x T / 16 row fragments: 192-240 registers of the 512 a wave has when it is
// alone on its SIMD), and per head
//     stages K, V^T of the head in LDS  ->  S^T = K Q^T, softmax in registers, O = P V (as k_attn.hip)  ->  O (fp16) to LDS
//     ->  acc += W_o[:, 64 head columns] . O^T   with the weight fragments dequantised straight from the block-column-major planes
//        (one 32-bit word of quants + scale per lane and fragment, as k_skinny.hip: the weights never touch LDS)
// then bias + residual, and — LayerNorm fold (gemm_common.h) — fp16(x gamma_next) and the 64-column statistics slots of the tiled
// kernels' residual epilogue (resid_fold_tail).  The context rows never exist in HBM.
// Numerics: O is rounded to fp16 where the attention kernel rounds it; the out-projection accumulates in k order (head by head, 32
// k per MFMA) like every tiled kernel: bit-identical to attention + GEMM launches (tested).
// One workgroup per CU (512 registers per lane), so the kernel wants a multiple of 256 sequences: launch_attn_out() says no otherwise.

#include "gemm_common.h"

namespace clipamd {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

struct AttnOutParams {
    const half_t * qkv;   // [rows][3h], Q pre-scaled
    const int * seq_start;
    int T_uniform;
    int h, n_head, causal;
    DevWeight W;          // out-projection [h][h]
    const float * bias;
    float * x;            // residual stream [rows][h]: read and written in place
    half_t * xg_out;      // LayerNorm fold, producer half (nullable): fp16(x_new gamma)
    const float * xg_gamma;
    float2 * stats_out;   // [h / 64 slots][stats_stride]: the 64-column slots of gemm_common.h
    int stats_stride;
};

// NT: 16-row fragments of a sequence (4: T <= 64, 5: T <= 80) = key tiles = query blocks;  CF: 16-column fragments of the
// out-projection per wave (h = 64 CF)
template <int WT, int NT, int CF>
__global__ void __launch_bounds__(256, 1) attn_out_kernel(const AttnOutParams p) {
    constexpr int DH = 64, DKS = 2, DT = 4;
    constexpr int KSTRIDE = DH + 8;                  // halfs per K row (+16 B: spreads the ds_read_b128 over the banks)
    constexpr int NPR = (NT + 1) / 2;                // key-tile pairs = K = 32 slices of P.V
    constexpr int VSTRIDE = NPR * 32 + 8;
    constexpr int OSTRIDE = DH + 8;                  // halfs per O row
    __shared__ __attribute__((aligned(16))) half_t Ks[NT * 16 * KSTRIDE];
    __shared__ __attribute__((aligned(16))) half_t Vt[DH * VSTRIDE];
    __shared__ __attribute__((aligned(16))) half_t Os[NT * 16 * OSTRIDE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fq = lane & 15, fg = lane >> 4;        // (= frow, fgrp of the GEMM kernels)
    const int seq = blockIdx.x;
    int row0, len;
    if (p.seq_start) {
        row0 = p.seq_start[seq];
        len = p.seq_start[seq + 1] - row0;
    } else {
        row0 = seq * p.T_uniform;
        len = p.T_uniform;
    }
    const int h = p.h, ld = 3 * h;
    const int nqb = (len + 15) >> 4;
    const int ncol0 = wave * CF * 16;                // this wave's first out-projection column

    f4 acc[CF][NT];
#pragma unroll
    for (int a = 0; a < CF; a++)
#pragma unroll
        for (int b = 0; b < NT; b++) acc[a][b] = (f4){0.f, 0.f, 0.f, 0.f};

    // K / V staging registers of one head (as k_attn.hip: every global load is issued before the first LDS store)
    constexpr int KCH = DH / 8;                      // 16-byte chunks per row
    constexpr int KIT = (NT * 16 * KCH + 255) / 256;
    constexpr int NPAIR = NPR * 16;
    constexpr int VIT = (NPAIR * KCH + 255) / 256;
    u32x4 kv[KIT], va[VIT], vb[VIT];
    auto load_kv = [&](int head) {
        const half_t * Kg = p.qkv + (size_t)row0 * ld + h + head * DH;
        const half_t * Vg = Kg + h;
#pragma unroll
        for (int i = 0; i < KIT; i++) {
            const int it = tid + i * 256;
            const int key = it / KCH, c = it % KCH;
            const int kc = key < len ? key : len - 1;
            kv[i] = *(const u32x4 *)(Kg + (size_t)kc * ld + c * 8);
        }
#pragma unroll
        for (int i = 0; i < VIT; i++) {
            const int it = tid + i * 256;
            const int kp = it % NPAIR, c = (it / NPAIR) < KCH ? (it / NPAIR) : KCH - 1;
            const int k0 = 2 * kp;
            va[i] = *(const u32x4 *)(Vg + (size_t)(k0 < len ? k0 : len - 1) * ld + c * 8);
            vb[i] = *(const u32x4 *)(Vg + (size_t)(k0 + 1 < len ? k0 + 1 : len - 1) * ld + c * 8);
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < KIT; i++) {
            const int it = tid + i * 256;
            const int key = it / KCH, c = it % KCH;
            if (it < NT * 16 * KCH) {
                const u32x4 v = key < len ? kv[i] : (u32x4){0u, 0u, 0u, 0u};
                *(u32x4 *)(Ks + key * KSTRIDE + c * 8) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < VIT; i++) {
            const int it = tid + i * 256;
            const int kp = it % NPAIR, c = it / NPAIR;
            const int k0 = 2 * kp;
            if (it < NPAIR * KCH) {
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
    };
    // out-projection weight fragments of one head: k-blocks 2 head, 2 head + 1; rows ncol0 + a 16 + fq
    struct WReg { WFrag<WT> q; h8 h; };
    auto load_w = [&](WReg & w, int kb, int a) {
        const int n = ncol0 + a * 16 + fq;
        if constexpr (WT == W_F16) {
            w.h = *(const h8 *)((const half_t *)p.W.w16 + (size_t)n * p.W.Kpad + kb * 32 + fg * 8);
        } else {
            const size_t idx = (size_t)kb * p.W.Npad + n;
            if constexpr (WT == W_Q8_0) {
                const uint2 q = ((const uint2 *)p.W.qs)[idx * 4 + fg];
                w.q.q = q.x;
                w.q.q1 = q.y;
            } else {
                w.q.q = ((const uint32_t *)p.W.qs)[idx * 4 + fg];
            }
            if constexpr (WT == W_Q5_0 || WT == W_Q5_1) w.q.h = ((const uint32_t *)p.W.qh)[idx];
            if constexpr (WT == W_Q4_1 || WT == W_Q5_1) w.q.dm = ((const h2 *)p.W.dm)[idx];
            else w.q.d = ((const half_t *)p.W.dm)[idx];
        }
    };

    load_kv(0);
    for (int head = 0; head < p.n_head; head++) {
        store_kv();
        // this head's weight fragments: requested now, consumed after the attention part
        WReg wr[2][CF];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int a = 0; a < CF; a++) load_w(wr[ks][a], head * 2 + ks, a);
        __syncthreads();                               // (1) K, V^T of this head visible; every wave is past its Os reads of the previous head
        if (head + 1 < p.n_head) load_kv(head + 1);    // next head's K / V rows in flight under this head's math

        // ---- attention of this wave's query blocks (k_attn.hip attn_blocks, QB = 1), O -> Os as fp16
        const half_t * Qg = p.qkv + (size_t)row0 * ld + head * DH;
        for (int qb = wave; qb < NT; qb += 4) {
            if (qb >= nqb) {                           // rows past the sequence: zeros (their accumulator rows are never stored)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int dt = 0; dt < DT; dt++) Os[(qb * 16 + fg * 4 + r) * OSTRIDE + dt * 16 + fq] = (_Float16)0.f;
                continue;
            }
            const int qrow = qb * 16 + fq;
            const int qclamped = qrow < len ? qrow : len - 1;
            h8 qf[DKS];
#pragma unroll
            for (int kk = 0; kk < DKS; kk++) qf[kk] = *(const h8 *)(Qg + (size_t)qclamped * ld + kk * 32 + fg * 8);
            f4 s[NT];
#pragma unroll
            for (int kt = 0; kt < NT; kt++) {
                s[kt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < DKS; kk++) {
                    const h8 kf = *(const h8 *)(Ks + (kt * 16 + fq) * KSTRIDE + (kk * 4 + fg) * 8);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], s[kt], 0, 0, 0);
                }
            }
            const int kmax = p.causal ? (qrow < len - 1 ? qrow : len - 1) : len - 1;   // last visible key
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = kt * 16 + fg * 4 + r;
                    s[kt][r] = key <= kmax ? s[kt][r] : -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float L2E = 1.44269504088896340736f;
            const float nmx = -mx * L2E;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], L2E, nmx));
                    s[kt][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            f4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; dt++) o[dt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < NPR; pr++) {
                const f4 p0 = s[2 * pr];
                f4 p1 = (f4){0.f, 0.f, 0.f, 0.f};
                if (2 * pr + 1 < NT) p1 = s[(2 * pr + 1 < NT) ? 2 * pr + 1 : 0];
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
            }
            // O layout: row (query) = fg*4 + r, col (d) = dt*16 + fq.  Normalise, round to fp16 (where k_attn.hip rounds), park in LDS.
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float invr = __shfl(inv, fg * 4 + r);
#pragma unroll
                for (int dt = 0; dt < DT; dt++) Os[(qb * 16 + fg * 4 + r) * OSTRIDE + dt * 16 + fq] = (_Float16)(o[dt][r] * invr);
            }
        }
        __syncthreads();                               // (2) O of every query block visible; every wave is past its K / V^T reads

        // ---- out-projection, this head's 64 k: acc[a][b] += W_o[ncol0 + 16 a .. +15][64 head + 32 ks ..] . O[16 b .. +15][32 ks ..]^T
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            h8 xf[NT];
#pragma unroll
            for (int b = 0; b < NT; b++) xf[b] = *(const h8 *)(Os + (b * 16 + fq) * OSTRIDE + ks * 32 + fg * 8);
#pragma unroll
            for (int a = 0; a < CF; a++) {
                h8 wf;
                if constexpr (WT == W_F16) wf = wr[ks][a].h; else wf = dequant_wfrag<WT>(wr[ks][a].q, fg);
#pragma unroll
                for (int b = 0; b < NT; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: x += acc + bias (rows < len).  Lane: row m = 16 b + fq, columns n = ncol0 + 16 a + 4 fg .. +3
#pragma unroll
    for (int a = 0; a < CF; a++) {
        const int n = ncol0 + a * 16 + fg * 4;
        const f4 bias = p.bias ? *(const f4 *)(p.bias + n) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NT; b++) {
            const int m = b * 16 + fq;
            const int mc = m < len ? m : len - 1;
            float * xr = p.x + (size_t)(row0 + mc) * h + n;
            const f4 r = *(const f4 *)xr;
            acc[a][b] = r + (acc[a][b] + bias);        // (same expression as the residual epilogue of the GEMM kernels)
            if (m < len) *(f4 *)xr = acc[a][b];
        }
    }
    if (p.xg_out) {
        // LayerNorm fold, producer half, exactly as the residual epilogue of the tiled kernels does it (gemm_common.h resid_fold_tail:
        // 64-column statistics slots built from the canonical 32-column units, fp16(x gamma)): what the consumer computes from them
        // does not depend on whether this kernel or the out-projection GEMM produced the rows
        GemmParams gp;
        gp.M = row0 + len;
        gp.W.N = h;
        gp.xg_out = p.xg_out; gp.ldxg = h; gp.xg_gamma = p.xg_gamma; gp.stats_out = p.stats_out; gp.stats_stride = p.stats_stride;
        resid_fold_tail<CF, NT>(gp, acc, ncol0, row0, fq, fg, nullptr, lane);
    }
}

template <int WT, int NT>
bool launch_cf(const AttnOutParams & p, int nseq, hipStream_t stream) {
    switch (p.h / 64) {
    case 8: hipLaunchKernelGGL((attn_out_kernel<WT, NT, 8>), dim3(nseq), dim3(256), 0, stream, p); return true;
    case 12: hipLaunchKernelGGL((attn_out_kernel<WT, NT, 12>), dim3(nseq), dim3(256), 0, stream, p); return true;
    }
    return false;
}

template <int WT>
bool launch_nt(const AttnOutParams & p, int nseq, int max_len, hipStream_t stream) {
    if (max_len <= 64) return launch_cf<WT, 4>(p, nseq, stream);
    if (max_len <= 80) return launch_cf<WT, 5>(p, nseq, stream);
    return false;
}

}  // namespace

// false: shape not covered (the caller then runs launch_attention + the out-projection GEMM)
bool attn_out_supported(int nseq, int max_len, int h, int n_head, const DevWeight & W) {
    if (n_head <= 0 || h % 64 || h / n_head != 64 || (h != 512 && h != 768) || max_len <= 0 || max_len > 80) return false;
    if (W.N != h || W.K != h || W.Kpad != h) return false;
    // one workgroup per CU and sequence: worth it when the sequences fill the chip's 256 CUs in (nearly) whole rounds
    const int rounds = (nseq + 255) / 256;
    return nseq >= 192 && (float)nseq / (float)(rounds * 256) >= 0.75f;
}

bool launch_attn_out(const half_t * qkv, int nseq, int T_uniform, const int * seq_start, int max_len, int h, int n_head, bool causal,
                     const DevWeight & W, const float * bias, float * x, half_t * xg_out, const float * xg_gamma, float2 * stats_out,
                     int stats_stride, hipStream_t stream) {
    AttnOutParams p;
    p.qkv = qkv; p.seq_start = seq_start; p.T_uniform = T_uniform; p.h = h; p.n_head = n_head; p.causal = causal ? 1 : 0;
    p.W = W; p.bias = bias; p.x = x; p.xg_out = xg_out; p.xg_gamma = xg_gamma; p.stats_out = stats_out; p.stats_stride = stats_stride;
    switch (W.wtype) {
    case W_F16: return launch_nt<W_F16>(p, nseq, max_len, stream);
    case W_Q4_0: return launch_nt<W_Q4_0>(p, nseq, max_len, stream);
    case W_Q4_1: return launch_nt<W_Q4_1>(p, nseq, max_len, stream);
    case W_Q5_0: return launch_nt<W_Q5_0>(p, nseq, max_len, stream);
    case W_Q5_1: return launch_nt<W_Q5_1>(p, nseq, max_len, stream);
    case W_Q8_0: return launch_nt<W_Q8_0>(p, nseq, max_len, stream);
    }
    return false;
}

}  // namespace clipamd
