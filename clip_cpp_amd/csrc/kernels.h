// kernels.h — launch interface of the gfx950 HIP kernels (device side of the hot path).
//
// Hot path = reference clip.cpp:1247-1523 (vision) and :1016-1233 (text); the ggml ops those
// graph builders invoke are replaced by the kernels declared here.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace clipamd {

typedef _Float16 half_t;

// Kernels with more than 64 KB of dynamic LDS need hipFuncSetAttribute once PER DEVICE (a process may hold contexts on
// several GPUs): `done` is a per-kernel static bitmask of device ordinals already configured.
template <typename K>
inline void opt_in_dynamic_lds(K kernel, size_t bytes, unsigned long long & done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return;
    const hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        fprintf(stderr, "clip (hip): hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed: %s\n", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return;
    }
    done |= bit;
}

// Device weight formats (repacked from the ggml block layout at load time, see model.cpp).
enum WType : int { W_F16 = 0, W_Q4_0 = 1, W_Q4_1 = 2, W_Q5_0 = 3, W_Q5_1 = 4, W_Q8_0 = 5, W_F32 = 6 };

// One linear weight W[N][K] (y = W x), resident in HBM in the GEMM-friendly layout:
//   W_F16 : w16[Npad][Kpad] row-major fp16 (zero padded)
//   W_F32 : the same matrix in f32 behind the same pointer (f32 GGUF files: multiplied on the exact-f32 MFMA, k_gemm_f32.hip)
//   quant : "block-column-major" planes over nkb = Kpad/32 blocks of 32 weights:
//           qs[kb][n]  16 B (q8_0: 32 B) of packed quants, nibble/byte order permuted for cheap unpack
//           qh[kb][n]  4 B of fifth bits (q5_*), permuted
//           dm[kb][n]  fp16 scale d (q4_0,q5_0,q8_0) or half2 {d, m} (q4_1,q5_1)
//   so that the 128 rows of a tile at one kb are contiguous (coalesced 16 B/lane loads).
struct DevWeight {
    int wtype = W_F16;
    int N = 0, K = 0;      // logical shape
    int Npad = 0;          // multiple of 128 (padded rows are zero)
    int Kpad = 0;          // multiple of 64  (padded blocks are zero)
    const void * qs = nullptr;
    const void * qh = nullptr;
    const void * dm = nullptr;
    const void * w16 = nullptr;
};

enum Epilogue : int {
    EPI_F32 = 0,        // out f32 = acc (+bias)
    EPI_F16 = 1,        // out f16 = (acc + bias) * (n < qcols ? qscale : 1)
    EPI_GELU_F16 = 2,   // out f16 = gelu_tanh(acc + bias)
    EPI_QGELU_F16 = 3,  // out f16 = quick_gelu(acc + bias)
    EPI_RESID_F32 = 4,  // out f32 = resid + acc + bias      (resid may alias out)
    EPI_PATCH_F32 = 5,  // out f32 row (m/Np)*T+1+m%Np = acc + pos[1+m%Np]   (patch embedding)
    EPI_COUNT = 6
};

struct GemmParams {
    const half_t * A = nullptr;  // activations [M][lda] fp16, lda >= Kpad
    int lda = 0;
    int M = 0;
    DevWeight W;
    const float * bias = nullptr;
    void * out = nullptr;
    int ldc = 0;
    const float * resid = nullptr;
    float qscale = 1.0f;
    int qcols = 0;
    int Np = 0, T = 0;
    const float * pos = nullptr;
    // split-K (small-M problems): workspace for partial tiles + per-tile ticket counters (zeroed once; the kernel leaves
    // them zero).  ksplit is chosen by launch_gemm; 1 = off.  Null workspace = never split.
    float * sk_ws = nullptr;
    size_t sk_ws_floats = 0;
    unsigned * sk_cnt = nullptr;
    int sk_cnt_n = 0;
    int ksplit = 1;
    bool no_splitk = false;   // never split K (the patch stage of the host pipeline runs per 32-image piece: its rows must carry the bits of the one-launch form)
    // large-M path (k_gemm8.hip): fp16 panel of a block-quantised W ([Npad][Kpad] row-major).  w16_pre != null: the panel was
    // already filled for this launch (per-layer dequantisation, forward.cpp); else launch_gemm fills w16_scratch
    // (>= Npad * Kpad halfs) itself when it picks that path; with neither, quantised weights stay on the fused 4-wave kernel.
    const half_t * w16_pre = nullptr;
    half_t * w16_scratch = nullptr;
    size_t w16_scratch_halfs = 0;
    int debug = 0;   // ablation switches, honoured only by -DCLIPAMD_ABLATION tuning builds (scripts/build_variant.sh): 1 skip tile loads, 2 skip MFMAs, 4 skip W dequant-store
    // ---- LayerNorm folded into the GEMMs around it (reference clip.cpp:1350-1355,1400-1405; text :1071-1076,1121-1126) ----
    // y = W LN(x) + b with LN(x) = (x - mean) rstd gamma + beta  ==  rstd (W (x gamma) - mean c) + b',  c_n = sum_k gamma_k W_nk,
    // b'_n = sum_k beta_k W_nk + b_n (both precomputed at load from the weights as the kernels dequantise them: k_fold.hip).
    // Consumer (EPI_F16 / EPI_GELU_F16 / EPI_QGELU_F16 with ln_c != null): A holds fp16(x gamma), bias holds b', and the epilogue
    // applies rstd_m (acc - mean_m c_n) + b'_n before the Q scale / activation; (mean, rstd) of row m come from ln_stats.
    const float * ln_c = nullptr;          // [N]
    const float2 * ln_stats = nullptr;     // [ln_slots][ln_stride]: (sum, sum of squared deviations from the slot mean) of row m over
    int ln_slots = 0, ln_slotw = 0;        //   columns [slot * ln_slotw, (slot + 1) * ln_slotw) of the f32 residual stream
    int ln_stride = 0;
    float ln_eps = 0.f;
    // Producer (EPI_RESID_F32 with xg_out != null): besides out = resid + acc + bias the epilogue writes the next GEMM's operand
    // xg_out[m][n] = fp16(out[m][n] * xg_gamma[n]) and the partial row statistics stats_out[n / w][m], w = gemm_fold_slotw(tile) columns per slot.
    half_t * xg_out = nullptr;
    int ldxg = 0;
    const float * xg_gamma = nullptr;
    float2 * stats_out = nullptr;
    int stats_stride = 0;
    // Centring of the folded operand (round 4; the reference normalises FIRST, in f32 with double sums: clip.cpp:1350-1355).  fp16(x gamma)
    // carries the row's common mode: for a row with |mean| = k sigma its rounding error is ~k times that of fp16(LN(x)).  So the operand
    // is built about a per-row offset mu_m the producer already knows — the row's mean at the PREVIOUS LayerNorm, which the previous
    // consumer computed anyway: xg = fp16((x - mu_m) gamma), and the consumer applies rstd (acc - (mean - mu_m) c) + b'.  Exact algebra
    // for any mu; with mu = mean it IS the unfolded operand up to the 1/sigma scale.  All three null = the uncentred r03 form.
    const float * xg_mu = nullptr;         // producer: [M] offsets the new operand is centred on
    const float * ln_mu = nullptr;         // consumer: [M] offsets A was centred on
    float * mu_out = nullptr;              // consumer: [M], the workgroups of the first column tile leave this LayerNorm's row means here
    // the caller runs other work on this device at the same time (the other tower of a two-tower step on a second stream): the heuristic then
    // keeps to the kernels that share a CU (two workgroups of <= 80 KB LDS) — see pick_tile in k_gemm.hip, round 6
    bool shared_device = false;
    // f32 GGUF files (W_F32, k_gemm_f32.hip only; round 6): the activations between the kernels are f32 as in the reference (ggml f32 x f32): A is
    // float [M][lda] behind the half_t pointer, and the fp16-output epilogues (EPI_F16 / EPI_GELU_F16 / EPI_QGELU_F16) store float [M][ldc]
    bool act_f32 = false;
};

// tile: 0 = heuristic, else [ksplit*1000000 +] BM*1000 + BN  (BM in {64,128,160,192}, BN in {64,128}; ksplit only with BM = 64 / 65;
// BM = 65: the ring kernel of k_gemm_ring.hip on 64-row tiles, BN in {64,128};
// BN = 256 with BM in {96,128,160}: the 8-wave large-M kernel of k_gemm8.hip)
void launch_gemm(const GemmParams & p, int epilogue, int tile, hipStream_t stream);
int gemm_tile_for(int M, int N, int Kpad, bool quantised, bool shared_device = false);   // the tile (BM*1000+BN) the heuristic picks for this shape
// the shape runs on a large-M kernel that multiplies an fp16 W panel (k_gemm8.hip / k_gemm4.hip).  BN codes: 256 plain 8-wave tile
// (BM 96 / 128 / 160 / 256), 258 the 8-wave 256 x 256 tile on the rows that fill whole rounds of 256 workgroups + a second launch for the
// rest, 259 the 4-wave 256 x 256 tile (k_gemm4.hip), 260 = 259 with the whole-rounds split
inline bool gemm_tile_is_ring(int tile) { return (tile % 1000000) / 1000 == 65; }   // k_gemm_ring.hip (BM code 65)
// 261: the 32 x 32 x 16 kernel of k_gemm32.hip (fp16-output epilogues): 256261 / 320261 = 256 / 320 x 256 tiles, one workgroup per CU
inline bool gemm_tile_uses_panel(int tile) { const int bn = tile % 1000; return !gemm_tile_is_ring(tile) && (bn == 256 || (bn >= 258 && bn <= 261)); }

// k_gemm_ring.hip: mid-M GEMM (64 activation rows x bn weight rows per workgroup, bn in {64, 128}; a 3-4 stage LDS ring of K-tiles
// filled by LDS-DMA only, block-quantised weights staged raw and dequantised per MFMA fragment; split-K as k_gemm.hip: p.ksplit).
// Tile codes 65064 / 65128 of launch_gemm.
void launch_gemm_ring(const GemmParams & p, int epilogue, int bn, hipStream_t stream);

// k_gemm8.hip: 8-wave ping-pong GEMM on (32 tm) x 256 tiles, fp16 x fp16 (p.W.w16 = [Npad][Kpad] panel), tm in {3,4,5}
void launch_gemm8(const GemmParams & p, int epilogue, int tm, hipStream_t stream);
// k_gemm_f32.hip: every weight GEMM of an f32 file (W_F32: f32 weights x fp16 activations widened exactly, f32 MFMA); launch_gemm routes there
void launch_gemm_f32(const GemmParams & p, int epilogue, hipStream_t stream);
// k_gemm4.hip: 4 waves x (128 x 128) on 256 x 256 tiles, accumulators in AGPRs, fp16 x fp16 (p.W.w16 = [Npad][Kpad] panel)
void launch_gemm4(const GemmParams & p, int epilogue, hipStream_t stream);
// k_gemm32.hip: 4 waves x ((32 wm) x 128) of v_mfma_f32_32x32x16_f16 on (64 wm) x 256 tiles, wm in {4, 5}; fp16-output epilogues on an fp16 panel
bool gemm32_supported(const GemmParams & p, int epilogue);
void launch_gemm32(const GemmParams & p, int epilogue, int form, hipStream_t stream);   // form: 4 = 256 x 256, 5 = 320 x 256
// dequantise n block-quantised weights into fp16 [Npad][Kpad] panels (one launch per run of equal weight type, <= 4 weights each)
struct DequantJobs { DevWeight W[4]; half_t * out[4]; int blk_end[4]; int n = 0; };
void launch_dequant(const DevWeight * const * ws, half_t * const * outs, int n, hipStream_t stream);

// LayerNorm fold (gemm_common.h): columns per statistics slot that the residual epilogue of the kernel behind `tile` writes (64, or
// 32 for the ring kernel and the BN = 64 tiles whose waves span 32 columns), and the same for the tile launch_gemm's heuristic picks
// for a shape in fold mode (the rows past the whole rounds of a split launch are then kept on a 64-column kernel).
int gemm_fold_slotw(int tile);
int gemm_fold_slotw_for(int M, int N, int Kpad, bool quantised, bool shared_device = false);
// k_fold.hip (once per model load): c[n] = sum_k gamma[k] W[n][k], b_out[n] = sum_k beta[k] W[n][k] + bias[n], W as the GEMMs dequantise it
void launch_fold_vectors(const DevWeight & W, const float * gamma, const float * beta, const float * bias, float * c_out, float * b_out,
                         hipStream_t stream);

// k_skinny.hip: latency-oriented weight GEMM for small M (one image, one text; the layers use it up to 64 rows): N / 16 x ceil(M / 16)
// workgroups, weights dequantised in registers straight from the planes, intra-workgroup split-K, optional LayerNorm fused on the A operand (A = LN(x32) with row
// statistics taken from the producer's partial slots) and partial statistics of the new residual rows out of EPI_RESID_F32.
struct SkinnyParams {
    const half_t * A16 = nullptr;   // fp16 activations [M][lda] ...
    int lda = 0;
    const float * x32 = nullptr;    // ... or the f32 residual stream [M][ldx], normalised on the fly with ln_w / ln_b / eps
    int ldx = 0;
    const float * ln_w = nullptr, * ln_b = nullptr;
    float eps = 0.f;
    const float2 * stats_in = nullptr;   // [row][stats_cap] partial (sum, sum of squares) of x32 rows: the first stats_slots entries of a row are valid
    int stats_slots = 0;
    int M = 0;
    DevWeight W;
    const float * bias = nullptr;
    void * out = nullptr;
    int ldc = 0;
    const float * resid = nullptr;
    float qscale = 1.0f;
    int qcols = 0;
    int Np = 0, T = 0;
    const float * pos = nullptr;
    float2 * stats_out = nullptr;        // EPI_RESID_F32: [row][stats_cap], entry blockIdx.x = statistics of the row over this workgroup's 16 columns
    int stats_cap = 128;
    // LayerNorm fold on the small-M path (same algebra as GemmParams::ln_* / xg_*; gemm_common.h).  Consumer (fp16 epilogues with
    // ln_c != null): A16 holds fp16(x gamma), bias holds b', the epilogue applies rstd (acc - mean c) + b'; the row statistics are
    // reduced in the kernel's prologue from fstats [fslots][fstride] (slots of fslotw columns: 16 out of the skinny residual epilogue,
    // or ONE slot of width h out of the entry kernel).  Producer (EPI_RESID_F32 with xg_out != null): also writes xg_out = fp16(out gamma_next)
    // and fstats_out[blockIdx.x][row] = (sum, sum of squared deviations) of the row over this workgroup's 16 columns.
    const float * ln_c = nullptr;
    const float2 * fstats = nullptr;
    int fslots = 0, fslotw = 0, fstride = 0;
    half_t * xg_out = nullptr;
    int ldxg = 0;
    const float * xg_gamma = nullptr;
    float2 * fstats_out = nullptr;
    int fstride_out = 0;
    const float * xg_mu = nullptr, * ln_mu = nullptr;      // centring of the folded operand: as GemmParams::xg_mu / ln_mu / mu_out
    float * mu_out = nullptr;
    unsigned long long * stamps = nullptr;   // -DCLIPAMD_SK_TIMING builds (scripts/build_sk_timing.sh): 16 phase stamps of the first and the last workgroup
};
constexpr int SKINNY_MAX_ROWS = 512;    // rows the statistics buffers are sized for ([2][SKINNY_MAX_ROWS][stats_cap] float2)
bool skinny_supported(const SkinnyParams & p, int epilogue);
void launch_skinny(const SkinnyParams & p, int epilogue, hipStream_t stream);
void launch_row_stats(const float * x, int ldx, int rows, int h, float2 * stats, hipStream_t stream);   // slot 0 := statistics of every row

// LayerNorm over rows of h floats (ggml_norm + mul + add, reference clip.cpp:1350-1355).
// Row r reads x[in_rows[r]] when in_rows != nullptr, else x[r * in_row_mul] (strided gather, e.g. the
// CLS rows b*T); out16/out32 may be null.
void launch_layernorm(const float * x, int ldx, const int * in_rows, int in_row_mul, const float * w, const float * b, float eps,
                      int rows, int h, half_t * out16, int ld16, float * out32, int ld32, hipStream_t stream);
// Entry of the LayerNorm-folded layer chain (gemm_common.h): what the first layer's q/k/v GEMM needs from the rows of the residual
// stream y — xg[r] = fp16(y[r] * gamma_next) and the whole-row statistics stats[r] = (sum, sum of squared deviations): ONE slot of
// width h (GemmParams::ln_slots = 1, ln_slotw = h).  launch_layernorm_prep: y = LayerNorm(x) w + b (the vision tower's pre-LN,
// reference clip.cpp:1334-1339) written to out32 in the same launch; launch_text_embed with xg != null: y = the embedded rows.
// mu_out != null: the centred form — xg = fp16((y - mean) gamma_next), mu_out[r] = mean (GemmParams::ln_mu of the first consumer).
void launch_layernorm_prep(const float * x, int ldx, const float * w, const float * b, float eps, int rows, int h, float * out32, int ld32,
                           const float * gamma_next, half_t * xg, int ldxg, float2 * stats, hipStream_t stream, float * mu_out = nullptr,
                           const float * class_embd = nullptr, const float * pos0 = nullptr, int T = 0);   // class_embd != null: rows r % T == 0 are class_embd + pos0 (never loaded from x)

// Multi-head self-attention softmax(QK^T)V (reference clip.cpp:1382-1388; causal for text :1101).
// qkv: [rows][3h] fp16 with Q pre-scaled; sequences given by seq_start[nseq+1] (device) or, when
// seq_start == nullptr, uniform length T.  out: [rows][h] fp16.
bool launch_attention(const half_t * qkv, half_t * out, int nseq, int T_uniform, const int * seq_start, int max_len,
                      int h, int n_head, bool causal, hipStream_t stream);

// The same attention in f32 for f32 GGUF files (k_attn_f32.hip; the reference's KQ, soft_max and KQV are f32 for every file type): qkv float [rows][3h]
// with Q pre-scaled, out float [rows][h]; any sequence length, d_head <= 128 and a multiple of 4.
bool launch_attention_f32(const float * qkv, float * out, int nseq, int T_uniform, const int * seq_start, int max_len,
                          int h, int n_head, bool causal, hipStream_t stream);

// im2col for the stride-P patch convolution (reference clip.cpp:1309; ggml conv_2d im2col, fp16):
// imgs [B][S][S][3] interleaved, f32 or (imgs_f16) already rounded to fp16 -> col [B*Np][Kpad] fp16, k = (c*P + ky)*P + kx, zero padded.
void launch_im2col(const void * imgs, bool imgs_f16, half_t * col, int B, int S, int P, int Kpad, hipStream_t stream);

// pooled rows of the last layer: xp[r] = x[src(r)], ap[r] = a[src(r)], src(r) = in_rows[r] or r * in_row_mul (h % 4 == 0)
void launch_gather_rows(const float * x, const half_t * a, const int * in_rows, int in_row_mul, int rows, int h, float * xp, half_t * ap, hipStream_t stream);

// x[b*T + 0][:] = class_embd + pos[0]   (reference clip.cpp:1315-1331, class-token row)
void launch_cls_rows(float * x, const float * class_embd, const float * pos, int B, int T, int h, hipStream_t stream);

// text embedding: x[r][:] = dequant(token_embd[ids[r]]) + pos[r - seq_start(r)]  (reference clip.cpp:1059-1061)
// tok_raw is the token_embd tensor in its ggml block layout (type = ggml type id).
void launch_meta_upload(const int * src_mapped, int * seq, int * last, int n_texts, unsigned * done_mapped, unsigned stamp, hipStream_t stream);
void launch_text_embed(const int32_t * ids, const int * seq_start, int nseq, int rows, const void * tok_raw,
                       int tok_type, const float * pos, int h, float * x, hipStream_t stream,
                       const float * gamma_next = nullptr, half_t * xg = nullptr, int ldxg = 0, float2 * stats = nullptr, float * mu_out = nullptr);

// out[r][:] = v[r][:] / ||v[r]||_2  (reference clip.cpp:1446-1455) or plain copy when !normalize
void launch_l2norm(const float * v, float * out, int rows, int n, bool normalize, hipStream_t stream);

// GPU image preprocessing (k_preproc.hip; reference clip.cpp:728-1008): separable antialiased bicubic resize of the
// shorter side to S, centre crop, normalise.  Tap tables come from the host (preprocess.cpp) so that the result is
// bit-identical to the host path.
struct PreImg {            // one per image
    long long src_off;     // byte offset of the [ny][nx][3] u8 pixels in `raw`
    int nx, ny;            // source size
    int x0, y0;            // crop origin in the resized image
    int ylo, nrows;        // first source row the vertical pass needs / number of such rows
    long long hbuf_off;    // float offset of this image's horizontal-pass rows [nrows][S][3] in `hbuf`
    int th, tv;            // tap-table indices (horizontal, vertical)
};
struct PreTaps { long long w_off; int first_off, count_off, ksize; };   // weights [out][ksize] in wpool; first/count [out] in ipool
void launch_preprocess(const uint8_t * raw, const PreImg * imgs, const PreTaps * taps, const double * wpool, const int * ipool, float * hbuf,
                       float * out, int n_imgs, int S, int max_rows, const float * mean, const float * stdv, hipStream_t stream);

// Zero-shot scoring: per image, softmax_with_sorting (reference clip.cpp:1591-1622) of its similarities with n text
// embeddings.  img [B][dim], txt [n][dim] -> scores [B][n] (descending), indices [B][n].  false if n > 8192 or dim > 4096.
bool launch_zero_shot(const float * img, int B, const float * txt, int n, int dim, float * scores, int * indices, hipStream_t stream);

// fp32 -> fp16 conversion of a [rows][cols] matrix into a padded fp16 matrix (test hooks / inputs)
void launch_f32_to_f16(const float * src, int lds, half_t * dst, int ldd, int rows, int cols, int cols_pad,
                       hipStream_t stream);
void launch_f16_to_f32(const half_t * src, int lds, float * dst, int ldd, int rows, int cols, hipStream_t stream);

}  // namespace clipamd
