// forward.cpp — the ViT / text-transformer forward pass as a sequence of HIP kernel launches.
//
// Follows, step for step, the graphs that reference clip.cpp builds for ggml:
//   vision: clip_image_batch_encode, clip.cpp:1247-1523 (step table in SURVEY §3.3)
//   text  : clip_text_encode,        clip.cpp:1016-1233
// with these MI355X-side fusions: HWC->CHW pack + im2col in one kernel; patch GEMM epilogue writes
// straight into the token-major residual stream and adds the position embedding; q/k/v are one GEMM
// whose epilogue applies bias and the 1/sqrt(d_head) Q scale; attention is one fused kernel; bias +
// GELU and bias + residual live in GEMM epilogues; LayerNorm emits the fp16 operand of the next GEMM.
// Residual stream: f32 [rows][h] in HBM for the whole pass (as in ggml).  5 launches per layer (both LayerNorms folded into the GEMM
// epilogues; 7 where fold_pays() keeps the LayerNorm launches); behind the last layer's attention only the pooled rows are computed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>

#include "model.h"

namespace clipamd {

namespace {

// CLIP_AMD_GUARD=1 (debug aid; tests/test_gpu_parity.py runs the batch-size sweeps under it): 4 KB of canary bytes behind every buffer carved
// out of an activation workspace, written before a forward and verified after it (synchronously; graphs off) — a kernel that stores past the
// end of its buffer fails the call instead of silently touching a neighbour that the shape at hand happens not to read
constexpr size_t kGuardBytes = 4096;
constexpr int kGuardByte = 0xA5;
bool guard_mode() {
    static const bool on = [] { const char * e = getenv("CLIP_AMD_GUARD"); return e && e[0] == '1'; }();
    return on;
}

struct Carver {
    uint8_t * base;
    size_t off = 0;
    std::vector<size_t> * gaps = nullptr;       // guard mode: offsets of the canary blocks
    explicit Carver(void * b, std::vector<size_t> * g = nullptr) : base((uint8_t *)b), gaps(g) { if (gaps) gaps->clear(); }
    template <typename T> T * take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T * p = base ? (T *)(base + off) : nullptr;
        off += n * sizeof(T);
        if (guard_mode()) {
            if (gaps) gaps->push_back(off);
            off += kGuardBytes;
        }
        return p;
    }
};

bool guard_arm(clip_ctx * ctx) {
    if (!guard_mode()) return true;
    // ... and the whole workspace starts as NaN bit patterns (0xFF bytes: fp16 and f32 NaN, int -1): a kernel that lets bytes nobody wrote
    // reach an embedding (a padding column multiplied by a zero weight is enough: NaN x 0) shows up as a non-finite result
    if (hipMemsetAsync(ctx->ws.base, 0xFF, ctx->ws.bytes, ctx->stream) != hipSuccess) return false;
    for (size_t o : ctx->guard_gaps)
        if (hipMemsetAsync((uint8_t *)ctx->ws.base + o, kGuardByte, kGuardBytes, ctx->stream) != hipSuccess) return false;
    static const bool selftest = [] { const char * e = getenv("CLIP_AMD_GUARD_SELFTEST"); return e && e[0] == '1'; }();
    if (selftest && !ctx->guard_gaps.empty())      // the checker's own test: one stray byte, 100 bytes behind the last buffer
        return hipMemsetAsync((uint8_t *)ctx->ws.base + ctx->guard_gaps.back() + 100, 0, 1, ctx->stream) == hipSuccess;
    return true;
}

bool guard_check(clip_ctx * ctx, const char * who) {
    if (!guard_mode()) return true;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    std::vector<uint8_t> h(kGuardBytes);
    bool ok = true;
    for (size_t i = 0; i < ctx->guard_gaps.size(); i++) {
        if (hipMemcpy(h.data(), (uint8_t *)ctx->ws.base + ctx->guard_gaps[i], kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (size_t k = 0; k < kGuardBytes; k++)
            if (h[k] != (uint8_t)kGuardByte) {
                fprintf(stderr, "%s: CLIP_AMD_GUARD: workspace buffer %zu was written %zu bytes past its end\n", who, i, k);
                ok = false;
                break;
            }
    }
    return ok;
}

struct ProfScope {
    clip_ctx * ctx;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    std::string tag;
    double flops, bytes;
    ProfScope(clip_ctx * c, const char * family, long m, long n, long k, double fl, double by) : ctx(c), on(c->profiling), flops(fl), bytes(by) {
        if (!on) return;
        char buf[96];
        snprintf(buf, sizeof buf, "%s:%ldx%ldx%ld", family, m, n, k);
        tag = buf;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, ctx->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->pending.push_back({a, b, tag, flops, bytes});
    }
};

double weight_bytes(const DevWeight & W) {
    const double nblk = (double)(W.K / 32) * W.N;
    switch (W.wtype) {
    case W_F16: return (double)W.N * W.K * 2;
    case W_Q4_0: return nblk * 18;
    case W_Q4_1: return nblk * 20;
    case W_Q5_0: return nblk * 22;
    case W_Q5_1: return nblk * 24;
    case W_Q8_0: return nblk * 34;
    case W_F32: return (double)W.N * W.K * 4;
    }
    return 0;
}

// f32 GGUF file: the activations between the kernels are f32 too (k_gemm_f32.hip act_f32, k_attn_f32.hip; the half_t workspace pointers then
// address float rows: twice the elements are carved).  CLIP_AMD_F32_ACTS=0 at load time restores the fp16 activations of round 5 (A/B: clip_ctx::f32_acts).
bool acts_f32(const clip_ctx * ctx, const DevTower & tw) {
    return ctx->f32_acts && !tw.layers.empty() && tw.layers[0].qkv.wtype == W_F32 && tw.layers[0].o.wtype == W_F32 && tw.layers[0].ff1.wtype == W_F32 && tw.layers[0].ff2.wtype == W_F32 &&
           tw.proj.wtype == W_F32;
}
// LayerNorm of rows of x into an activation buffer: fp16, or f32 behind the same pointer
void layernorm_act(bool f32, const float * x, int ldx, const int * in_rows, int in_row_mul, const float * w, const float * b, float eps, int rows, int h, half_t * dst,
                   hipStream_t s) {
    if (f32) launch_layernorm(x, ldx, in_rows, in_row_mul, w, b, eps, rows, h, nullptr, 0, (float *)dst, h, s);
    else launch_layernorm(x, ldx, in_rows, in_row_mul, w, b, eps, rows, h, dst, h, nullptr, 0, s);
}

// the device carries other work next to this context's: declared by the caller, or this context is / has a busy sibling (two-tower pair calls)
bool device_shared(const clip_ctx * ctx) { return ctx->device_shared || ctx->sibling_busy || ctx->owner != nullptr; }

void gemm(clip_ctx * ctx, const char * what, const GemmParams & p0, int epi) {
    GemmParams p = p0;
    p.shared_device = device_shared(ctx);
    p.sk_ws = ctx->sk_ws; p.sk_ws_floats = ctx->sk_ws_floats; p.sk_cnt = ctx->sk_cnt; p.sk_cnt_n = ctx->sk_cnt_n;
    if (!ctx->profiling) {
        launch_gemm(p, epi, 0, ctx->stream);
        return;
    }
    const int wt = p.w16_pre ? (int)W_F16 : p.W.wtype;       // a dequantised panel is multiplied as an f16 weight (launch_gemm)
    const int tile = gemm_tile_for(p.M, p.W.N, p.W.Kpad, wt != W_F16, p.shared_device);
    const bool panel = gemm_tile_uses_panel(tile) && wt == W_F16;
    const double fl = 2.0 * p.M * (double)p.W.N * p.W.K;
    const double wb = wt == W_F16 ? (double)p.W.N * p.W.K * 2 : weight_bytes(p.W);
    const double by = wb + (double)p.M * p.W.K * 2 + (double)p.M * p.W.N * (epi == EPI_F16 || epi == EPI_GELU_F16 || epi == EPI_QGELU_F16 ? 2 : epi == EPI_RESID_F32 ? 8 : 4)   // fused residual: read + write
                      + (p.xg_out ? (double)p.M * p.W.N * 2 : 0.0);     // LayerNorm fold: the residual epilogue also writes the next GEMM's fp16 operand
    // tag = kernel instantiation (matches the rocprofv3 kernel names gemm_dma_kernel<WT, BM, BN, EPI> / gemm8_kernel<TM, EPI>) + role
    char fam[96];
    if (wt == W_F32) snprintf(fam, sizeof fam, "gemm_f32_kernel<%d>/%s", epi, what);      // f32 file: exact-f32 MFMA (k_gemm_f32.hip)
    else if (panel && tile % 1000 == 261 && (epi == EPI_F16 || epi == EPI_GELU_F16 || epi == EPI_QGELU_F16)) snprintf(fam, sizeof fam, "gemm32_kernel<%d,4,%d>/%s", tile / 1000 == 320 ? 5 : 4, epi, what);
    else if (panel && tile % 1000 >= 259) snprintf(fam, sizeof fam, "gemm4_kernel<%d>/%s", epi, what);   // (+ a short second launch for the rows past the whole rounds)
    else if (panel) snprintf(fam, sizeof fam, "gemm8_kernel<%d,%d>/%s", tile / 32000, epi, what);
    else if (gemm_tile_is_ring(tile)) snprintf(fam, sizeof fam, "gemm_ring_kernel<%d,%d,4,%d,%d>/%s", wt, tile % 1000, tile % 1000 >= 128 ? 4 : 2, epi, what);
    else snprintf(fam, sizeof fam, "gemm_dma_kernel<%d,%d,%d,%d>/%s", wt, gemm_tile_uses_panel(tile) ? 160 : tile / 1000, gemm_tile_uses_panel(tile) ? 128 : tile % 1000, epi, what);
    ProfScope ps(ctx, fam, p.M, p.W.N, p.W.K, fl, by);
    launch_gemm(p, epi, 0, ctx->stream);
}

// Large-M layers (k_gemm8.hip): the block-quantised weights of ONE layer are dequantised once into fp16 panels (L2 / Infinity-Cache
// resident scratch, re-used by the next layer) before its four GEMMs; f16 weights are multiplied where they lie.
struct LayerPanels { const half_t * qkv = nullptr, * o = nullptr, * ff1 = nullptr, * ff2 = nullptr; };

bool ensure_panel(clip_ctx * ctx, size_t halfs) {
    if (ctx->w16_panel_halfs >= halfs) return true;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->w16_panel) (void)hipFree(ctx->w16_panel);
    ctx->w16_panel = nullptr;
    ctx->w16_panel_halfs = 0;
    if (hipMalloc((void **)&ctx->w16_panel, halfs * sizeof(half_t)) != hipSuccess) { (void)hipGetLastError(); return false; }
    ctx->w16_panel_halfs = halfs;
    return true;
}

// res: this layer's four entries of the resident-panel table (resident_panels below), or null
LayerPanels dequant_layer(clip_ctx * ctx, const DevLayer & l, int rows, const half_t * const * res = nullptr) {
    LayerPanels lp;
    const DevWeight * ws[4] = {&l.qkv, &l.o, &l.ff1, &l.ff2};
    const half_t ** slot[4] = {&lp.qkv, &lp.o, &lp.ff1, &lp.ff2};
    const DevWeight * jw[4];
    half_t * jo[4];
    int nj = 0;
    size_t need = 0;
    bool todo[4];
    for (int i = 0; i < 4; i++) {
        // kept from the first large batch: nothing to dequantise.  (A q/k/v or FFN-up panel exists for the 32 x 32 x 16 kernel only: where the
        // heuristic does not pick that kernel — device shared with the other tower, another row count — the fused kernels multiply the quantised planes)
        const bool for32 = i == 0 || i == 2;
        if (res && res[i] && (!for32 || gemm_tile_for(rows, ws[i]->N, ws[i]->Kpad, false, device_shared(ctx)) % 1000 == 261)) { *slot[i] = res[i]; todo[i] = false; continue; }
        todo[i] = ws[i]->wtype != W_F16 && ws[i]->wtype != W_F32 && gemm_tile_uses_panel(gemm_tile_for(rows, ws[i]->N, ws[i]->Kpad, true));
        if (todo[i]) need += (size_t)ws[i]->Npad * ws[i]->Kpad;
    }
    if (!need || !ensure_panel(ctx, need)) return lp;
    size_t off = 0;
    for (int i = 0; i < 4; i++)
        if (todo[i]) {
            jw[nj] = ws[i];
            jo[nj] = ctx->w16_panel + off;
            *slot[i] = jo[nj];
            off += (size_t)ws[i]->Npad * ws[i]->Kpad;
            nj++;
        }
    ProfScope ps(ctx, "dequant_layer", rows, nj, 0, 0, (double)need * 2.5625);
    launch_dequant(jw, jo, nj, ctx->stream);
    return lp;
}

// ---- small-M path (rows <= 64: one ViT-B/32 image, one text, a few short labels): k_skinny.hip, 5 launches per layer ----
bool skinny_enabled() {
    static int on = -1;
    if (on < 0) { const char * e = getenv("CLIP_AMD_SKINNY"); on = !(e && e[0] == '0'); }
    return on != 0;
}

#ifdef CLIPAMD_SK_TIMING   // tuning builds (scripts/build_sk_timing.sh): per-launch phase stamps of the small-M kernels
unsigned long long * g_sk_stamps = nullptr;
int g_sk_stamp_launch = 0;
constexpr int SK_STAMP_LAUNCHES = 512;
#endif

void skinny(clip_ctx * ctx, const char * what, const SkinnyParams & p0, int epi) {
#ifdef CLIPAMD_SK_TIMING
    SkinnyParams p = p0;
    if (!g_sk_stamps && hipMalloc((void **)&g_sk_stamps, SK_STAMP_LAUNCHES * 16 * 8) == hipSuccess) (void)hipMemset(g_sk_stamps, 0, SK_STAMP_LAUNCHES * 16 * 8);
    if (g_sk_stamps) p.stamps = g_sk_stamps + (size_t)(g_sk_stamp_launch++ % SK_STAMP_LAUNCHES) * 16;
#else
    const SkinnyParams & p = p0;
#endif
    if (!ctx->profiling) { launch_skinny(p, epi, ctx->stream); return; }
    char fam[96];
    // tag = kernel instantiation as rocprofv3 prints it: skinny_kernel<WT, MF, NW, EPI, LNA> (MF = 1; 8 waves for the long-K residual / patch GEMMs)
    const int nw = (!p.x32 && p.W.Kpad >= 2048 && (epi == EPI_RESID_F32 || epi == EPI_PATCH_F32)) ? 8 : 4;
    snprintf(fam, sizeof fam, "skinny_kernel<%d,1,%d,%d,%s>/%s", p.W.wtype, nw, epi, p.x32 ? "true" : "false", what);
    const double fl = 2.0 * p.M * (double)p.W.N * p.W.K;
    const double by = weight_bytes(p.W) + (double)p.M * p.W.K * (p.x32 ? 4 : 2) + (double)p.M * p.W.N * (epi == EPI_RESID_F32 ? 8 : epi == EPI_F32 || epi == EPI_PATCH_F32 ? 4 : 2);
    ProfScope ps(ctx, fam, p.M, p.W.N, p.W.K, fl, by);
    launch_skinny(p, epi, ctx->stream);
}

int skinny_row_limit() {     // rows up to which the small-M kernels carry the layers (CLIP_AMD_SKINNY_ROWS: tuning override)
    static int lim = -1;
    if (lim < 0) {
        const char * e = getenv("CLIP_AMD_SKINNY_ROWS");
        lim = e ? atoi(e) : 64;
        if (lim > SKINNY_MAX_ROWS) lim = SKINNY_MAX_ROWS;
    }
    return lim;
}

bool layers_fit_skinny(const DevTower & tw, int rows, int h, int ff) {
    if (!skinny_enabled() || rows <= 0 || rows > skinny_row_limit() || h > 2048 || h % 16 || h / 16 > 128 || tw.layers.empty()) return false;
    const DevLayer & l = tw.layers[0];
    if (l.qkv.wtype == W_F32 || l.o.wtype == W_F32 || l.ff1.wtype == W_F32 || l.ff2.wtype == W_F32) return false;   // f32 file: k_gemm_f32.hip carries every row count
    return l.qkv.K == l.qkv.Kpad && l.ff1.K == l.ff1.Kpad && l.o.N == h && l.ff2.N == h && l.ff1.N == ff;
}

// precondition: slot 0 of ctx->sk_stats holds the row statistics of x (launch_row_stats)
bool run_layers_skinny(clip_ctx * ctx, const DevTower & tw, int rows, int h, int nh, int ff, float eps, int nseq, int T_uniform,
                       const int * d_seq_start, int max_len, bool causal, float * x, half_t * qkv, half_t * att, half_t * mid) {
    hipStream_t s = ctx->stream;
    const int dh = h / nh;
    const float qscale = 1.0f / sqrtf((float)dh);
    const int act = ctx->use_gelu ? EPI_GELU_F16 : EPI_QGELU_F16;
    float2 * stA = ctx->sk_stats, * stB = ctx->sk_stats + (size_t)SKINNY_MAX_ROWS * 128;
    int slotsA = 1;
    for (const DevLayer & l : tw.layers) {
        SkinnyParams q;   // LN1 + q/k/v projection (+ Q scale after the bias, clip.cpp:1363)
        q.x32 = x; q.ldx = h; q.ln_w = l.ln1_w; q.ln_b = l.ln1_b; q.eps = eps; q.stats_in = stA; q.stats_slots = slotsA;
        q.M = rows; q.W = l.qkv; q.bias = l.qkv_b; q.out = qkv; q.ldc = 3 * h; q.qscale = qscale; q.qcols = h;
        skinny(ctx, "ln1_qkv", q, EPI_F16);
        {
            const double afl = 4.0 * (double)nseq * nh * (double)max_len * max_len * dh;
            ProfScope ps(ctx, "attention", nseq * nh, max_len, dh, afl, (double)rows * h * 8);
            if (!launch_attention(qkv, att, nseq, T_uniform, d_seq_start, max_len, h, nh, causal, s)) {
                fprintf(stderr, "clip (hip): attention kernel does not support T=%d d_head=%d\n", max_len, dh);
                return false;
            }
        }
        SkinnyParams o;   // out-projection + residual; leaves the statistics of the new rows for LN2
        o.A16 = att; o.lda = h; o.M = rows; o.W = l.o; o.bias = l.o_b; o.out = x; o.ldc = h; o.resid = x; o.stats_out = stB;
        skinny(ctx, "out_resid", o, EPI_RESID_F32);
        SkinnyParams u;   // LN2 + FFN-up + activation
        u.x32 = x; u.ldx = h; u.ln_w = l.ln2_w; u.ln_b = l.ln2_b; u.eps = eps; u.stats_in = stB; u.stats_slots = h / 16;
        u.M = rows; u.W = l.ff1; u.bias = l.ff1_b; u.out = mid; u.ldc = ff;
        skinny(ctx, "ln2_ffn_up", u, act);
        SkinnyParams d;   // FFN-down + residual; statistics for the next layer's LN1
        d.A16 = mid; d.lda = ff; d.M = rows; d.W = l.ff2; d.bias = l.ff2_b; d.out = x; d.ldc = h; d.resid = x; d.stats_out = stA;
        skinny(ctx, "ffn_down_resid", d, EPI_RESID_F32);
        slotsA = h / 16;
    }
    return true;
}

// The small-M chain with the LayerNorm folded (kernels.h SkinnyParams::ln_c / xg_out): 5 launches per layer as above, but the q/k/v and
// FFN-up kernels read the fp16 operand xn = fp16(x gamma) the residual epilogues leave (half the bytes of the f32 rows, no gamma / beta
// staging, no normalisation in registers) and finish the LayerNorm in their epilogue.  r03 stamps: the LayerNorm prologue was 2.7 us of
// the 6-7 us these two kernels live.  Precondition as run_layers_fold: xn and ONE statistics slot per row from the entry kernel.
bool run_layers_skinny_fold(clip_ctx * ctx, const DevTower & tw, int rows, int h, int nh, int ff, float eps, int nseq, int T_uniform,
                            const int * d_seq_start, int max_len, bool causal, float * x, half_t * xn, half_t * qkv, half_t * att, half_t * mid,
                            float2 * stats, int stats_stride, float * mu) {
    hipStream_t s = ctx->stream;
    const int dh = h / nh;
    const float qscale = 1.0f / sqrtf((float)dh);
    const int act = ctx->use_gelu ? EPI_GELU_F16 : EPI_QGELU_F16;
    int slots = 1, slotw = h;
    // centring offsets (kernels.h GemmParams::xg_mu): two [stats_stride] buffers; every consumer reads the one its operand was centred on
    // and leaves its own row means in the other, which the next producer centres on.  mu == null: the uncentred form.
    float * mu_cur = mu, * mu_alt = mu ? mu + stats_stride : nullptr;
    auto consume = [&](SkinnyParams & p, const float * c, const float * bf) {
        p.A16 = xn; p.lda = h; p.bias = bf; p.ln_c = c; p.fstats = stats; p.fslots = slots; p.fslotw = slotw; p.fstride = stats_stride; p.eps = eps;
        if (mu) { p.ln_mu = mu_cur; p.mu_out = mu_alt; std::swap(mu_cur, mu_alt); }
    };
    auto produce = [&](SkinnyParams & p, const float * gamma_next) {
        if (!gamma_next) return;
        p.xg_mu = mu_cur;
        p.xg_out = xn; p.ldxg = h; p.xg_gamma = gamma_next; p.fstats_out = stats; p.fstride_out = stats_stride;
        slots = h / 16; slotw = 16;
    };
    for (size_t li = 0; li < tw.layers.size(); li++) {
        const DevLayer & l = tw.layers[li];
        SkinnyParams q;   // q/k/v projection of the folded LN1 (+ Q scale after the bias, clip.cpp:1363)
        q.M = rows; q.W = l.qkv; q.out = qkv; q.ldc = 3 * h; q.qscale = qscale; q.qcols = h;
        consume(q, l.qkv_c, l.qkv_bf);
        skinny(ctx, "ln1_qkv", q, EPI_F16);
        {
            const double afl = 4.0 * (double)nseq * nh * (double)max_len * max_len * dh;
            ProfScope ps(ctx, "attention", nseq * nh, max_len, dh, afl, (double)rows * h * 8);
            if (!launch_attention(qkv, att, nseq, T_uniform, d_seq_start, max_len, h, nh, causal, s)) {
                fprintf(stderr, "clip (hip): attention kernel does not support T=%d d_head=%d\n", max_len, dh);
                return false;
            }
        }
        SkinnyParams o;   // out-projection + residual; leaves xn = fp16(x ln2_w) and the 16-column statistics for LN2
        o.A16 = att; o.lda = h; o.M = rows; o.W = l.o; o.bias = l.o_b; o.out = x; o.ldc = h; o.resid = x;
        produce(o, l.ln2_w);
        skinny(ctx, "out_resid", o, EPI_RESID_F32);
        SkinnyParams u;   // FFN-up of the folded LN2 + activation
        u.M = rows; u.W = l.ff1; u.out = mid; u.ldc = ff;
        consume(u, l.ff1_c, l.ff1_bf);
        skinny(ctx, "ln2_ffn_up", u, act);
        SkinnyParams d;   // FFN-down + residual; operand and statistics for the next layer's LN1
        d.A16 = mid; d.lda = ff; d.M = rows; d.W = l.ff2; d.bias = l.ff2_b; d.out = x; d.ldc = h; d.resid = x;
        produce(d, li + 1 < tw.layers.size() ? tw.layers[li + 1].ln1_w : nullptr);
        skinny(ctx, "ffn_down_resid", d, EPI_RESID_F32);
    }
    return true;
}

// L x { LN1, QKV, attention, out-proj(+res), LN2, FFN-up(+act), FFN-down(+res) }   (clip.cpp:1342-1423 / :1064-1143)
bool run_layers(clip_ctx * ctx, const DevTower & tw, int rows, int h, int nh, int ff, float eps, int nseq, int T_uniform,
                const int * d_seq_start, int max_len, bool causal, float * x, half_t * xn, half_t * qkv, half_t * att, half_t * mid, bool prune_last,
                const half_t * const * resp = nullptr) {
    hipStream_t s = ctx->stream;
    const int dh = h / nh;
    const float qscale = 1.0f / sqrtf((float)dh);
    const int act = ctx->use_gelu ? EPI_GELU_F16 : EPI_QGELU_F16;
    const bool f32 = acts_f32(ctx, tw);                       // f32 file: xn / qkv / att / mid hold float rows
    for (size_t li = 0; li < tw.layers.size(); li++) {
        const DevLayer & l = tw.layers[li];
        const LayerPanels lp = dequant_layer(ctx, l, rows, resp ? resp + 4 * li : nullptr);
        {
            ProfScope ps(ctx, "layernorm", rows, h, 0, 0, (double)rows * h * 6);
            layernorm_act(f32, x, h, nullptr, 1, l.ln1_w, l.ln1_b, eps, rows, h, xn, s);
        }
        GemmParams p;
        p.A = xn; p.lda = h; p.M = rows; p.W = l.qkv; p.bias = l.qkv_b; p.out = qkv; p.ldc = 3 * h; p.w16_pre = lp.qkv;
        p.qscale = qscale; p.qcols = h;   // Q = (W_q x + b_q) / sqrt(d_head): scale after bias (clip.cpp:1363)
        p.act_f32 = f32;
        gemm(ctx, "gemm_qkv", p, EPI_F16);
        {
            const double afl = 4.0 * (double)nseq * nh * (double)max_len * max_len * dh;
            ProfScope ps(ctx, "attention", nseq * nh, max_len, dh, afl, (double)rows * h * 8);
            if (!(f32 ? launch_attention_f32((const float *)qkv, (float *)att, nseq, T_uniform, d_seq_start, max_len, h, nh, causal, s)
                      : launch_attention(qkv, att, nseq, T_uniform, d_seq_start, max_len, h, nh, causal, s))) {
                fprintf(stderr, "clip (hip): attention kernel does not support T=%d d_head=%d\n", max_len, dh);
                return false;
            }
        }
        if (prune_last && li + 1 == tw.layers.size()) break;     // the rest of the last layer runs on the pooled rows only (pooled_tail)
        GemmParams po;
        po.A = att; po.lda = h; po.M = rows; po.W = l.o; po.bias = l.o_b; po.out = x; po.ldc = h; po.resid = x; po.w16_pre = lp.o; po.act_f32 = f32;
        gemm(ctx, "gemm_out", po, EPI_RESID_F32);
        {
            ProfScope ps(ctx, "layernorm", rows, h, 0, 0, (double)rows * h * 6);
            layernorm_act(f32, x, h, nullptr, 1, l.ln2_w, l.ln2_b, eps, rows, h, xn, s);
        }
        GemmParams p1;
        p1.A = xn; p1.lda = h; p1.M = rows; p1.W = l.ff1; p1.bias = l.ff1_b; p1.out = mid; p1.ldc = ff; p1.w16_pre = lp.ff1; p1.act_f32 = f32;
        gemm(ctx, "gemm_ffn_up", p1, act);
        GemmParams p2;
        p2.A = mid; p2.lda = ff; p2.M = rows; p2.W = l.ff2; p2.bias = l.ff2_b; p2.out = x; p2.ldc = h; p2.resid = x; p2.w16_pre = lp.ff2; p2.act_f32 = f32;
        gemm(ctx, "gemm_ffn_down", p2, EPI_RESID_F32);
    }
    return true;
}

// The same chain with every LayerNorm folded into the GEMMs around it (gemm_common.h; reference ops clip.cpp:1350-1355,1400-1405 and
// text :1071-1076,1121-1126): 5 launches per layer.  The residual epilogues (out-projection, FFN-down) leave xg = fp16(x gamma_next)
// in xn and the partial row statistics in `stats`; the q/k/v and FFN-up epilogues apply rstd (acc - mean c) + b'.
// Precondition: xn = fp16(x * ln1_w of layer 0) and stats = ONE slot per row over all h columns (launch_layernorm_prep / launch_text_embed).
bool run_layers_fold(clip_ctx * ctx, const DevTower & tw, int rows, int h, int nh, int ff, float eps, int nseq, int T_uniform,
                     const int * d_seq_start, int max_len, bool causal, float * x, half_t * xn, half_t * qkv, half_t * att, half_t * mid,
                     float2 * stats, int stats_stride, float * mu, bool prune_last, const half_t * const * ff2p = nullptr) {
    hipStream_t s = ctx->stream;
    const int dh = h / nh;
    const float qscale = 1.0f / sqrtf((float)dh);
    const int act = ctx->use_gelu ? EPI_GELU_F16 : EPI_QGELU_F16;
    int slots = 1, slotw = h;
    float * mu_cur = mu, * mu_alt = mu ? mu + stats_stride : nullptr;      // centring offsets, as run_layers_skinny_fold
    auto consume = [&](GemmParams & p, const float * c, const float * bf) {
        p.A = xn; p.lda = h; p.bias = bf; p.ln_c = c; p.ln_stats = stats; p.ln_slots = slots; p.ln_slotw = slotw; p.ln_stride = stats_stride; p.ln_eps = eps;
        if (mu) { p.ln_mu = mu_cur; p.mu_out = mu_alt; std::swap(mu_cur, mu_alt); }
    };
    auto produce = [&](GemmParams & p, const float * gamma_next) {
        if (!gamma_next) return;                              // last layer: the pooled rows go through the post-LN launch
        p.xg_mu = mu_cur;
        p.xg_out = xn; p.ldxg = h; p.xg_gamma = gamma_next; p.stats_out = stats; p.stats_stride = stats_stride;
        const bool quant = !p.w16_pre && p.W.wtype != W_F16;
        slotw = gemm_fold_slotw_for(p.M, p.W.N, p.W.Kpad, quant, device_shared(ctx));
        slots = h / slotw;
    };
    for (size_t li = 0; li < tw.layers.size(); li++) {
        const DevLayer & l = tw.layers[li];
        const LayerPanels lp = dequant_layer(ctx, l, rows, ff2p ? ff2p + 4 * li : nullptr);
        GemmParams p;
        p.M = rows; p.W = l.qkv; p.out = qkv; p.ldc = 3 * h; p.w16_pre = lp.qkv;
        p.qscale = qscale; p.qcols = h;   // Q = (W_q LN(x) + b_q) / sqrt(d_head): scale after bias (clip.cpp:1363)
        consume(p, l.qkv_c, l.qkv_bf);
        gemm(ctx, "gemm_qkv", p, EPI_F16);
        {
            const double afl = 4.0 * (double)nseq * nh * (double)max_len * max_len * dh;
            ProfScope ps(ctx, "attention", nseq * nh, max_len, dh, afl, (double)rows * h * 8);
            if (!launch_attention(qkv, att, nseq, T_uniform, d_seq_start, max_len, h, nh, causal, s)) {
                fprintf(stderr, "clip (hip): attention kernel does not support T=%d d_head=%d\n", max_len, dh);
                return false;
            }
        }
        if (prune_last && li + 1 == tw.layers.size()) break;     // the rest of the last layer runs on the pooled rows only (pooled_tail)
        GemmParams po;
        po.A = att; po.lda = h; po.M = rows; po.W = l.o; po.bias = l.o_b; po.out = x; po.ldc = h; po.resid = x; po.w16_pre = lp.o;
        produce(po, l.ln2_w);
        gemm(ctx, "gemm_out", po, EPI_RESID_F32);
        GemmParams p1;
        p1.M = rows; p1.W = l.ff1; p1.out = mid; p1.ldc = ff; p1.w16_pre = lp.ff1;
        consume(p1, l.ff1_c, l.ff1_bf);
        gemm(ctx, "gemm_ffn_up", p1, act);
        GemmParams p2;
        p2.A = mid; p2.lda = ff; p2.M = rows; p2.W = l.ff2; p2.bias = l.ff2_b; p2.out = x; p2.ldc = h; p2.resid = x; p2.w16_pre = lp.ff2;
        produce(p2, li + 1 < tw.layers.size() ? tw.layers[li + 1].ln1_w : nullptr);
        gemm(ctx, "gemm_ffn_down", p2, EPI_RESID_F32);
    }
    return true;
}

// Only ONE row per sequence leaves the tower — the class-token row of an image (reference clip.cpp:1426-1431), the last token of a text
// (:1154-1155) — and a row of the last layer's out-projection / FFN depends on no other row.  So the last layer runs q/k/v and attention on
// every row (keys and values of all tokens feed the pooled query) and everything behind the attention on the `n` pooled rows only:
// 1 / T of the out-projection and of both FFN GEMMs, i.e. ~(9 / 12) / L of the tower's linear FLOPs for free (ViT-B/32 batch 256: three
// GEMMs of 12800 rows become three of 256).  Same arithmetic per row as the full-row layer (LayerNorm launch form), different tiles.
// xp [n][h] f32 (out: the pooled rows of the final residual stream), ap / xnp [n][h] fp16, midp [n][ff] fp16: workspace of the caller.
bool pooled_tail(clip_ctx * ctx, const DevLayer & l, int n, int h, int ff, float eps, const float * x, const half_t * att, const int * in_rows,
                 int in_row_mul, float * xp, half_t * ap, half_t * xnp, half_t * midp) {
    hipStream_t s = ctx->stream;
    const int act = ctx->use_gelu ? EPI_GELU_F16 : EPI_QGELU_F16;
    const bool f32 = ctx->f32_acts && l.o.wtype == W_F32 && l.ff1.wtype == W_F32 && l.ff2.wtype == W_F32;     // f32 file: att / ap / xnp / midp hold float rows
    {
        ProfScope ps(ctx, "gather_pooled", n, h, 0, 0, (double)n * h * 12);
        if (f32) {
            launch_gather_rows(x, nullptr, in_rows, in_row_mul, n, h, xp, nullptr, s);
            launch_gather_rows((const float *)att, nullptr, in_rows, in_row_mul, n, h, (float *)ap, nullptr, s);
        } else launch_gather_rows(x, att, in_rows, in_row_mul, n, h, xp, ap, s);
    }
    GemmParams po;
    po.A = ap; po.lda = h; po.M = n; po.W = l.o; po.bias = l.o_b; po.out = xp; po.ldc = h; po.resid = xp; po.act_f32 = f32;
    gemm(ctx, "gemm_out_pooled", po, EPI_RESID_F32);
    {
        ProfScope ps(ctx, "layernorm", n, h, 0, 0, (double)n * h * 6);
        layernorm_act(f32, xp, h, nullptr, 1, l.ln2_w, l.ln2_b, eps, n, h, xnp, s);
    }
    GemmParams p1;
    p1.A = xnp; p1.lda = h; p1.M = n; p1.W = l.ff1; p1.bias = l.ff1_b; p1.out = midp; p1.ldc = ff; p1.act_f32 = f32;
    gemm(ctx, "gemm_ffn_up_pooled", p1, act);
    GemmParams p2;
    p2.A = midp; p2.lda = ff; p2.M = n; p2.W = l.ff2; p2.bias = l.ff2_b; p2.out = xp; p2.ldc = h; p2.resid = xp; p2.act_f32 = f32;
    gemm(ctx, "gemm_ffn_down_pooled", p2, EPI_RESID_F32);
    return true;
}

// RESIDENT fp16 panels of block-quantised weights (round 4; profiles/r04_experiments.txt section 9): the FFN-down weight where the 8-wave
// kernel on a panel beats the fused-dequant 4-wave kernel but a per-layer dequantisation launch would cost more than it gains (K >= 2048
// and >= 200 tiles of 160 x 256 — ViT-B/32 at batch 256: 75.9 -> 71.7 us per launch, +1.0 ... +1.5 % on the BASELINE configuration on four
// boxes; 57 MB, built on the first such batch).  Measured and NOT done: panels for all four weights of such a layer lose 2.3 % (the K = 768
// GEMMs are slower on fp16 weights and everything slows down with the 4 x larger weight stream, as round 2 found); keeping the panels that
// dequant_layer() rebuilds per layer at M >= 32768 (ViT-L/14 q5_1 batch 128: 606 MB) saves the 16 us launches but the GEMMs then read
// cold fp16 weights from HBM instead of a panel that is hot in L2 / Infinity Cache: 4779 vs 4781 img/s.
// Table entry 4 * layer + {0 q/k/v, 1 out, 2 FFN-up, 3 FFN-down}; null = multiply the quantised planes.  Built outside graph captures.
const half_t * const * resident_panels(clip_ctx * ctx, const DevTower & tw, int which, int rows) {
    if (!ctx->resident_panels_on || tw.layers.empty()) return nullptr;
    if (ctx->owner) {
        // a sibling / twin multiplies the owner's weight image and borrows the owner's panels too (read-only, same device) instead of
        // building a second 57 MB copy; it never builds: until the owner has, it runs the fused kernels (same bits — tested)
        clip_ctx * o = ctx->owner;
        return (o->res_panel_mask[which] && o->res_panels[which].size() == 4 * tw.layers.size()) ? o->res_panels[which].data() : nullptr;
    }
    const DevLayer & l0 = tw.layers[0];
    unsigned want = 0;
    if (l0.ff2.wtype != W_F16 && l0.ff2.wtype != W_F32 && rows >= 4096 && gemm_tile_uses_panel(gemm_tile_for(rows, l0.ff2.N, l0.ff2.Kpad, false))) want |= 8u;
    if (l0.o.wtype != W_F16 && l0.o.wtype != W_F32 && rows >= 4096 && rows < 32768 && gemm_tile_uses_panel(gemm_tile_for(rows, l0.o.N, l0.o.Kpad, false))) want |= 2u;
    // round 6: q/k/v and FFN-up where the 32 x 32 x 16 kernel (k_gemm32.hip: fp16 x fp16) takes the shape (ViT-B/32 at batch 256: 99 MB per tower)
    // ... from 32768 rows only (ViT-L/14 / H/14 batches, where the alternative is a per-layer dequantisation launch + the 256 x 256 kernel).  Below that
    // (ViT-B/32-class batches) the fused-dequant two-per-CU kernels, after the instruction trims of round 6, are ahead of the 32 x 32 x 16 kernel on 99 MB of
    // resident panels for block-quantised files: 80.5-80.6 k against 79.0-79.3 k img/s, texts 161 k against 163 k (profiles/r06_experiments.txt section 19);
    // f16 files, which need no panel, keep that kernel (+2 %).
    if (rows >= 32768 && l0.qkv.wtype != W_F16 && l0.qkv.wtype != W_F32 && gemm_tile_for(rows, l0.qkv.N, l0.qkv.Kpad, false, device_shared(ctx)) % 1000 == 261) want |= 1u;
    if (rows >= 32768 && l0.ff1.wtype != W_F16 && l0.ff1.wtype != W_F32 && gemm_tile_for(rows, l0.ff1.N, l0.ff1.Kpad, false, device_shared(ctx)) % 1000 == 261) want |= 4u;
    if (!want) return nullptr;
    auto & tab = ctx->res_panels[which];
    if ((ctx->res_panel_mask[which] & want) == want && tab.size() == 4 * tw.layers.size()) return tab.data();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return (ctx->res_panel_mask[which] && tab.size() == 4 * tw.layers.size()) ? tab.data() : nullptr;
    want |= ctx->res_panel_mask[which];
    (void)hipStreamSynchronize(ctx->stream);                   // (re)build: nothing may still read the old table
    if (ctx->res_panel_buf[which]) (void)hipFree(ctx->res_panel_buf[which]);
    ctx->res_panel_buf[which] = nullptr;
    ctx->res_panel_mask[which] = 0;
    tab.clear();
    size_t halfs = 0;
    for (const DevLayer & l : tw.layers) {
        const DevWeight * ws[4] = {&l.qkv, &l.o, &l.ff1, &l.ff2};
        for (int i = 0; i < 4; i++) if (want >> i & 1) halfs += (size_t)ws[i]->Npad * ws[i]->Kpad;
    }
    if (hipMalloc((void **)&ctx->res_panel_buf[which], halfs * sizeof(half_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    size_t off = 0;
    for (const DevLayer & l : tw.layers) {
        const DevWeight * ws[4] = {&l.qkv, &l.o, &l.ff1, &l.ff2};
        for (int i = 0; i < 4; i++) {
            if (!(want >> i & 1)) { tab.push_back(nullptr); continue; }
            half_t * o = ctx->res_panel_buf[which] + off;
            launch_dequant(&ws[i], &o, 1, ctx->stream);
            tab.push_back(o);
            off += (size_t)ws[i]->Npad * ws[i]->Kpad;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);                   // one-time build: siblings read the table on their own streams (borrowed above)
    ctx->res_panel_mask[which] = want;
    return tab.data();
}

// The fold pays where the GEMMs run two workgroups per CU (k_gemm.hip, k_gemm_ring.hip): the statistics prologue and the xg / statistics
// tail of one workgroup hide under the other's K loop.  The large-M kernels (k_gemm4.hip, k_gemm8.hip: ONE workgroup per CU) expose both
// — measured r03cfg, ViT-L/14 f16 batch 256: FFN-down +128 us, out-projection +106 us, FFN-up +61 us, q/k/v +31 us per launch against
// 2 x 82 us of LayerNorm launches saved per layer: 4374 img/s folded vs 4690 unfolded — so a tower whose layers touch those kernels at
// this row count keeps the LayerNorm launches.
bool fold_pays(const DevTower & tw, int rows, bool shared) {
    if (tw.layers.empty()) return false;
    const DevLayer & l = tw.layers[0];
    for (const DevWeight * w : {&l.qkv, &l.o, &l.ff1, &l.ff2})
        if (gemm_tile_uses_panel(gemm_tile_for(rows, w->N, w->Kpad, w->wtype != W_F16, shared)) &&
            gemm_tile_for(rows, w->N, w->Kpad, w->wtype != W_F16, shared) % 1000 != 261) return false;      // (k_gemm32.hip carries the consumer half at no visible cost)
    return true;
}

bool check_device(clip_ctx * ctx, const char * who) {
    if (!ctx || ctx->device < 0) {
        fprintf(stderr, "%s: no HIP device bound to this context — the encoders have no CPU fallback\n", who);
        return false;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) {
        fprintf(stderr, "%s: hipSetDevice(%d) failed\n", who, ctx->device);
        return false;
    }
    return true;
}

bool launch_ok(const char * who) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "%s: HIP launch error: %s\n", who, hipGetErrorString(e));
        return false;
    }
    return true;
}

}  // namespace

void drop_graphs(clip_ctx * ctx) {
    for (auto & g : ctx->vgraphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    ctx->vgraphs.clear();
    for (auto & g : ctx->tgraphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    ctx->tgraphs.clear();
}

bool ensure_workspace(clip_ctx * ctx, size_t bytes) {
    if (ctx->ws.bytes >= bytes) return true;
    (void)hipStreamSynchronize(ctx->stream);
    drop_graphs(ctx);   // captured graphs hold pointers into the old workspace
    if (ctx->owner) {   // ... and so do the owner's, for the half-batches this sibling carries (vision_forward_launch)
        (void)hipStreamSynchronize(ctx->owner->stream);
        drop_graphs(ctx->owner);
    }
    if (ctx->ws.base) (void)hipFree(ctx->ws.base);
    ctx->ws.base = nullptr;
    ctx->ws.bytes = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    if (hipMalloc(&ctx->ws.base, want) != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "clip (hip): cannot allocate %zu MB of workspace\n", want >> 20);
        return false;
    }
    ctx->ws.bytes = want;
    return true;
}

bool ensure_pinned(clip_ctx * ctx, size_t bytes) {
    if (ctx->pinned_bytes >= bytes) return true;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    if (hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    ctx->pinned_bytes = bytes;
    return true;
}

bool ensure_io(clip_ctx * ctx, size_t in_bytes, size_t out_bytes) {
    auto grow = [&](void *& p, size_t & have, size_t want) {
        if (have >= want) return true;
        (void)hipStreamSynchronize(ctx->stream);
        if (p) (void)hipFree(p);
        p = nullptr;
        have = 0;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        have = want;
        return true;
    };
    return grow(ctx->io_in, ctx->io_in_bytes, in_bytes) && grow(ctx->io_out, ctx->io_out_bytes, out_bytes);
}

void prof_collect(clip_ctx * ctx) {
    if (ctx->pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto & p : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ProfEntry & e = ctx->prof[p.tag];
            e.ms += ms;
            e.launches += 1;
            e.flops += p.flops;
            e.bytes += p.bytes;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    ctx->pending.clear();
}

// ---------------------------------------------------------------------------------------------
static bool vision_forward_one(clip_ctx * ctx, const float * d_imgs, int B, float * d_out, bool normalize);

// One forward of a few dozen images leaves a good part of the chip idle (r03: three batch-32 requests in flight 50.6 k img/s against
// 30.7 k for one): such a call runs as two half-batches on two streams — this context and its weight-sharing sibling (own workspace, own split-K
// buffers) — forked and joined with events, so that the two dependent launch chains fill each other's gaps.  Rows are independent of
// each other, so each half is exactly the forward of those images at that batch size (deterministic; same kernels as a call of n / 2).
static bool vision_forward_launch(clip_ctx * ctx, const float * d_imgs, int B, float * d_out, bool normalize) {
    bool split = !ctx->weights_borrowed && !ctx->sibling_busy && !ctx->profiling && ctx->has_vision_encoder && B >= 2;
    if (split) {
        if (ctx->split_max >= 0) split = B >= ctx->split_min && B <= ctx->split_max;           // CLIP_AMD_SPLIT
        else {
            // measured (profiles/r04_batch_split_sweep.txt, one forward vs two halves, hipGraph replay where it applies): ViT-B/32 q4_0
            // +6.4 / +4.1 / +2.6 / +6.3 / +5.9 % at 16 / 24 / 32 / 48 / 64 images (800-3200 token rows), -3 ... -5 % at 4 / 8 / 96;
            // ViT-L/14 f16 +23 / +11 / +5 % at 8 / 16 / 32 images (2056-8224 rows), -2 % at 4.  3 and 4 parts are slower everywhere.
            const auto & hp = ctx->vision_hparams;
            const int G = hp.image_size / hp.patch_size;
            const long rows = (long)B * (G * G + 1);
            // With the text tower of the same step on a second stream (bench.py's two-tower cells) a third chain only helps the wide
            // model: 32 + 32 ViT-B/32 59.9 k -> 58.3 k emb/s (-2.7 %, as much as 32 images alone gain), 32 + 32 ViT-L/14 7.98 k -> 8.38 k (+5 %).
            // So the narrow models split only where the gain is ~6 % (48-64 images), the wide ones from 8 images.
            split = hp.hidden_size >= 1024 ? (rows >= 2000 && rows <= 8400) : (rows >= 2000 && rows <= 3300);
        }
    }
    int ways = split ? std::min(ctx->split_ways, B) : 1;
    clip_ctx * part_ctx[4] = {ctx, nullptr, nullptr, nullptr};
    if (split) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(ctx->stream, &cs);
        for (int w = 1; w < ways; w++) {
            clip_ctx * have = w == 1 ? ctx->sibling : ctx->more_siblings[w - 2];
            part_ctx[w] = have ? have : (cs == hipStreamCaptureStatusNone ? sibling_context(ctx, w - 1) : nullptr);   // (created outside captures only)
            if (!part_ctx[w]) { ways = 1; break; }
        }
    }
    ctx->last_launch_split = ways >= 2;
    if (ways < 2) return vision_forward_one(ctx, d_imgs, B, d_out, normalize);
    const size_t per = (size_t)ctx->vision_hparams.image_size * ctx->vision_hparams.image_size * 3;
    const int proj = ctx->vision_hparams.projection_dim;
    if (hipEventRecord(ctx->ev_fork, ctx->stream) != hipSuccess) return false;
    bool ok = true;
    int b0 = 0;
    for (int w = 0; w < ways; w++) {          // contiguous parts, sizes differing by at most one (larger first)
        const int n = B / ways + (w < B % ways ? 1 : 0);
        clip_ctx * c = part_ctx[w];
        c->input_f16 = ctx->input_f16;
        const float * src = ctx->input_f16 ? (const float *)((const half_t *)d_imgs + (size_t)b0 * per) : d_imgs + (size_t)b0 * per;
        if (w > 0 && hipStreamWaitEvent(c->stream, ctx->ev_fork, 0) != hipSuccess) return false;
        ok = vision_forward_one(c, src, n, d_out + (size_t)b0 * proj, normalize) && ok;
        b0 += n;
    }
    for (int w = 1; w < ways; w++) {
        hipEvent_t ej = w == 1 ? ctx->ev_join : ctx->ev_join_more[w - 2];
        if (hipEventRecord(ej, part_ctx[w]->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, ej, 0) != hipSuccess) return false;
    }
    return ok;
}

// A capture that a foreign thread invalidated (the application's own hipMalloc / hipFree on another thread: model.h RelaxCapture covers
// this library's threads only) can leave the stream refusing every later launch on this ROCm ("operation not permitted when stream is
// capturing" after hipStreamEndCapture has returned the invalidation).  The context and its siblings then move to fresh streams of their
// own, behind a device synchronisation (work queued before the capture keeps its order).  False: a stream set by the caller
// (clip_amd_set_stream) is still unusable — that one is the caller's to replace.
bool renew_streams_after_failed_capture(clip_ctx * ctx) {
    bool ok = true, synced = false;
    auto one = [&](clip_ctx * c) {
        if (!c) return;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const hipError_t qe = hipStreamIsCapturing(c->stream, &st);
        (void)hipGetLastError();
        if (qe == hipSuccess && st == hipStreamCaptureStatusNone) return;
        if (c->stream != c->own_stream) { ok = false; return; }
        if (!synced) { (void)hipDeviceSynchronize(); (void)hipGetLastError(); synced = true; }
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
        (void)hipStreamDestroy(c->own_stream);
        (void)hipGetLastError();
        c->own_stream = c->stream = fresh;
        if (c->verbosity >= 1 || (c->owner && c->owner->verbosity >= 1)) fprintf(stderr, "clip (hip): stream replaced after an invalidated capture\n");
    };
    one(ctx);
    one(ctx->sibling);
    for (clip_ctx * ms : ctx->more_siblings) one(ms);
    return ok;
}

namespace {
// CLIP_AMD_TEST_BREAK_CAPTURE=1 (tests of the recovery above): an allocation in the thread-local checking mode inside the capture that has
// just begun — what HIP forbids — so that the capture is invalidated the way a colliding thread would invalidate it
void break_capture_for_test() {
    static const bool on = [] { const char * e = getenv("CLIP_AMD_TEST_BREAK_CAPTURE"); return e && e[0] == '1'; }();
    if (!on) return;
    hipStreamCaptureMode m = hipStreamCaptureModeThreadLocal;
    (void)hipThreadExchangeStreamCaptureMode(&m);
    void * q = nullptr;
    if (hipMalloc(&q, 256) == hipSuccess && q) (void)hipFree(q);
    (void)hipThreadExchangeStreamCaptureMode(&m);
    (void)hipGetLastError();
}
}  // namespace

bool vision_forward_device(clip_ctx * ctx, const float * d_imgs, int B, float * d_out, bool normalize) {
    if (!check_device(ctx, "clip_image_batch_encode")) return false;
    if (!ctx->graphs_enabled || ctx->profiling || B <= 0 || B > 32 || !ctx->has_vision_encoder || guard_mode())
        return vision_forward_launch(ctx, d_imgs, B, d_out, normalize);   // big batches are GPU-bound: no graph needed
    // (looked up again after every vision_forward_launch: growing a sibling's workspace drops this context's graphs — ensure_workspace).
    // The sibling's state is part of the key: the split decision differs with it, so each form has its own entry — first sighting
    // eager, second captured — and a caller that alternates pair calls and plain calls still reaches its captures.
    auto find = [&]() -> clip_ctx::GraphEntry * {
        for (auto & g : ctx->vgraphs)
            if (g.B == B && g.in == d_imgs && g.out == d_out && g.norm == normalize && g.in_f16 == ctx->input_f16 && g.busy_seen == ctx->sibling_busy) return &g;
        return nullptr;
    };
    clip_ctx::GraphEntry * e = find();
    if (e && e->exec) {
        // a graph captured with the batch split over the sibling context holds kernel nodes that write the sibling's workspace: while the
        // sibling carries the text tower of a pair call (host_pipeline.cpp) it must not be replayed — the eager path honours sibling_busy
        if (e->split && ctx->sibling_busy) return vision_forward_launch(ctx, d_imgs, B, d_out, normalize);
        return hipGraphLaunch(e->exec, ctx->stream) == hipSuccess;
    }
    if (!e) {   // first sighting: run eagerly (allocates the workspace, sets kernel attributes)
        if (ctx->vgraphs.size() >= 16) drop_graphs(ctx);
        ctx->vgraphs.push_back({B, d_imgs, d_out, normalize, ctx->input_f16, 1, nullptr, nullptr, false, ctx->sibling_busy});
        return vision_forward_launch(ctx, d_imgs, B, d_out, normalize);
    }
    // second sighting: capture
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return vision_forward_launch(ctx, d_imgs, B, d_out, normalize);
    }
    break_capture_for_test();
    const bool ok = vision_forward_launch(ctx, d_imgs, B, d_out, normalize);
    const bool captured_split = ctx->last_launch_split;
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(ctx->stream, &graph);
    e = find();              // (the vector may have been cleared under the launch: never keep the pointer across it)
    if (!ok || ce != hipSuccess || !graph || !e) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        ctx->graphs_enabled = false;   // capture not possible here: stay eager from now on
        if (!renew_streams_after_failed_capture(ctx)) { fprintf(stderr, "clip_image_batch_encode: the caller's stream is left in an invalidated capture\n"); return false; }
        return vision_forward_launch(ctx, d_imgs, B, d_out, normalize);
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(graph);
        ctx->graphs_enabled = false;
        return vision_forward_launch(ctx, d_imgs, B, d_out, normalize);
    }
    e->graph = graph;
    e->exec = exec;
    e->split = captured_split;
    return hipGraphLaunch(exec, ctx->stream) == hipSuccess;
}

// The vision tower of one workspace chunk in three stages, so that the host pipeline can run the first one per 32-image copy piece
// while later pieces are still crossing PCIe (host_pipeline.cpp) and the layers ONCE on the whole chunk:
//   vision_stage_begin : workspace for Bc images
//   vision_stage_patch : images [i0, i0 + n) of the chunk: HWC -> im2col -> patch GEMM (+ position embedding) + class-token rows
//   vision_stage_finish: pre-LN, the layers, pool + post-LN + projection + L2 for all Bc images
bool vision_stage_begin(clip_ctx * ctx, int Bc, VisionStage & st) {
    const auto & hp = ctx->vision_hparams;
    const DevTower & V = ctx->vision;
    const int S = hp.image_size, P = hp.patch_size, G = S / P, Np = G * G, T = Np + 1;
    const int h = hp.hidden_size, ff = hp.n_intermediate, proj = hp.projection_dim;
    const int rows = Bc * T;
    st.Bc = Bc;
    st.st_stride = (rows + 63) & ~63;                      // LayerNorm-fold statistics: [<= h / 16 slots][st_stride rows] float2
    auto carve = [&](Carver & c) {
        st.stats = c.take<float2>((size_t)(h / 16) * st.st_stride);
        st.mu = c.take<float>((size_t)2 * st.st_stride);      // centring offsets of the folded LayerNorms (two buffers, ping-pong)
        st.x = c.take<float>((size_t)rows * h);
        const size_t aw = acts_f32(ctx, V) ? 2 : 1;                // f32 file: the activation buffers hold float rows
        st.xn = c.take<half_t>(aw * rows * h);
        st.qkv = c.take<half_t>(aw * rows * 3 * h);
        st.att = c.take<half_t>(aw * rows * h);
        st.mid = c.take<half_t>(aw * rows * ff);
        st.col = c.take<half_t>((size_t)Bc * Np * V.patch.Kpad);
        st.pooled = c.take<half_t>(aw * Bc * h);
        st.emb = c.take<float>((size_t)Bc * proj);
        st.xp = c.take<float>((size_t)Bc * h);                // pooled rows of the last layer (pooled_tail)
        st.ap = c.take<half_t>(aw * Bc * h);
        st.xnp = c.take<half_t>(aw * Bc * h);
        st.midp = c.take<half_t>(aw * Bc * ff);
    };
    Carver sizer(nullptr);
    carve(sizer);
    if (!ensure_workspace(ctx, sizer.off + 4096)) return false;
    Carver c(ctx->ws.base, &ctx->guard_gaps);
    carve(c);
    return guard_arm(ctx);
}

// imgs: the n images themselves (f32, or fp16 when ctx->input_f16), i0: their index inside the chunk
bool vision_stage_patch(clip_ctx * ctx, const VisionStage & st, const void * imgs, int i0, int n) {
    const bool piece = n < st.Bc;                             // part of a chunk: no split-K, so that the rows do not depend on the piece size
    const auto & hp = ctx->vision_hparams;
    const DevTower & V = ctx->vision;
    const int S = hp.image_size, P = hp.patch_size, G = S / P, Np = G * G, T = Np + 1, h = hp.hidden_size;
    hipStream_t s = ctx->stream;
    half_t * col = st.col + (size_t)i0 * Np * V.patch.Kpad;
    float * x = st.x + (size_t)i0 * T * h;
    // patch embedding = im2col + GEMM (ggml_conv_2d, clip.cpp:1309-1312); epilogue scatters to token rows + pos
    {
        ProfScope ps(ctx, "im2col", n * Np, V.patch.Kpad, 0, 0, (double)n * S * S * 3 * (ctx->input_f16 ? 2 : 4) + (double)n * Np * V.patch.Kpad * 2);
        launch_im2col(imgs, ctx->input_f16, col, n, S, P, V.patch.Kpad, s);
    }
    GemmParams pp;
    pp.A = col; pp.lda = V.patch.Kpad; pp.M = n * Np; pp.W = V.patch; pp.out = x; pp.ldc = h;
    pp.Np = Np; pp.T = T; pp.pos = V.pos; pp.no_splitk = piece;
    gemm(ctx, "gemm_patch", pp, EPI_PATCH_F32);
    // (the class-token rows — class_embd + pos[0], clip.cpp:1315-1331 — are produced by the pre-LN launch of vision_stage_finish)
    return true;
}

bool vision_stage_finish(clip_ctx * ctx, const VisionStage & st, float * d_out, bool normalize) {
    const auto & hp = ctx->vision_hparams;
    const DevTower & V = ctx->vision;
    const int S = hp.image_size, P = hp.patch_size, G = S / P, Np = G * G, T = Np + 1;
    const int h = hp.hidden_size, ff = hp.n_intermediate, nh = hp.n_head, proj = hp.projection_dim;
    hipStream_t s = ctx->stream;
    const int Bc = st.Bc, rows = Bc * T;
    float * x = st.x;
    const bool skinny = layers_fit_skinny(V, rows, h, ff);
    const bool fold = ctx->ln_fold && !V.layers.empty() && (ctx->ln_fold_force || fold_pays(V, rows, device_shared(ctx)));
    {
        ProfScope ps(ctx, "layernorm", rows, h, 0, 0, (double)rows * h * (fold ? 10 : 8));
        if (fold)   // class-token rows (:1315-1331) + pre-LN (:1334-1339) + entry of the folded chain: xn = fp16(x ln1_w[0]), whole-row statistics
            launch_layernorm_prep(x, h, V.pre_ln_w, V.pre_ln_b, hp.eps, rows, h, x, h, V.layers[0].ln1_w, st.xn, h, st.stats, s, ctx->ln_fold_centre ? st.mu : nullptr,
                                  V.class_embd, V.pos, T);
        else {
            launch_cls_rows(x, V.class_embd, V.pos, Bc, T, h, s);   // class token + pos[0] (clip.cpp:1315-1331)
            launch_layernorm(x, h, nullptr, 1, V.pre_ln_w, V.pre_ln_b, hp.eps, rows, h, nullptr, 0, x, h, s);
        }
    }
    // rows beyond the small-M path: the last layer's out-projection and FFN run on the Bc class-token rows only (pooled_tail)
    const bool prune = ctx->prune_last && !skinny && !V.layers.empty() && T > 1;
    if (skinny && fold) {
        if (!run_layers_skinny_fold(ctx, V, rows, h, nh, ff, hp.eps, Bc, T, nullptr, T, false, x, st.xn, st.qkv, st.att, st.mid, st.stats, st.st_stride, ctx->ln_fold_centre ? st.mu : nullptr)) return false;
    } else if (skinny) {
        launch_row_stats(x, h, rows, h, ctx->sk_stats, s);
        if (!run_layers_skinny(ctx, V, rows, h, nh, ff, hp.eps, Bc, T, nullptr, T, false, x, st.qkv, st.att, st.mid)) return false;
    } else if (fold) {
        if (!run_layers_fold(ctx, V, rows, h, nh, ff, hp.eps, Bc, T, nullptr, T, false, x, st.xn, st.qkv, st.att, st.mid, st.stats, st.st_stride, ctx->ln_fold_centre ? st.mu : nullptr, prune, resident_panels(ctx, V, 0, rows))) return false;
    } else if (!run_layers(ctx, V, rows, h, nh, ff, hp.eps, Bc, T, nullptr, T, false, x, st.xn, st.qkv, st.att, st.mid, prune, resident_panels(ctx, V, 0, rows))) return false;
    // CLS pool + post-LN (:1426-1438): LayerNorm with a strided row gather (row b*T)
    const bool f32 = acts_f32(ctx, V);
    if (prune) {
        if (!pooled_tail(ctx, V.layers.back(), Bc, h, ff, hp.eps, x, st.att, nullptr, T, st.xp, st.ap, st.xnp, st.midp)) return false;
        layernorm_act(f32, st.xp, h, nullptr, 1, V.post_ln_w, V.post_ln_b, hp.eps, Bc, h, st.pooled, s);
    } else {
        layernorm_act(f32, x, h, nullptr, T, V.post_ln_w, V.post_ln_b, hp.eps, Bc, h, st.pooled, s);
    }
    GemmParams pj;
    pj.A = st.pooled; pj.lda = h; pj.M = Bc; pj.W = V.proj; pj.out = st.emb; pj.ldc = proj; pj.act_f32 = f32;
    gemm(ctx, "gemm_proj", pj, EPI_F32);   // projection, no bias (:1443)
    launch_l2norm(st.emb, d_out, Bc, proj, normalize, s);  // (:1446-1455)
    return launch_ok("clip_image_batch_encode") && guard_check(ctx, "clip_image_batch_encode");
}

int vision_max_chunk(const clip_ctx * ctx) {
    const auto & hp = ctx->vision_hparams;
    const int S = hp.image_size, P = hp.patch_size, G = S / P, Np = G * G, T = Np + 1;
    const int h = hp.hidden_size, ff = hp.n_intermediate;
    // images are processed in chunks so that the workspace stays bounded (<= ~6 GB even for ViT-H)
    const size_t aw = acts_f32(ctx, ctx->vision) ? 2 : 1;
    const size_t per_img = (size_t)T * ((size_t)h * 4 + aw * ((size_t)h * 2 * 2 + (size_t)3 * h * 2 + (size_t)ff * 2)) + (size_t)Np * ctx->vision.patch.Kpad * 2;
    return (int)std::min<size_t>(1024, std::max<size_t>(1, ((size_t)6 << 30) / per_img));
}

static bool vision_forward_one(clip_ctx * ctx, const float * d_imgs, int B, float * d_out, bool normalize) {
    if (!ctx->has_vision_encoder) {
        printf("This gguf file seems to have no vision encoder\n");
        return false;
    }
    if (B <= 0) return true;
    const int S = ctx->vision_hparams.image_size, proj = ctx->vision_hparams.projection_dim;
    const int chunk = std::min(B, vision_max_chunk(ctx));
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        VisionStage st;
        if (!vision_stage_begin(ctx, Bc, st)) return false;
        // (host-pointer path: the staging buffer holds fp16 pixels, converted while packing — host_pipeline.cpp)
        const void * imgs = ctx->input_f16 ? (const void *)((const half_t *)d_imgs + (size_t)b0 * S * S * 3) : (const void *)(d_imgs + (size_t)b0 * S * S * 3);
        if (!vision_stage_patch(ctx, st, imgs, 0, Bc)) return false;
        if (!vision_stage_finish(ctx, st, d_out + (size_t)b0 * proj, normalize)) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
bool text_forward_device(clip_ctx * ctx, const int32_t * d_ids, const int32_t * h_offsets, int n_texts, float * d_out,
                         bool normalize) {
    if (!check_device(ctx, "clip_text_encode")) return false;
    if (!ctx->has_text_encoder) {
        printf("This GGUF file seems to have no text encoder\n");
        return false;
    }
    if (n_texts <= 0) return true;
    const auto & hp = ctx->text_hparams;
    const DevTower & Tw = ctx->text;
    const int h = hp.hidden_size, ff = hp.n_intermediate, nh = hp.n_head, proj = hp.projection_dim;
    hipStream_t s = ctx->stream;
    const int rows = h_offsets[n_texts] - h_offsets[0];
    int max_len = 0;
    for (int i = 0; i < n_texts; i++) {
        const int len = h_offsets[i + 1] - h_offsets[i];
        if (len <= 0 || len > hp.num_positions) {
            // the reference would index past position_embeddings here (clip.cpp:1054-1061); refuse instead
            fprintf(stderr, "clip_text_encode: %d tokens is outside [1, %d]\n", len, hp.num_positions);
            return false;
        }
        max_len = std::max(max_len, len);
    }
    const int st_stride = (rows + 63) & ~63;                   // LayerNorm-fold statistics: [<= h / 16 slots][st_stride rows] float2
    float2 * stats = nullptr;
    float * mu = nullptr;                                      // centring offsets of the folded LayerNorms: [2][st_stride]
    float * xp = nullptr;                                      // pooled rows of the last layer (pooled_tail)
    half_t * ap = nullptr, * xnp = nullptr, * midp = nullptr;
    auto carve = [&](Carver & c, float *& x, half_t *& xn, half_t *& qkv, half_t *& att, half_t *& mid, half_t *& pooled,
                     float *& emb, int *& seq, int *& last) {
        stats = c.take<float2>((size_t)(h / 16) * st_stride);
        mu = c.take<float>((size_t)2 * st_stride);
        x = c.take<float>((size_t)rows * h);
        const size_t aw = acts_f32(ctx, Tw) ? 2 : 1;               // f32 file: the activation buffers hold float rows
        xn = c.take<half_t>(aw * rows * h);
        qkv = c.take<half_t>(aw * rows * 3 * h);
        att = c.take<half_t>(aw * rows * h);
        mid = c.take<half_t>(aw * rows * ff);
        pooled = c.take<half_t>(aw * n_texts * h);
        emb = c.take<float>((size_t)n_texts * proj);
        seq = c.take<int>((size_t)n_texts + 1);
        last = c.take<int>((size_t)n_texts);
        xp = c.take<float>((size_t)n_texts * h);
        ap = c.take<half_t>(aw * n_texts * h);
        xnp = c.take<half_t>(aw * n_texts * h);
        midp = c.take<half_t>(aw * n_texts * ff);
    };
    float *x, *emb;
    half_t *xn, *qkv, *att, *mid, *pooled;
    int *seq, *last;
    Carver sizer(nullptr);
    carve(sizer, x, xn, qkv, att, mid, pooled, emb, seq, last);
    if (!ensure_workspace(ctx, sizer.off + 4096)) return false;
    Carver c(ctx->ws.base, &ctx->guard_gaps);
    carve(c, x, xn, qkv, att, mid, pooled, emb, seq, last);
    if (!guard_arm(ctx)) return false;
    {
        // sequence offsets + last-token rows (EOS, clip.cpp:1154-1155): written into a slot of a pinned ring and uploaded
        // asynchronously by a one-workgroup launch that reads the (device-mapped) slot — the host does not wait for the stream; a slot is
        // re-used 8 calls later, and only if its upload is somehow still pending then does the host spin on the stamp the launch leaves
        MetaRing & mr = ctx->meta;
        const int sl = mr.next;
        mr.next = (mr.next + 1) % MetaRing::SLOTS;
        const size_t n_ints = 2 * (size_t)n_texts + 1;
        if (!mr.done) {      // first text call of the context: the stamp array
            unsigned * d = nullptr;
            if (hipHostMalloc((void **)&d, MetaRing::SLOTS * sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
                memset(d, 0, MetaRing::SLOTS * sizeof(unsigned));
                if (hipHostGetDevicePointer((void **)&mr.done_dev, d, 0) == hipSuccess) mr.done = d;
                else { (void)hipGetLastError(); (void)hipHostFree(d); }
            } else (void)hipGetLastError();
        }
        // the slot was last used SLOTS calls ago: wait until that upload has read it (normally long done)
        if (mr.done && mr.expect[sl]) {
            // bounded: after ~1 ms of yields fall back to the stream itself — a failed launch, a faulted stream or a stream
            // under capture never writes the stamp, and the old event path reported exactly those cases as an error
            for (unsigned spins = 0; mr.done[sl] != mr.expect[sl]; spins++) {
                if (spins > 64) std::this_thread::yield();
                if (spins > 20000) {
                    const hipError_t qe = hipStreamSynchronize(s);
                    if (qe != hipSuccess || mr.done[sl] != mr.expect[sl]) {
                        fprintf(stderr, "clip_text_encode: the metadata upload of an earlier call never completed (%s)\n", hipGetErrorString(qe));
                        (void)hipGetLastError();
                        mr.expect[sl] = 0;
                        if (qe != hipSuccess) return false;
                    }
                    break;
                }
            }
        } else if (mr.busy[sl]) {
            (void)hipEventSynchronize(mr.ev[sl]);
        }
        if (mr.cap[sl] < n_ints) {
            if (mr.pin[sl]) (void)hipHostFree(mr.pin[sl]);
            mr.pin[sl] = nullptr;
            mr.dev[sl] = nullptr;
            mr.cap[sl] = 0;
            const size_t want = std::max<size_t>(n_ints, 1024);
            if (hipHostMalloc((void **)&mr.pin[sl], want * sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipHostGetDevicePointer((void **)&mr.dev[sl], mr.pin[sl], 0) != hipSuccess) { (void)hipGetLastError(); mr.dev[sl] = nullptr; }
            mr.cap[sl] = want;
        }
        int * hs = mr.pin[sl];
        for (int i = 0; i <= n_texts; i++) hs[i] = h_offsets[i] - h_offsets[0];
        for (int i = 0; i < n_texts; i++) hs[n_texts + 1 + i] = hs[i + 1] - 1;
        if (mr.done && mr.dev[sl]) {
            if (++mr.stamp == 0) mr.stamp = 1;
            mr.expect[sl] = mr.stamp;
            mr.busy[sl] = false;
            launch_meta_upload(mr.dev[sl], seq, last, n_texts, mr.done_dev + sl, mr.stamp, s);      // one launch reading the mapped slot
            if (hipGetLastError() != hipSuccess) {      // the launch did not happen: nobody will stamp this slot
                mr.expect[sl] = 0;
                fprintf(stderr, "clip_text_encode: metadata upload launch failed\n");
                return false;
            }
        } else {
            // fallback (the slot could not be mapped): two copies + an event.  seq and last are carved back to back but 256-byte aligned
            if (!mr.ev[sl] && hipEventCreateWithFlags(&mr.ev[sl], hipEventDisableTiming) != hipSuccess) return false;
            (void)hipMemcpyAsync(seq, hs, ((size_t)n_texts + 1) * 4, hipMemcpyHostToDevice, s);
            (void)hipMemcpyAsync(last, hs + n_texts + 1, (size_t)n_texts * 4, hipMemcpyHostToDevice, s);
            (void)hipEventRecord(mr.ev[sl], s);
            mr.expect[sl] = 0;
            mr.busy[sl] = true;
        }
    }
    auto launch_all = [&]() -> bool {
        const bool skinny = layers_fit_skinny(Tw, rows, h, ff);
        const bool fold = ctx->ln_fold && !Tw.layers.empty() && (ctx->ln_fold_force || fold_pays(Tw, rows, device_shared(ctx)));
        const bool prune = ctx->prune_last && !skinny && !Tw.layers.empty() && rows > n_texts;
        if (fold)   // embedding (:1059-1061) + entry of the folded chain: xn = fp16(x ln1_w[0]), whole-row statistics
            launch_text_embed(d_ids + h_offsets[0], seq, n_texts, rows, Tw.tok_raw, Tw.tok_type, Tw.pos, h, x, s, Tw.layers[0].ln1_w, xn, h, stats, ctx->ln_fold_centre ? mu : nullptr);
        else launch_text_embed(d_ids + h_offsets[0], seq, n_texts, rows, Tw.tok_raw, Tw.tok_type, Tw.pos, h, x, s);  // (:1059-1061)
        if (skinny && fold) {
            if (!run_layers_skinny_fold(ctx, Tw, rows, h, nh, ff, hp.eps, n_texts, 0, seq, max_len, true, x, xn, qkv, att, mid, stats, st_stride, ctx->ln_fold_centre ? mu : nullptr)) return false;
        } else if (skinny) {
            launch_row_stats(x, h, rows, h, ctx->sk_stats, s);
            if (!run_layers_skinny(ctx, Tw, rows, h, nh, ff, hp.eps, n_texts, 0, seq, max_len, true, x, qkv, att, mid)) return false;
        } else if (fold) {
            if (!run_layers_fold(ctx, Tw, rows, h, nh, ff, hp.eps, n_texts, 0, seq, max_len, true, x, xn, qkv, att, mid, stats, st_stride, ctx->ln_fold_centre ? mu : nullptr, prune, resident_panels(ctx, Tw, 1, rows))) return false;
        } else if (!run_layers(ctx, Tw, rows, h, nh, ff, hp.eps, n_texts, 0, seq, max_len, true, x, xn, qkv, att, mid, prune, resident_panels(ctx, Tw, 1, rows))) return false;
        // final LN on the pooled (last) row only — LayerNorm is row-wise, so LN-then-gather == gather-then-LN (:1146-1155)
        if (prune) {    // ... and so is everything behind the last layer's attention: out-projection + FFN on the n_texts last-token rows only
            if (!pooled_tail(ctx, Tw.layers.back(), n_texts, h, ff, hp.eps, x, att, last, 1, xp, ap, xnp, midp)) return false;
            layernorm_act(acts_f32(ctx, Tw), xp, h, nullptr, 1, Tw.post_ln_w, Tw.post_ln_b, hp.eps, n_texts, h, pooled, s);
        } else {
            layernorm_act(acts_f32(ctx, Tw), x, h, last, 1, Tw.post_ln_w, Tw.post_ln_b, hp.eps, n_texts, h, pooled, s);
        }
        GemmParams pj;
        pj.A = pooled; pj.lda = h; pj.M = n_texts; pj.W = Tw.proj; pj.out = emb; pj.ldc = proj; pj.act_f32 = acts_f32(ctx, Tw);
        gemm(ctx, "gemm_proj", pj, EPI_F32);     // (:1160)
        launch_l2norm(emb, d_out, n_texts, proj, normalize, s);  // (:1163-1166)
        return true;
    };
    // Small token counts are launch-bound (~90 dependent launches): capture the chain on the second sighting of a signature and
    // replay it afterwards.  The kernels read the sequence offsets from device memory (uploaded above), so a graph only depends on
    // (texts, token rows, attention key-tile bucket, pointers) — not on the individual lengths.
    const int nt_bucket = (max_len + 15) / 16;
    if (ctx->graphs_enabled && !ctx->profiling && rows <= 1024 && !guard_mode()) {
        clip_ctx::TextGraphEntry * e = nullptr;
        const void * ids_key = d_ids + h_offsets[0];
        for (auto & g : ctx->tgraphs)
            if (g.n_texts == n_texts && g.rows == rows && g.nt == nt_bucket && g.ids == ids_key && g.out == d_out && g.norm == normalize) { e = &g; break; }
        if (e && e->exec) return hipGraphLaunch(e->exec, s) == hipSuccess;
        if (!e) {
            if (ctx->tgraphs.size() >= 96) drop_graphs(ctx);
            ctx->tgraphs.push_back({n_texts, rows, nt_bucket, ids_key, d_out, normalize, 1, nullptr, nullptr});
        } else if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            break_capture_for_test();
            const bool ok = launch_all();
            hipGraph_t graph = nullptr;
            const hipError_t ce = hipStreamEndCapture(s, &graph);
            hipGraphExec_t exec = nullptr;
            if (ok && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                e->graph = graph;
                e->exec = exec;
                return hipGraphLaunch(exec, s) == hipSuccess;
            }
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            ctx->graphs_enabled = false;      // capture not possible here: stay eager from now on
            if (!renew_streams_after_failed_capture(ctx)) { fprintf(stderr, "clip_text_encode: the caller's stream is left in an invalidated capture\n"); return false; }
            s = ctx->stream;
        } else {
            (void)hipGetLastError();
        }
    }
    if (!launch_all()) return false;
    return launch_ok("clip_text_encode") && guard_check(ctx, "clip_text_encode");
}

}  // namespace clipamd

#ifdef CLIPAMD_SK_TIMING
// the stamps of the last forward's small-M launches (16 x u64 per launch, launch order), restarting the launch counter
extern "C" int clip_amd_debug_read_sk_stamps(unsigned long long * out, int cap_launches) {
    using namespace clipamd;
    (void)hipDeviceSynchronize();
    const int n = g_sk_stamp_launch < cap_launches ? g_sk_stamp_launch : cap_launches;
    if (g_sk_stamps && n > 0) (void)hipMemcpy(out, g_sk_stamps, (size_t)n * 16 * 8, hipMemcpyDeviceToHost);
    g_sk_stamp_launch = 0;
    return n;
}
#endif
