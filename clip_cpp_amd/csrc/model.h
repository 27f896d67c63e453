// model.h — clip_ctx: host-side model state + HBM-resident weights + forward workspace.
//
// Mirrors what reference clip.cpp keeps in `struct clip_ctx` (clip.cpp:240-253), `clip_text_model`
// (:192-205), `clip_vision_model` (:207-224) and `clip_layer` (:164-190), re-laid-out for the GPU:
// weights are uploaded once at load (clip_model_load, reference :334-596) in the GEMM-friendly
// repacked format of kernels.h, q/k/v are fused into one [3h][h] weight, and the fixed-size arenas of
// reference :261-331 are replaced by a workspace sized from hparams x batch.
#pragma once

#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/clip.h"
#include "gguf.h"
#include "kernels.h"

namespace clipamd {

struct DevLayer {
    DevWeight qkv, o, ff1, ff2;
    const float *qkv_b = nullptr, *o_b = nullptr, *ff1_b = nullptr, *ff2_b = nullptr;
    const float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
    // LayerNorm fold (gemm_common.h): c_n = sum_k gamma_k W_nk and b'_n = sum_k beta_k W_nk + b_n of (LN1, q/k/v) and (LN2, FFN-up),
    // filled on the device right after the upload (k_fold.hip)
    const float *qkv_c = nullptr, *qkv_bf = nullptr, *ff1_c = nullptr, *ff1_bf = nullptr;
};

struct DevTower {
    std::vector<DevLayer> layers;
    const float * pos = nullptr;          // [T][h] f32 (dequantised once; ggml_get_rows semantics)
    const float *post_ln_w = nullptr, *post_ln_b = nullptr;
    DevWeight proj;
    // vision only
    DevWeight patch;                      // [h][3*P*P] f16, k = (c,ky,kx)
    const float * class_embd = nullptr;
    const float *pre_ln_w = nullptr, *pre_ln_b = nullptr;
    // text only
    const void * tok_raw = nullptr;       // token_embd in its ggml block layout
    int tok_type = 0;
};

struct ProfEntry {
    double ms = 0;
    int64_t launches = 0;
    double flops = 0;
    double bytes = 0;
};

struct Workspace {
    void * base = nullptr;
    size_t bytes = 0;
};

// Host -> device staging of the host-pointer API (host_pipeline.cpp): two pinned / device buffer pairs that alternate between
// chunks of a batch, a copy stream, and the events that order pack -> H2D -> forward.
struct HostPipe {
    hipStream_t copy_stream = nullptr;
    void * pin_in[2] = {nullptr, nullptr};
    size_t pin_in_bytes[2] = {0, 0};
    void * dev_in[2] = {nullptr, nullptr};
    size_t dev_in_bytes[2] = {0, 0};
    hipEvent_t ev_copied[2] = {nullptr, nullptr};     // last H2D out of pin_in[i]
    hipEvent_t ev_consumed[2] = {nullptr, nullptr};   // last forward reading dev_in[i]
    hipEvent_t ev_sub = nullptr;
    bool used[2] = {false, false};
    struct PackPool * pool = nullptr;                 // persistent packer threads (host_pipeline.cpp), created on first use
};

// Pinned ring for the small per-call metadata of the text tower (sequence offsets): the upload is asynchronous and the
// host never blocks on the stream (text_forward_device synchronised on every call).
struct MetaRing {
    static constexpr int SLOTS = 8;
    int * pin[SLOTS] = {nullptr};
    int * dev[SLOTS] = {nullptr};     // the slot's device-side address (hipHostGetDevicePointer)
    size_t cap[SLOTS] = {0};
    hipEvent_t ev[SLOTS] = {nullptr};     // (fallback path only: slots that could not be mapped)
    bool busy[SLOTS] = {false};
    int next = 0;
    // slot hand-back without HIP events: the upload kernel stamps done[slot] (pinned, mapped) when it has read the slot
    volatile unsigned * done = nullptr;   // [SLOTS], host view
    unsigned * done_dev = nullptr;        // device view of the same memory
    unsigned expect[SLOTS] = {0};         // the stamp the last upload from this slot will write (0 = never used)
    unsigned stamp = 0;
};

}  // namespace clipamd

// The opaque handle of the C API (reference clip.h:8).
struct clip_ctx {
    bool has_text_encoder = false;
    bool has_vision_encoder = false;
    bool use_gelu = false;
    int32_t ftype = 1;
    clip_text_hparams text_hparams{};
    clip_vision_hparams vision_hparams{};
    float image_mean[3] = {0, 0, 0};
    float image_std[3] = {1, 1, 1};

    // tokenizer (reference clip_vocab, clip.cpp:149-158)
    std::vector<std::string> id_to_token;
    std::unordered_map<std::string, int32_t> token_to_id;
    size_t max_token_len = 0;

    // device state
    int device = -1;                 // -1: host-only ctx (no encoders)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;    // stream in use (own or user-provided)
    void * weights_base = nullptr;   // one HBM allocation holding every tensor
    size_t weights_bytes = 0;
    bool weights_from_cache = false;      // the HBM image came from the repacked-weight cache (CLIP_AMD_WEIGHT_CACHE), not from a repack
    clipamd::DevTower vision, text;
    clipamd::Workspace ws;           // activations, grown on demand
    std::vector<size_t> guard_gaps;  // CLIP_AMD_GUARD=1: offsets of the canary blocks behind the buffers carved out of ws (forward.cpp)
    void * pinned = nullptr;         // pinned host staging
    size_t pinned_bytes = 0;
    void * io_in = nullptr;          // persistent device staging of the host-pointer API (stable pointers -> graph hits)
    size_t io_in_bytes = 0;
    void * io_out = nullptr;
    size_t io_out_bytes = 0;
    bool input_f16 = false;          // the vision forward's input pointer holds fp16 pixels (set by the host pipeline around its calls)
    clipamd::HostPipe pipe;          // pinned double-buffered staging of clip_image_batch_encode (host_pipeline.cpp)
    clipamd::MetaRing meta;          // pinned ring for the text tower's per-call offsets
    void * multi = nullptr;          // clipamd::MultiCtx* when loaded with clip_amd_model_load_multi (replicas on the other devices)
    // Sibling context on the same device SHARING this context's weight image (load.cpp sibling_context): own stream, activation
    // workspace and split-K buffers, no second copy of the weights.  Carries the second half of a mid-size batch (forward.cpp: one
    // forward of 8-64 images leaves ~40 % of the chip idle — two half-batches on two streams fill it) and the text tower of the
    // two-tower multi-GPU call.  Created on first use, freed with the owner.
    clip_ctx * sibling = nullptr;
    clip_ctx * more_siblings[2] = {nullptr, nullptr};     // (3- / 4-way splits: CLIP_AMD_SPLIT=min,max,ways)
    hipEvent_t ev_join_more[2] = {nullptr, nullptr};
    int split_ways = 2;
    bool weights_borrowed = false;   // this IS a sibling: weights_base belongs to the owner
    clip_ctx * owner = nullptr;      // (sibling only) the context whose captured graphs hold pointers into this one's workspace
    bool last_launch_split = false;  // vision_forward_launch: the last call ran as parts on this context and its sibling(s)
    bool f32_acts = true;            // f32 GGUF files: f32 activations between the kernels (forward.cpp acts_f32); CLIP_AMD_F32_ACTS=0 at load: the fp16 form of round 5
    bool device_shared = false;      // clip_amd_set_device_shared: the caller runs other work on this device concurrently (kernels.h GemmParams::shared_device)
    bool sibling_busy = false;       // the sibling is carrying something else right now (the text tower of a two-tower multi call): no batch split
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int split_min = -1, split_max = -1;    // images per call for which the forward is split over two contexts; -1 = the measured default rule by token rows (forward.cpp); CLIP_AMD_SPLIT=min,max[,ways] overrides, 0,0 = off
    hipEvent_t ev_stream_switch = nullptr;   // orders a new stream behind the work queued on the previous one (clip_amd_set_stream)
    void * pre_buf = nullptr;        // device blob of the GPU preprocessing path (descriptors, taps, raw pixels, row buffer)
    size_t pre_bytes = 0;
    // ... and its ring of staging slots for calls of more than 128 images (clip_amd_image_batch_encode_u8): pieces of <= 128 images; while
    // piece p is copied / preprocessed (and the previous chunk encoded), the host fills the next slot and the copy stream ships it (preprocess.cpp)
    struct PreSlot { void * pin = nullptr; size_t pin_bytes = 0; void * dev = nullptr; size_t dev_bytes = 0; hipEvent_t ev_h2d = nullptr, ev_done = nullptr; bool used = false; };
    static constexpr int PRE_SLOTS = 4;       // = staging pieces per forward chunk: a whole chunk can be staged under the previous chunk's forward
    PreSlot pre_slot[PRE_SLOTS];
    hipStream_t pre_copy_stream = nullptr;
    // split-K workspace of the GEMM (kernels.h GemmParams::sk_*): partial tiles + per-tile ticket counters (kept zero)
    float * sk_ws = nullptr;
    size_t sk_ws_floats = 0;
    unsigned * sk_cnt = nullptr;
    int sk_cnt_n = 0;
    // LayerNorm partial statistics of the small-M path (k_skinny.hip): two [128 slots][128 rows] float2 buffers, written by the
    // residual epilogues and read by the LayerNorm-fused projections of the next sub-layer
    float2 * sk_stats = nullptr;
    bool ln_fold_force = false;      // CLIP_AMD_LNFOLD=2
    bool prune_last = true;          // the last layer's out-projection + FFN run on the pooled rows only (forward.cpp pooled_tail; CLIP_AMD_PRUNE_LAST=0: every row, for A/B)
    bool ln_fold_centre = true;      // the folded operand is built about the row's mean at the previous LayerNorm (kernels.h GemmParams::xg_mu; CLIP_AMD_LNFOLD_CENTRE=0: the r03 form, for A/B)
    bool ln_fold = true;             // LayerNorm folded into the GEMM epilogues for > 64 rows (CLIP_AMD_LNFOLD=0: the two-launch form, for A/B)
    // fp16 panels of one layer's block-quantised weights for the large-M GEMM (k_gemm8.hip); grown on demand, re-filled per layer
    clipamd::half_t * w16_panel = nullptr;
    size_t w16_panel_halfs = 0;
    // RESIDENT fp16 panels of block-quantised weights (forward.cpp resident_panels; round 4), per tower ([0] vision, [1] text): built on the
    // first batch that wants them, entry 4 * layer + {0 q/k/v, 1 out, 2 FFN-up, 3 FFN-down}, null = multiply the quantised planes.
    clipamd::half_t * res_panel_buf[2] = {nullptr, nullptr};
    std::vector<const clipamd::half_t *> res_panels[2];
    unsigned res_panel_mask[2] = {0, 0};      // which of the four weights the table holds (bit = index above)
    bool resident_panels_on = true;           // CLIP_AMD_RESIDENT_PANELS=0: per-layer dequantisation launches / fused kernels only (A/B)

    // profiling (HIP events on the ctx stream)
    bool profiling = false;
    struct PendingEvent { hipEvent_t a, b; std::string tag; double flops, bytes; };
    std::vector<PendingEvent> pending;
    std::map<std::string, clipamd::ProfEntry> prof;

    // hipGraph cache for the vision forward (launch-bound at small batch: ~90 launches per pass).  A pass is captured the
    // second time the same (batch, input pointer, output pointer, normalize) signature is seen and replayed afterwards.
    // split: the capture holds kernel nodes on the sibling context (batch split over two streams) — such a graph must not be replayed
    // while the sibling carries something else (sibling_busy); busy_seen: sibling_busy at the last eager sighting — a capture happens
    // only behind an eager run that took the same split decision, so every workspace the captured form needs already has its size
    struct GraphEntry { int B; const void * in; void * out; bool norm; bool in_f16; int seen; hipGraph_t graph; hipGraphExec_t exec; bool split; bool busy_seen; };
    std::vector<GraphEntry> vgraphs;
    // ... and for the text forward at small token counts (single texts, short label lists): keyed on (texts, token rows,
    // attention key-tile bucket, pointers); the per-call sequence offsets live in device memory and are uploaded before the replay
    struct TextGraphEntry { int n_texts, rows, nt; const void * ids; void * out; bool norm; int seen; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<TextGraphEntry> tgraphs;
    bool graphs_enabled = true;      // CLIP_AMD_GRAPHS=0 disables

    int verbosity = 0;
    std::string path;
};

namespace clipamd {

// HIP checks "potentially unsafe" calls (hipMalloc, hipFree, hipMemset, stream / event creation, synchronous copies ...) against stream
// captures in flight and, with a thread's default mode, against captures begun by OTHER threads too: the call fails and the capture is
// invalidated (seen on ROCm 7.0 under scripts/fuzz/run_gpu.sh: one serving thread capturing a small text batch while another grew a
// workspace).  This library captures short launch chains that contain no such call, so every entry point (and every worker thread)
// that may allocate runs in the relaxed mode for its duration and hands the thread back as it found it.
struct RelaxCapture {
    hipStreamCaptureMode prev = hipStreamCaptureModeRelaxed;
    bool armed = false;
    RelaxCapture() {
        static const bool off = [] { const char * e = getenv("CLIP_AMD_NO_RELAX_CAPTURE"); return e && e[0] == '1'; }();    // A/B aid for tests of this guard
        if (off) return;
        armed = hipThreadExchangeStreamCaptureMode(&prev) == hipSuccess;
        if (!armed) (void)hipGetLastError();
    }
    ~RelaxCapture() { if (armed) (void)hipThreadExchangeStreamCaptureMode(&prev); }
    RelaxCapture(const RelaxCapture &) = delete;
    RelaxCapture & operator=(const RelaxCapture &) = delete;
};

// load.cpp
clip_ctx * load_model(const char * fname, int verbosity, int device);
clip_ctx * sibling_context(clip_ctx * owner, int index = 0);     // owner->sibling (index 0) / more_siblings[index - 1], created on first use (nullptr on failure)
void free_model(clip_ctx * ctx);
bool repack_for_test(int type, const void * w_raw, int64_t N, int64_t K, DevWeight & W, void ** dev_base);

// forward.cpp  (device pointers; asynchronous on ctx->stream)
bool vision_forward_device(clip_ctx * ctx, const float * d_imgs, int B, float * d_out, bool normalize);
bool text_forward_device(clip_ctx * ctx, const int32_t * d_ids, const int32_t * h_offsets, int n_texts, float * d_out,
                         bool normalize);
// the vision tower of one workspace chunk in stages (forward.cpp): the host pipeline runs the patch stage per copy piece
struct VisionStage {
    int Bc = 0, st_stride = 0;
    float * x = nullptr, * emb = nullptr;
    half_t * xn = nullptr, * qkv = nullptr, * att = nullptr, * mid = nullptr, * col = nullptr, * pooled = nullptr;
    float2 * stats = nullptr;
    float * mu = nullptr;
    float * xp = nullptr;            // pooled rows of the last layer (forward.cpp pooled_tail)
    half_t * ap = nullptr, * xnp = nullptr, * midp = nullptr;
};
bool vision_stage_begin(clip_ctx * ctx, int Bc, VisionStage & st);
bool vision_stage_patch(clip_ctx * ctx, const VisionStage & st, const void * imgs, int i0, int n);
bool vision_stage_finish(clip_ctx * ctx, const VisionStage & st, float * d_out, bool normalize);
int vision_max_chunk(const clip_ctx * ctx);
bool ensure_workspace(clip_ctx * ctx, size_t bytes);
bool ensure_pinned(clip_ctx * ctx, size_t bytes);
bool ensure_io(clip_ctx * ctx, size_t in_bytes, size_t out_bytes);
void prof_collect(clip_ctx * ctx);
void drop_graphs(clip_ctx * ctx);
bool renew_streams_after_failed_capture(clip_ctx * ctx);

// host_pipeline.cpp — pinned, threaded host -> device staging + the multi-GPU form of clip_image_batch_encode
bool encode_images_from_host(clip_ctx * ctx, const clip_image_f32 * imgs, int n, float * d_out, bool normalize, int n_threads);
void free_host_pipe(clip_ctx * ctx);
void multi_shard(int total, int n_dev, int g, int * lo, int * hi, int * per_dev);
clip_ctx * multi_load(const char * fname, int verbosity, int n_devices);
void multi_free(clip_ctx * primary);
int multi_device_count(const clip_ctx * primary);
bool multi_image_batch_encode(clip_ctx * primary, const clip_image_f32 * imgs, int B, float * vec, bool normalize, int n_threads);
// generic sharded call: run(g, ctx_g, lo, hi, d_send) queues the work of items [lo, hi) on replica g; see host_pipeline.cpp
bool multi_run(clip_ctx * primary, int total, int proj, float * vec, const char * who,
               const std::function<bool(int, clip_ctx *, int, int, float *)> & run);
bool multi_run_pair(clip_ctx * primary, int n_img, int n_txt, int proj, float * vec_img, float * vec_txt, const char * who,
                    const std::function<bool(int, clip_ctx *, int, int, float *)> & run_img,
                    const std::function<bool(int, clip_ctx *, int, int, float *)> & run_txt);
clip_ctx * multi_replica(const clip_ctx * primary, int g);
const float * multi_gathered(const clip_ctx * primary, int g);

// host pieces
bool tokenize_text(const clip_ctx * ctx, const char * text, std::vector<int32_t> & out);            // tokenizer.cpp
bool preprocess_image(const clip_ctx * ctx, const clip_image_u8 * img, clip_image_f32 * res);      // preprocess.cpp
bool preprocess_batch_device(clip_ctx * ctx, const clip_image_u8 * imgs, int n, float * d_out, int slot = -1);   // preprocess.cpp + k_preproc.hip; slot 0 / 1: pipelined staging
void free_preprocess_slots(clip_ctx * ctx);
bool load_image_file(const char * fname, clip_image_u8 * img);                                      // image_io.cpp

// quant.cpp — host codecs for the ggml block formats
void dequantize_row(int type, const void * src, float * dst, int64_t k);
size_t quantize_rows(int type, const float * src, void * dst, int64_t nrows, int64_t k);
uint16_t f32_to_f16_bits(float x);
float f16_bits_to_f32(uint16_t h);

}  // namespace clipamd
