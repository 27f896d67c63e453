// load.cpp — clip_model_load for the MI355X build: GGUF -> host staging -> ONE HBM allocation.
//
// Reference behaviour being replaced: clip.cpp:334-596 (parse KV, read every tensor into a ggml
// arena, bind by name).  Here every linear weight is repacked ONCE into the block-column-major planes
// the GEMM kernel streams (kernels.h / k_gemm.hip), q/k/v are fused into a single [3h][h] weight,
// position embeddings are dequantised to f32 (what ggml_get_rows would do on every call), everything
// is laid out in a single host buffer and uploaded with one hipMemcpy.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <ctime>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <fcntl.h>
#include <cerrno>
#include <atomic>
#include <thread>
#include <functional>

#include "model.h"

namespace clipamd {

namespace {

// ---- repack of one ggml-layout weight tensor into the GEMM planes (layout: kernels.h, k_gemm.hip) ----
struct RepackPlan {
    DevWeight W;
    size_t bytes_qs = 0, bytes_qh = 0, bytes_dm = 0, bytes_w16 = 0;
    int blk_bytes = 0, o_m = -1, o_qh = -1, o_qs = 2;
    bool is8 = false;
};

bool plan_repack(int type, int64_t N, int64_t K, RepackPlan & pl, std::string & err) {
    DevWeight & W = pl.W;
    W.N = (int)N;
    W.K = (int)K;
    W.Npad = (int)((N + 127) / 128 * 128);
    W.Kpad = (int)((K + 63) / 64 * 64);
    if (type == GT_F32) {
        // f32 files keep their linear weights in f32 (round 5; until then they were rounded to fp16 here — narrower than the reference's
        // f32 vec_dot): multiplied on the exact-f32 MFMA by k_gemm_f32.hip
        W.wtype = W_F32;
        pl.bytes_w16 = (size_t)W.Npad * W.Kpad * 4;
        return true;
    }
    if (type == GT_F16) {
        W.wtype = W_F16;
        pl.bytes_w16 = (size_t)W.Npad * W.Kpad * 2;
        return true;
    }
    if (K % 32) { err = "quantised tensor with K % 32 != 0"; return false; }
    switch (type) {
    case GT_Q4_0: W.wtype = W_Q4_0; pl.blk_bytes = 18; break;
    case GT_Q4_1: W.wtype = W_Q4_1; pl.blk_bytes = 20; pl.o_m = 2; pl.o_qs = 4; break;
    case GT_Q5_0: W.wtype = W_Q5_0; pl.blk_bytes = 22; pl.o_qh = 2; pl.o_qs = 6; break;
    case GT_Q5_1: W.wtype = W_Q5_1; pl.blk_bytes = 24; pl.o_m = 2; pl.o_qh = 4; pl.o_qs = 8; break;
    case GT_Q8_0: W.wtype = W_Q8_0; pl.blk_bytes = 34; pl.is8 = true; break;
    default: err = "unsupported weight type " + std::to_string(type); return false;
    }
    const size_t nblk = (size_t)(W.Kpad / 32) * W.Npad;
    pl.bytes_qs = nblk * (pl.is8 ? 32 : 16);
    pl.bytes_qh = pl.o_qh >= 0 ? nblk * 4 : 0;
    pl.bytes_dm = nblk * (pl.o_m >= 0 ? 4 : 2);
    return true;
}

// src: N_each rows of K weights in ggml layout; destination planes are zero-initialised by the caller.
void repack_rows(const RepackPlan & pl, int type, const uint8_t * src, int64_t N_each, int64_t K, int64_t row_off, uint8_t * qs_out,
                 uint8_t * qh_out, uint8_t * dm_out, uint8_t * w16_out) {
    const DevWeight & W = pl.W;
    if (W.wtype == W_F32) {
        float * dst = (float *)w16_out;
        for (int64_t n = 0; n < N_each; n++) memcpy(dst + (row_off + n) * W.Kpad, src + (size_t)n * K * 4, (size_t)K * 4);
        return;
    }
    if (W.wtype == W_F16) {
        uint16_t * dst = (uint16_t *)w16_out;
        for (int64_t n = 0; n < N_each; n++) memcpy(dst + (row_off + n) * W.Kpad, src + (size_t)n * K * 2, (size_t)K * 2);
        return;
    }
    const int64_t nb_src = K / 32;
    const size_t rb = (size_t)nb_src * pl.blk_bytes;
    for (int64_t n = 0; n < N_each; n++) {
        const int64_t ng = row_off + n;
        for (int64_t kb = 0; kb < nb_src; kb++) {
            const uint8_t * blk = src + n * rb + kb * pl.blk_bytes;
            const size_t idx = (size_t)kb * W.Npad + ng;
            if (pl.o_m >= 0) {
                memcpy(dm_out + idx * 4, blk, 2);
                memcpy(dm_out + idx * 4 + 2, blk + pl.o_m, 2);
            } else {
                memcpy(dm_out + idx * 2, blk, 2);
            }
            if (pl.is8) {
                // word t holds elements 4t..4t+3 as bytes [e0,e2,e1,e3], each XOR 0x80 (k_gemm.hip dequant_block)
                uint8_t * o = qs_out + idx * 32;
                const uint8_t * q = blk + 2;
                for (int t = 0; t < 8; t++) {
                    o[4 * t + 0] = q[4 * t + 0] ^ 0x80;
                    o[4 * t + 1] = q[4 * t + 2] ^ 0x80;
                    o[4 * t + 2] = q[4 * t + 1] ^ 0x80;
                    o[4 * t + 3] = q[4 * t + 3] ^ 0x80;
                }
                continue;
            }
            const uint8_t * qs = blk + pl.o_qs;
            uint32_t qh = 0;
            if (pl.o_qh >= 0) memcpy(&qh, blk + pl.o_qh, 4);
            uint32_t words[4] = {0, 0, 0, 0}, qh2 = 0;
            for (int e = 0; e < 32; e++) {
                const uint32_t nib = e < 16 ? (qs[e] & 0x0Fu) : (qs[e - 16] >> 4);
                const int j = e >> 3, w = e & 7;                    // word j, element w within the word
                const int p = (w & 1) ? 4 + (w >> 1) : (w >> 1);    // nibble position: pairs (2s,2s+1) -> nibbles (s, s+4)
                words[j] |= nib << (4 * p);
                if (pl.o_qh >= 0) {
                    const uint32_t bit = (qh >> e) & 1u;
                    const int pi = 4 * j + (w >> 1);                // pair index
                    qh2 |= bit << ((w & 1) ? 16 + pi : pi);
                }
            }
            memcpy(qs_out + idx * 16, words, 16);
            if (pl.o_qh >= 0) memcpy(qh_out + idx * 4, &qh2, 4);
        }
    }
}

struct Stage {                       // host image of the device weight allocation
    std::vector<uint8_t> buf;
    size_t size = 0;                 // bytes allocated so far (== buf.size() unless plan_only)
    bool plan_only = false;          // lay the image out without materialising it (its content comes from the weight cache)
    size_t alloc(size_t bytes) {     // 256-byte aligned bump allocation (zero filled)
        size_t off = (size + 255) & ~(size_t)255;
        size = off + bytes;
        if (!plan_only) buf.resize(size, 0);
        return off;
    }
};

struct Fix {                         // pointer fix-ups applied after the upload
    const void ** slot;
    size_t off;
};

struct Loader {
    const GgufFile & g;
    Stage st;
    std::vector<Fix> fixes;
    std::string err;
    explicit Loader(const GgufFile & gg) : g(gg) {}

    const GgufTensorInfo * need(const std::string & name) {
        const GgufTensorInfo * t = g.tensor(name);
        if (!t && err.empty()) err = "unable to find tensor " + name;
        return t;
    }
    void fix(const void ** slot, size_t off) { fixes.push_back({slot, off}); }

    // 1-D f32 tensor (bias / LN / class_embd), optionally several concatenated
    bool vec_f32(const std::vector<std::string> & names, const float ** slot, int64_t expect_each) {
        std::vector<const GgufTensorInfo *> ts;
        for (auto & n : names) {
            const GgufTensorInfo * t = need(n);
            if (!t) return false;
            if (t->type != GT_F32 || t->ne[0] != expect_each || t->nrows() != 1) { err = "tensor " + n + ": expected f32[" + std::to_string(expect_each) + "]"; return false; }
            ts.push_back(t);
        }
        size_t off = st.alloc((size_t)expect_each * 4 * ts.size());
        if (!st.plan_only) for (size_t i = 0; i < ts.size(); i++) memcpy(&st.buf[off + i * expect_each * 4], ts[i]->data, (size_t)expect_each * 4);
        fix((const void **)slot, off);
        return true;
    }

    // n zero-filled floats the device fills after the upload (LayerNorm fold vectors, k_fold.hip)
    bool zeros_f32(const float ** slot, int64_t n) {
        fix((const void **)slot, st.alloc((size_t)n * 4));
        return true;
    }

    // 2-D embedding table dequantised to f32 [rows][k]
    bool table_f32(const std::string & name, const float ** slot, int64_t k, int64_t rows) {
        const GgufTensorInfo * t = need(name);
        if (!t) return false;
        if (t->ne[0] != k || t->nrows() != rows) { err = "tensor " + name + ": unexpected shape"; return false; }
        size_t off = st.alloc((size_t)rows * k * 4);
        const size_t rb = ggml_row_bytes(t->type, k);
        if (!st.plan_only) for (int64_t r = 0; r < rows; r++) dequantize_row(t->type, t->data + r * rb, (float *)&st.buf[off] + r * k, k);
        fix((const void **)slot, off);
        return true;
    }

    bool raw(const std::string & name, const void ** slot, int * type, int64_t k, int64_t rows) {
        const GgufTensorInfo * t = need(name);
        if (!t) return false;
        if (t->ne[0] != k || t->nrows() != rows) { err = "tensor " + name + ": unexpected shape"; return false; }
        size_t off = st.alloc(t->nbytes);
        if (!st.plan_only) memcpy(&st.buf[off], t->data, t->nbytes);
        *type = t->type;
        fix(slot, off);
        return true;
    }

    // Linear weight(s) [N_i][K] concatenated along N and repacked for the GEMM kernel.
    bool linear(const std::vector<std::string> & names, DevWeight & W, int64_t K, int64_t N_each, bool as_conv = false) {
        std::vector<const uint8_t *> srcs;
        int type = -1;
        for (auto & n : names) {
            const GgufTensorInfo * t = need(n);
            if (!t) return false;
            const int64_t tk = as_conv ? t->ne[0] * t->ne[1] * t->ne[2] : t->ne[0];
            const int64_t tn = as_conv ? t->ne[3] : t->nrows();
            if (tk != K || tn != N_each) { err = "tensor " + n + ": unexpected shape"; return false; }
            if (type >= 0 && t->type != type) { err = "tensor " + n + ": mixed weight types in one fused projection"; return false; }
            type = t->type;
            srcs.push_back(t->data);
        }
        RepackPlan plan;
        if (!plan_repack(type, N_each * (int64_t)srcs.size(), K, plan, err)) return false;
        W = plan.W;
        size_t off_qs = 0, off_qh = 0, off_dm = 0, off_w16 = 0;
        if (plan.bytes_w16) off_w16 = st.alloc(plan.bytes_w16);
        if (plan.bytes_qs) off_qs = st.alloc(plan.bytes_qs);
        if (plan.bytes_qh) off_qh = st.alloc(plan.bytes_qh);
        if (plan.bytes_dm) off_dm = st.alloc(plan.bytes_dm);
        if (!st.plan_only)
            for (size_t i = 0; i < srcs.size(); i++)
                repack_rows(plan, type, srcs[i], N_each, K, (int64_t)i * N_each, &st.buf[off_qs], &st.buf[off_qh], &st.buf[off_dm], &st.buf[off_w16]);
        if (plan.bytes_w16) fix(&W.w16, off_w16);
        if (plan.bytes_qs) fix(&W.qs, off_qs);
        if (plan.bytes_qh) fix(&W.qh, off_qh);
        if (plan.bytes_dm) fix(&W.dm, off_dm);
        return true;
    }

    bool layers(const char * prefix, int n_layer, int h, int ff, std::vector<DevLayer> & L) {
        // a layer is 16 tensors of the file: a block count the file cannot hold (corrupt / hostile metadata) is refused before anything is sized by it
        if (n_layer < 0 || (size_t)n_layer > g.tensors.size() / 16) {
            if (err.empty()) err = "block_count " + std::to_string(n_layer) + " exceeds the tensors in the file";
            return false;
        }
        L.resize(n_layer);
        char b[96];
        for (int i = 0; i < n_layer; i++) {
            auto nm = [&](const char * what, const char * suf) {
                snprintf(b, sizeof b, "%s.blk.%d.%s.%s", prefix, i, what, suf);
                return std::string(b);
            };
            DevLayer & l = L[i];
            if (!linear({nm("attn_q", "weight"), nm("attn_k", "weight"), nm("attn_v", "weight")}, l.qkv, h, h)) return false;
            if (!vec_f32({nm("attn_q", "bias"), nm("attn_k", "bias"), nm("attn_v", "bias")}, &l.qkv_b, h)) return false;
            if (!linear({nm("attn_out", "weight")}, l.o, h, h) || !vec_f32({nm("attn_out", "bias")}, &l.o_b, h)) return false;
            // (sic) "ffn_down" is the h->ff projection, "ffn_up" the ff->h one (reference clip.cpp:510-511,572-573)
            if (!linear({nm("ffn_down", "weight")}, l.ff1, h, ff) || !vec_f32({nm("ffn_down", "bias")}, &l.ff1_b, ff)) return false;
            if (!linear({nm("ffn_up", "weight")}, l.ff2, ff, h) || !vec_f32({nm("ffn_up", "bias")}, &l.ff2_b, h)) return false;
            if (!vec_f32({nm("ln1", "weight")}, &l.ln1_w, h) || !vec_f32({nm("ln1", "bias")}, &l.ln1_b, h)) return false;
            if (!vec_f32({nm("ln2", "weight")}, &l.ln2_w, h) || !vec_f32({nm("ln2", "bias")}, &l.ln2_b, h)) return false;
            zeros_f32(&l.qkv_c, 3 * (int64_t)h); zeros_f32(&l.qkv_bf, 3 * (int64_t)h);
            zeros_f32(&l.ff1_c, ff); zeros_f32(&l.ff1_bf, ff);
        }
        return true;
    }
};

// File "<dir>/<basename>.<key>.hbm" = 32-byte header {magic, version, key, image bytes} + the HBM image.
struct WeightCache {
    static constexpr uint32_t kVersion = 4;    // bump when the device layout of any tensor changes (2: LayerNorm fold vectors; 4: f32 tensors of f32 files stay f32)
    bool enabled = false, readable = false;
    std::string path;
    uint64_t key = 0, image_bytes = 0;
    struct Header { char magic[8]; uint32_t version, plan_hash; uint64_t key, image_bytes; };   // plan_hash: offsets of every tensor in the image
    uint32_t plan_hash = 0, file_plan_hash = 0;

    static uint64_t fnv(uint64_t h, const void * p, size_t n) {
        const uint8_t * b = (const uint8_t *)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
        return h;
    }
    // 8 bytes at a time (memcpy: tensor data is only 2-byte aligned in some files), multiply-xorshift: GB/s instead of FNV's byte loop
    static uint64_t mix_words(uint64_t h, const void * p, size_t n) {
        const uint8_t * b = (const uint8_t *)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            memcpy(&w, b + i, 8);
            h = (h ^ w) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
        }
        return i < n ? fnv(h, b + i, n - i) : h;
    }
    // a writer that died between mkstemp and rename leaves "<image>.tmp.XXXXXX" behind for good: temporaries of THIS image older than
    // ten minutes (no live writer takes that long) are unlinked when the cache is opened
    static void sweep_stale_temporaries(const char * dir, const std::string & prefix) {
        DIR * d = opendir(dir);
        if (!d) return;
        const time_t now = time(nullptr);
        while (const dirent * e = readdir(d)) {
            if (strncmp(e->d_name, prefix.c_str(), prefix.size()) != 0) continue;
            const std::string p = std::string(dir) + "/" + e->d_name;
            struct stat st;
            if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode) && now - st.st_mtime > 600) (void)remove(p.c_str());
        }
        closedir(d);
    }
    void init(const GgufFile & g, const char * fname) {
        const char * dir = getenv("CLIP_AMD_WEIGHT_CACHE");
        if (!dir || !dir[0] || !g.file_base()) return;
        // content identity: size, all metadata + tensor infos, the layout version, and the data of EVERY tensor — by default 16 blocks
        // of 4 KB spread evenly over each tensor (first and last block included: ~13 MB hashed for ViT-L/14, a few ms; ADVICE r2: the
        // first / last 256 KB of the file alone let a fine-tune or re-quantisation of inner layers collide), with
        // CLIP_AMD_WEIGHT_CACHE_FULLHASH=1 every byte (word-wise mix: the file's pages are all read once).
        uint64_t h = 1469598103934665603ull;
        const uint64_t sz = g.file_size(), ver = kVersion;
        h = fnv(h, &sz, 8);
        h = fnv(h, &ver, 8);
        const size_t meta = (size_t)std::min<uint64_t>(g.data_offset, sz);
        h = fnv(h, g.file_base(), meta);
        const char * fh = getenv("CLIP_AMD_WEIGHT_CACHE_FULLHASH");
        const bool full = fh && fh[0] == '1';
        for (const GgufTensorInfo & t : g.tensors) {
            if (!t.data || !t.nbytes) continue;
            if (full) {
                h = mix_words(h, t.data, t.nbytes);
                continue;
            }
            constexpr size_t BLK = 4096, NB = 16;
            if (t.nbytes <= BLK * NB) { h = mix_words(h, t.data, t.nbytes); continue; }
            for (size_t b = 0; b < NB; b++) {
                const size_t off = (size_t)((double)(t.nbytes - BLK) * (double)b / (double)(NB - 1)) & ~(size_t)7;
                h = mix_words(h, t.data + off, BLK);
            }
        }
        key = h;
        const char * base = strrchr(fname, '/');
        char hex[24];
        snprintf(hex, sizeof hex, "%016llx", (unsigned long long)key);
        path = std::string(dir) + "/" + (base ? base + 1 : fname) + "." + hex + ".hbm";
        enabled = true;
        sweep_stale_temporaries(dir, std::string(base ? base + 1 : fname) + "." + hex + ".hbm.tmp.");
        FILE * f = fopen(path.c_str(), "rb");
        if (!f) return;
        Header hd;
        struct stat st;
        if (fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, "CLAMDHBM", 8) == 0 && hd.version == kVersion && hd.key == key &&
            fstat(fileno(f), &st) == 0 && (uint64_t)st.st_size == sizeof hd + hd.image_bytes) {
            readable = true;
            image_bytes = hd.image_bytes;
            file_plan_hash = hd.plan_hash;
        }
        fclose(f);
    }
    bool read_image(std::vector<uint8_t> & buf, size_t bytes) const {
        FILE * f = fopen(path.c_str(), "rb");
        if (!f) return false;
        buf.resize(bytes);
        const bool ok = fseek(f, sizeof(Header), SEEK_SET) == 0 && fread(buf.data(), 1, bytes, f) == bytes;
        fclose(f);
        return ok;
    }
    void write_image(const std::vector<uint8_t> & buf) const {   // best effort; written to a temporary name and renamed into place
        if (!enabled) return;
        // one temporary file PER WRITER: clip_amd_model_load_multi loads its G replicas on concurrent threads of one process,
        // and a name built from the pid alone had them all truncate and fill the same inode while the first finisher renamed it into
        // place (a published image with zero-filled holes).  mkstemp gives every thread / process its own file; rename stays atomic.
        // The file is created with open(O_CREAT | O_EXCL, 0666): the KERNEL applies the process umask atomically, so a cache directory shared
        // between users / services stays shared, and no thread ever touches the umask (the mkstemp + umask(0) / umask(um) read-back
        // of round 4 could leave the process umask at 0 when two loader threads interleaved).  The suffix is per writer: pid, thread, a counter.
        static std::atomic<unsigned> seq{0};
        int fd = -1;
        std::string tmp;
        for (int attempt = 0; attempt < 8 && fd < 0; attempt++) {
            char suf[96];
            snprintf(suf, sizeof suf, ".tmp.%ld.%zx.%u", (long)getpid(), std::hash<std::thread::id>()(std::this_thread::get_id()), seq.fetch_add(1));
            tmp = path + suf;
            fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY | O_CLOEXEC, 0666);
            if (fd < 0 && errno != EEXIST) return;
        }
        if (fd < 0) return;
        FILE * f = fdopen(fd, "wb");
        if (!f) { (void)close(fd); (void)remove(tmp.c_str()); return; }
        Header hd;
        memcpy(hd.magic, "CLAMDHBM", 8);
        hd.version = kVersion; hd.plan_hash = plan_hash; hd.key = key; hd.image_bytes = buf.size();
        const bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
    }
};

bool kv_u32(const GgufFile & g, const std::string & key, int32_t & out, std::string & err) {
    uint32_t v;
    if (!g.get_u32(key, v)) { if (err.empty()) err = "key " + key + " not found in file"; return false; }
    out = (int32_t)v;
    return true;
}

// What the kernels support, checked at load time so that an unsupported model fails HERE with a clear message instead of loading
// and then producing garbage or failing at the first encode: LayerNorm holds a row in <= 8 x 256 floats; epilogues
// store 4 columns per lane; the attention kernel is instantiated for d_head in {32, 64, 80, 88, 96, 104} and <= 592 (d_head <= 64) or 288 keys.
std::string kernel_limits(const char * tower, int hidden, int n_head, int proj, int seq_len) {
    char buf[192];
    const int dh = hidden / n_head;
    if (hidden > 2048) { snprintf(buf, sizeof buf, "unsupported %s hparams: hidden_size %d > 2048", tower, hidden); return buf; }
    if (proj <= 0 || proj % 4) { snprintf(buf, sizeof buf, "unsupported %s hparams: projection_dim %d is not a multiple of 4", tower, proj); return buf; }
    if (dh != 32 && dh != 64 && dh != 80 && dh != 88 && dh != 96 && dh != 104) { snprintf(buf, sizeof buf, "unsupported %s hparams: head size %d (supported: 32, 64, 80, 88, 96, 104)", tower, dh); return buf; }
    const int max_len = dh <= 64 ? 592 : 288;
    if (seq_len > max_len) { snprintf(buf, sizeof buf, "unsupported %s hparams: %d tokens per sequence > %d (head size %d)", tower, seq_len, max_len, dh); return buf; }
    return std::string();
}

}  // namespace

namespace { bool alloc_runtime_buffers(clip_ctx * ctx); }

clip_ctx * load_model(const char * fname, int verbosity, int device) {
    GgufFile g;
    std::string err;
    if (!g.open(fname, err)) {
        fprintf(stderr, "clip_model_load: %s\n", err.c_str());
        return nullptr;
    }
    clip_ctx * ctx = new clip_ctx();
    ctx->verbosity = verbosity;
    ctx->path = fname;
    auto fail = [&](const std::string & why) -> clip_ctx * {
        fprintf(stderr, "clip_model_load: %s (%s)\n", why.c_str(), fname);
        free_model(ctx);
        return nullptr;
    };

    uint32_t ftype = 1;
    if (!g.get_u32("general.file_type", ftype)) return fail("key general.file_type not found in file");
    switch (ftype) {
    case 0: case 1: case 2: case 3: case 6: case 7: case 8: break;
    default: return fail("unrecognized file type " + std::to_string(ftype));
    }
    ctx->ftype = (int32_t)ftype;
    if (!g.get_bool("clip.has_text_encoder", ctx->has_text_encoder)) return fail("key clip.has_text_encoder not found in file");
    if (!g.get_bool("clip.has_vision_encoder", ctx->has_vision_encoder)) return fail("key clip.has_vision_encoder not found in file");
    if (!g.get_bool("clip.use_gelu", ctx->use_gelu)) return fail("key clip.use_gelu not found in file");

    if (verbosity >= 1) {
        const GgufValue * name = g.find("general.name");
        const GgufValue * desc = g.find("general.description");
        if (name) printf("%s: model name:   %s\n", "clip_model_load", name->str.c_str());
        if (desc) printf("%s: description:  %s\n", "clip_model_load", desc->str.c_str());
        printf("%s: GGUF version: %u\n", "clip_model_load", g.version);
        printf("%s: alignment:    %zu\n", "clip_model_load", (size_t)g.alignment);
        printf("%s: n_tensors:    %zu\n", "clip_model_load", g.tensors.size());
        printf("%s: n_kv:         %zu\n", "clip_model_load", g.kv.size());
        printf("%s: ftype:        %s\n\n", "clip_model_load", ggml_type_name((int)ftype));
    }
    if (verbosity >= 3) {
        for (size_t i = 0; i < g.kv.size(); i++) printf("%s: kv[%zu]: key = %s\n", "clip_model_load", i, g.kv[i].first.c_str());
        for (size_t i = 0; i < g.tensors.size(); i++) {
            const auto & t = g.tensors[i];
            printf("%s: tensor[%zu]: n_dims = %d, name = %s, tensor_size=%zu, offset=%zu\n", "clip_model_load", i, t.n_dims,
                   t.name.c_str(), t.nbytes, (size_t)t.offset);
        }
    }

    // ---- weight image: laid out (and, unless the cache holds it, filled) by the Loader ----
    // Repacked-weight cache, opt-in: CLIP_AMD_WEIGHT_CACHE=<directory>.  The HBM image is a pure function of the GGUF
    // content, so a file keyed by that content replaces the host-side repack (block-column-major planes, fused QKV, dequantised
    // tables: 0.3 s for ViT-B/32 q4_0, 2 s for ViT-L/14 f16 on one core) with one sequential read; the GGUF's tensor pages are then
    // never touched.  A missing, truncated or mismatching cache file is ignored and rewritten.
    WeightCache cache;
    cache.init(g, fname);
    std::unique_ptr<Loader> Lp;
    auto build = [&](Loader & L, std::string & why) -> bool {
        std::string kerr;
        ctx->vision = DevTower();
        ctx->text = DevTower();
        ctx->id_to_token.clear();
        ctx->token_to_id.clear();
        ctx->max_token_len = 0;
        if (ctx->has_text_encoder) {
            auto & hp = ctx->text_hparams;
            bool ok = kv_u32(g, "clip.text.embedding_length", hp.hidden_size, kerr) & kv_u32(g, "clip.text.attention.head_count", hp.n_head, kerr) &
                      kv_u32(g, "clip.text.feed_forward_length", hp.n_intermediate, kerr) & kv_u32(g, "clip.text.block_count", hp.n_layer, kerr) &
                      kv_u32(g, "clip.text.context_length", hp.num_positions, kerr) & kv_u32(g, "clip.text.projection_dim", hp.projection_dim, kerr);
            if (!g.get_f32("clip.text.attention.layer_norm_epsilon", hp.eps)) { ok = false; if (kerr.empty()) kerr = "key clip.text.attention.layer_norm_epsilon not found in file"; }
            const GgufValue * toks = g.find("tokenizer.ggml.tokens");
            if (!toks || toks->type != GV_ARR || toks->elem_type != GV_STR) { ok = false; if (kerr.empty()) kerr = "key tokenizer.ggml.tokens not found in file"; }
            if (!ok) { why = kerr; return false; }
            hp.n_vocab = (int32_t)toks->strs.size();
            ctx->id_to_token = toks->strs;
            ctx->token_to_id.reserve(toks->strs.size() * 2);
            for (int32_t id = 0; id < hp.n_vocab; id++) {
                ctx->token_to_id[toks->strs[id]] = id;   // later duplicates win, as with the reference's std::map assignment
                ctx->max_token_len = std::max(ctx->max_token_len, toks->strs[id].size());
            }
            if (hp.hidden_size <= 0 || hp.n_head <= 0 || hp.hidden_size % hp.n_head || hp.hidden_size % 64 || hp.n_intermediate % 64)
                { why = "unsupported text hparams (hidden/ff must be multiples of 64)"; return false; }
            { const std::string lim = kernel_limits("text", hp.hidden_size, hp.n_head, hp.projection_dim, hp.num_positions); if (!lim.empty()) { why = lim; return false; } }
            if (verbosity >= 2) {
                printf("\n%s: text model hparams\n", "clip_model_load");
                printf("n_vocab            %d\nnum_positions      %d\nt_hidden_size      %d\nt_n_intermediate   %d\n", hp.n_vocab, hp.num_positions, hp.hidden_size, hp.n_intermediate);
                printf("t_projection_dim   %d\nt_n_head           %d\nt_n_layer          %d\n", hp.projection_dim, hp.n_head, hp.n_layer);
            }
            DevTower & T = ctx->text;
            const int h = hp.hidden_size;
            if (!L.raw("t.token_embd.weight", &T.tok_raw, &T.tok_type, h, hp.n_vocab) ||
                !L.table_f32("t.position_embd.weight", &T.pos, h, hp.num_positions) ||
                !L.vec_f32({"t.post_ln.weight"}, &T.post_ln_w, h) || !L.vec_f32({"t.post_ln.bias"}, &T.post_ln_b, h) ||
                !L.linear({"text_projection.weight"}, T.proj, h, hp.projection_dim) ||
                !L.layers("t", hp.n_layer, h, hp.n_intermediate, T.layers))
                { why = L.err; return false; }
        }
        if (ctx->has_vision_encoder) {
            auto & hp = ctx->vision_hparams;
            bool ok = kv_u32(g, "clip.vision.embedding_length", hp.hidden_size, kerr) & kv_u32(g, "clip.vision.attention.head_count", hp.n_head, kerr) &
                      kv_u32(g, "clip.vision.feed_forward_length", hp.n_intermediate, kerr) & kv_u32(g, "clip.vision.block_count", hp.n_layer, kerr) &
                      kv_u32(g, "clip.vision.image_size", hp.image_size, kerr) & kv_u32(g, "clip.vision.patch_size", hp.patch_size, kerr) &
                      kv_u32(g, "clip.vision.projection_dim", hp.projection_dim, kerr);
            if (!g.get_f32("clip.vision.attention.layer_norm_epsilon", hp.eps)) { ok = false; if (kerr.empty()) kerr = "key clip.vision.attention.layer_norm_epsilon not found in file"; }
            const GgufValue * mean = g.find("clip.vision.image_mean");
            const GgufValue * stdv = g.find("clip.vision.image_std");
            if (!mean || !stdv || mean->type != GV_ARR || stdv->type != GV_ARR || mean->elem_type != GV_F32 || stdv->elem_type != GV_F32 ||
                mean->count < 3 || stdv->count < 3) { ok = false; if (kerr.empty()) kerr = "key clip.vision.image_mean/std not found in file"; }
            if (!ok) { why = kerr; return false; }
            memcpy(ctx->image_mean, mean->raw.data(), 12);
            memcpy(ctx->image_std, stdv->raw.data(), 12);
            if (hp.hidden_size <= 0 || hp.n_head <= 0 || hp.hidden_size % hp.n_head || hp.hidden_size % 64 || hp.n_intermediate % 64 ||
                hp.patch_size <= 0 || hp.image_size % hp.patch_size)
                { why = "unsupported vision hparams (hidden/ff must be multiples of 64)"; return false; }
            { const int g_ = hp.image_size / hp.patch_size; const std::string lim = kernel_limits("vision", hp.hidden_size, hp.n_head, hp.projection_dim, g_ * g_ + 1); if (!lim.empty()) { why = lim; return false; } }
            if (verbosity >= 2) {
                printf("\n%s: vision model hparams\n", "clip_model_load");
                printf("image_size         %d\npatch_size         %d\nv_hidden_size      %d\nv_n_intermediate   %d\n", hp.image_size, hp.patch_size, hp.hidden_size, hp.n_intermediate);
                printf("v_projection_dim   %d\nv_n_head           %d\nv_n_layer          %d\n", hp.projection_dim, hp.n_head, hp.n_layer);
            }
            DevTower & V = ctx->vision;
            const int h = hp.hidden_size, P = hp.patch_size, Gd = hp.image_size / P, T = Gd * Gd + 1;
            const GgufTensorInfo * pe = L.need("v.patch_embd.weight");
            if (!pe) { why = L.err; return false; }
            if (pe->type != GT_F16 && pe->type != GT_F32) { why = "v.patch_embd.weight must be f16"; return false; }
            if (!L.linear({"v.patch_embd.weight"}, V.patch, 3 * P * P, h, /*as_conv=*/true) ||
                !L.vec_f32({"v.class_embd"}, &V.class_embd, h) ||
                !L.table_f32("v.position_embd.weight", &V.pos, h, T) ||
                !L.vec_f32({"v.pre_ln.weight"}, &V.pre_ln_w, h) || !L.vec_f32({"v.pre_ln.bias"}, &V.pre_ln_b, h) ||
                !L.vec_f32({"v.post_ln.weight"}, &V.post_ln_w, h) || !L.vec_f32({"v.post_ln.bias"}, &V.post_ln_b, h) ||
                !L.linear({"visual_projection.weight"}, V.proj, h, hp.projection_dim) ||
                !L.layers("v", hp.n_layer, h, hp.n_intermediate, V.layers))
                { why = L.err; return false; }
        }
        return true;
    };
    for (int attempt = 0; attempt < 2; attempt++) {
        Lp.reset(new Loader(g));
        Lp->st.plan_only = cache.readable;
        std::string why;
        if (!build(*Lp, why)) return fail(why);
        {   // identity of the layout plan: where every tensor sits in the image (a layout change that keeps the total size must not read a stale cache)
            uint64_t ph = 1469598103934665603ull;
            for (const Fix & f : Lp->fixes) ph = WeightCache::fnv(ph, &f.off, sizeof f.off);
            const uint64_t tot = Lp->st.size;
            ph = WeightCache::fnv(ph, &tot, 8);
            cache.plan_hash = (uint32_t)(ph ^ (ph >> 32));
        }
        if (!Lp->st.plan_only) break;
        if (cache.image_bytes == Lp->st.size && cache.file_plan_hash == cache.plan_hash) break;   // the cached image has the planned layout: use it
        cache.readable = false;                          // layout changed (another library version wrote it): rebuild and overwrite
    }
    Loader & L = *Lp;
    if (verbosity >= 1) {
        printf("%s: text_encoder:   %d\n", "clip_model_load", ctx->has_text_encoder);
        printf("%s: vision_encoder: %d\n", "clip_model_load", ctx->has_vision_encoder);
        printf("%s: model size:     %.2f MB (HBM image, repacked)\n", "clip_model_load", L.st.size / 1024.0 / 1024.0);
    }

    // ---- device ----
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (ndev <= 0) {
        (void)hipGetLastError();
        const char * allow = getenv("CLIP_AMD_ALLOW_NO_DEVICE");
        if (allow && allow[0] == '1') {
            if (verbosity >= 1) fprintf(stderr, "clip_model_load: no HIP device — host-only context (tokenizer / preprocessing only; encoders will fail)\n");
            ctx->device = -1;
            if (L.st.plan_only) ctx->weights_from_cache = true;   // (a valid cache file exists; nothing to upload here)
            else cache.write_image(L.st.buf);                     // a GPU-less machine can pre-build the cache
            return ctx;
        }
        return fail("no HIP device available: this library has no CPU fallback (set CLIP_AMD_ALLOW_NO_DEVICE=1 for a host-only context)");
    }
    if (device < 0 || device >= ndev) return fail("invalid HIP device ordinal " + std::to_string(device));
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    ctx->device = device;
    { const char * g = getenv("CLIP_AMD_GRAPHS"); if (g && g[0] == '0') ctx->graphs_enabled = false; }
    if (!alloc_runtime_buffers(ctx)) return fail("stream / split-K workspace / LayerNorm statistics allocation failed");
    {
        int lo = 0, hi = 0, ways = 2;      // CLIP_AMD_SPLIT=min,max[,ways]: batch sizes whose forward is split over 2 (3, 4) streams (forward.cpp); "0,0" = never
        const char * e = getenv("CLIP_AMD_SPLIT");
        if (e && sscanf(e, "%d,%d,%d", &lo, &hi, &ways) >= 2) { ctx->split_min = lo; ctx->split_max = hi; ctx->split_ways = ways < 2 ? 2 : ways > 4 ? 4 : ways; }
    }
    ctx->weights_bytes = L.st.size + 256;
    if (hipMalloc(&ctx->weights_base, ctx->weights_bytes) != hipSuccess) return fail("hipMalloc of the weight image failed");
    if (L.st.plan_only) {
        if (!cache.read_image(L.st.buf, L.st.size)) {
            // unreadable after all (truncated behind our back): fall back to the GGUF
            Lp.reset(new Loader(g));
            std::string why;
            if (!build(*Lp, why)) return fail(why);
        } else {
            ctx->weights_from_cache = true;
        }
    }
    Loader & LL = *Lp;
    if (hipMemcpy(ctx->weights_base, LL.st.buf.data(), LL.st.size, hipMemcpyHostToDevice) != hipSuccess) return fail("weight upload failed");
    if (!ctx->weights_from_cache) cache.write_image(LL.st.buf);
    for (const Fix & f : LL.fixes) *f.slot = (const uint8_t *)ctx->weights_base + f.off;
    {   // LayerNorm fold vectors (k_fold.hip): computed on the device from the weights as the GEMM kernels dequantise them
        bool f32_weights = false;                    // f32 file: weights stay f32 (k_gemm_f32.hip) — no fold, no small-M kernels, no panels for them
        for (DevTower * tw : {&ctx->vision, &ctx->text})
            for (DevLayer & l : tw->layers) {
                if (l.qkv.wtype == W_F32 || l.o.wtype == W_F32 || l.ff1.wtype == W_F32 || l.ff2.wtype == W_F32) { f32_weights = true; continue; }
                launch_fold_vectors(l.qkv, l.ln1_w, l.ln1_b, l.qkv_b, (float *)l.qkv_c, (float *)l.qkv_bf, ctx->stream);
                launch_fold_vectors(l.ff1, l.ln2_w, l.ln2_b, l.ff1_b, (float *)l.ff1_c, (float *)l.ff1_bf, ctx->stream);
            }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) return fail("LayerNorm fold vectors: kernel failed");
        const char * e = getenv("CLIP_AMD_LNFOLD");
        ctx->ln_fold = !(e && e[0] == '0') && !f32_weights;
        ctx->ln_fold_force = e && e[0] == '2';       // tuning: fold even where forward.cpp fold_pays() says the LayerNorm launches are cheaper
        const char * e2 = getenv("CLIP_AMD_RESIDENT_PANELS");
        ctx->resident_panels_on = !(e2 && e2[0] == '0');
        const char * ef = getenv("CLIP_AMD_F32_ACTS");
        ctx->f32_acts = !(ef && ef[0] == '0');
        const char * ep = getenv("CLIP_AMD_PRUNE_LAST");
        ctx->prune_last = !(ep && ep[0] == '0');
        const char * ec = getenv("CLIP_AMD_LNFOLD_CENTRE");
        ctx->ln_fold_centre = !(ec && ec[0] == '0');  // A/B: the r03 form, fp16(x gamma) without the per-row offset
    }
    if (verbosity >= 1) printf("\n%s: %zu MB of HBM allocated for weights on device %d\n", "clip_model_load", ctx->weights_bytes / 1024 / 1024, device);
    return ctx;
}

namespace {
bool alloc_runtime_buffers(clip_ctx * ctx) {   // stream + split-K workspace (64 MB of partial tiles, 4096 ticket counters) + small-M LayerNorm statistics
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return false;
    ctx->stream = ctx->own_stream;
    const size_t nfl = (size_t)16 << 20;
    void * w = nullptr, * c = nullptr;
    if (hipMalloc(&w, nfl * sizeof(float)) != hipSuccess || hipMalloc(&c, 4096 * sizeof(unsigned)) != hipSuccess || hipMemset(c, 0, 4096 * sizeof(unsigned)) != hipSuccess) return false;
    ctx->sk_ws = (float *)w; ctx->sk_ws_floats = nfl; ctx->sk_cnt = (unsigned *)c; ctx->sk_cnt_n = 4096;
    return hipMalloc((void **)&ctx->sk_stats, (size_t)2 * SKINNY_MAX_ROWS * 128 * sizeof(float2)) == hipSuccess;
}
}  // namespace

// A second context on the owner's device that multiplies the owner's weight image (no reload, no second copy in HBM): own stream,
// own activation workspace / split-K buffers / statistics.  Encoder state only — no tokenizer tables, no host pipeline.
clip_ctx * sibling_context(clip_ctx * owner, int index) {
    if (!owner || owner->device < 0 || owner->weights_borrowed || index < 0 || index > 2) return nullptr;
    clip_ctx *& slot = index == 0 ? owner->sibling : owner->more_siblings[index - 1];
    if (slot) return slot;
    (void)hipSetDevice(owner->device);
    clip_ctx * c = new clip_ctx();
    c->has_text_encoder = owner->has_text_encoder; c->has_vision_encoder = owner->has_vision_encoder; c->use_gelu = owner->use_gelu; c->ftype = owner->ftype;
    c->text_hparams = owner->text_hparams; c->vision_hparams = owner->vision_hparams;
    memcpy(c->image_mean, owner->image_mean, sizeof c->image_mean); memcpy(c->image_std, owner->image_std, sizeof c->image_std);
    c->device = owner->device;
    c->vision = owner->vision; c->text = owner->text;            // device pointers into the owner's weight image
    c->weights_borrowed = true;
    c->owner = owner;
    c->ln_fold = owner->ln_fold; c->ln_fold_force = owner->ln_fold_force; c->ln_fold_centre = owner->ln_fold_centre; c->prune_last = owner->prune_last;
    c->f32_acts = owner->f32_acts;
    c->resident_panels_on = owner->resident_panels_on;           // (the table itself is the OWNER's: forward.cpp resident_panels — read-only, same device)
    c->graphs_enabled = false;                                   // (its launches are captured into the OWNER's graphs)
    c->split_min = c->split_max = 0;                             // (a sibling never splits)
    c->verbosity = 0; c->path = owner->path;
    hipEvent_t & evj = index == 0 ? owner->ev_join : owner->ev_join_more[index - 1];
    if (!alloc_runtime_buffers(c) || (!owner->ev_fork && hipEventCreateWithFlags(&owner->ev_fork, hipEventDisableTiming) != hipSuccess) ||
        hipEventCreateWithFlags(&evj, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "clip (hip): cannot create the sibling context\n");
        free_model(c);
        return nullptr;
    }
    slot = c;
    return c;
}

void free_model(clip_ctx * ctx) {
    if (!ctx) return;
    if (ctx->multi) multi_free(ctx);     // replicas on the other devices, RCCL communicators
    if (ctx->sibling) { free_model(ctx->sibling); ctx->sibling = nullptr; }
    for (clip_ctx *& ms : ctx->more_siblings) if (ms) { free_model(ms); ms = nullptr; }
    if (ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
        if (ctx->stream && ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
        free_host_pipe(ctx);
        for (int i = 0; i < MetaRing::SLOTS; i++) {
            if (ctx->meta.pin[i]) (void)hipHostFree(ctx->meta.pin[i]);
            if (ctx->meta.ev[i]) (void)hipEventDestroy(ctx->meta.ev[i]);
        }
        if (ctx->meta.done) (void)hipHostFree((void *)ctx->meta.done);
        if (ctx->ev_stream_switch) (void)hipEventDestroy(ctx->ev_stream_switch);
        if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
        for (hipEvent_t e : ctx->ev_join_more) if (e) (void)hipEventDestroy(e);
        for (auto & p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        drop_graphs(ctx);
        if (ctx->ws.base) (void)hipFree(ctx->ws.base);
        if (ctx->weights_base && !ctx->weights_borrowed) (void)hipFree(ctx->weights_base);
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        if (ctx->io_in) (void)hipFree(ctx->io_in);
        if (ctx->io_out) (void)hipFree(ctx->io_out);
        if (ctx->pre_buf) (void)hipFree(ctx->pre_buf);
        free_preprocess_slots(ctx);
        if (ctx->w16_panel) (void)hipFree(ctx->w16_panel);
        for (auto * b : ctx->res_panel_buf) if (b) (void)hipFree(b);
        if (ctx->sk_stats) (void)hipFree(ctx->sk_stats);
        if (ctx->sk_ws) (void)hipFree(ctx->sk_ws);
        if (ctx->sk_cnt) (void)hipFree(ctx->sk_cnt);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    }
    delete ctx;
}

// Test hook: repack one raw ggml-layout weight through the production path and upload it.
bool repack_for_test(int type, const void * w_raw, int64_t N, int64_t K, DevWeight & W, void ** dev_base) {
    RepackPlan pl;
    std::string err;
    if (!plan_repack(type, N, K, pl, err)) { fprintf(stderr, "repack_for_test: %s\n", err.c_str()); return false; }
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_w16 = 0, o_qs = up(pl.bytes_w16), o_qh = o_qs + up(pl.bytes_qs), o_dm = o_qh + up(pl.bytes_qh);
    const size_t total = o_dm + up(pl.bytes_dm) + 256;
    std::vector<uint8_t> host(total, 0);
    repack_rows(pl, type, (const uint8_t *)w_raw, N, K, 0, &host[o_qs], &host[o_qh], &host[o_dm], &host[o_w16]);
    void * base = nullptr;
    if (hipMalloc(&base, total) != hipSuccess) return false;
    if (hipMemcpy(base, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(base); return false; }
    W = pl.W;
    const uint8_t * b = (const uint8_t *)base;
    if (pl.bytes_w16) W.w16 = b + o_w16;
    if (pl.bytes_qs) W.qs = b + o_qs;
    if (pl.bytes_qh) W.qh = b + o_qh;
    if (pl.bytes_dm) W.dm = b + o_dm;
    *dev_base = base;
    return true;
}

}  // namespace clipamd
