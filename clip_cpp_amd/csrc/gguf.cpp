// gguf.cpp — GGUF v2/v3 reader (mmap) and writer.  See gguf.h.
#include "gguf.h"

#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace clipamd {

size_t ggml_row_bytes(int type, int64_t k) {
    switch (type) {
    case GT_F32: return (size_t)k * 4;
    case GT_F16: return (size_t)k * 2;
    case GT_Q4_0: return (k % 32) ? 0 : (size_t)(k / 32) * 18;
    case GT_Q4_1: return (k % 32) ? 0 : (size_t)(k / 32) * 20;
    case GT_Q5_0: return (k % 32) ? 0 : (size_t)(k / 32) * 22;
    case GT_Q5_1: return (k % 32) ? 0 : (size_t)(k / 32) * 24;
    case GT_Q8_0: return (k % 32) ? 0 : (size_t)(k / 32) * 34;
    }
    return 0;
}

const char * ggml_type_name(int type) {
    switch (type) {
    case GT_F32: return "f32";
    case GT_F16: return "f16";
    case GT_Q4_0: return "q4_0";
    case GT_Q4_1: return "q4_1";
    case GT_Q5_0: return "q5_0";
    case GT_Q5_1: return "q5_1";
    case GT_Q8_0: return "q8_0";
    }
    return "?";
}

namespace {

struct Cursor {
    const uint8_t * p;
    const uint8_t * end;
    bool ok = true;
    template <typename T> T rd() {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        uint64_t n = rd<uint64_t>();
        if (!ok || n > (size_t)(end - p)) { ok = false; return std::string(); }
        std::string s((const char *)p, (size_t)n);
        p += n;
        return s;
    }
    bool skip(size_t n) {
        if (n > (size_t)(end - p)) { ok = false; return false; }
        p += n;
        return true;
    }
};

size_t scalar_size(uint32_t t) {
    switch (t) {
    case GV_U8: case GV_I8: case GV_BOOL: return 1;
    case GV_U16: case GV_I16: return 2;
    case GV_U32: case GV_I32: case GV_F32: return 4;
    case GV_U64: case GV_I64: case GV_F64: return 8;
    }
    return 0;
}

double scalar_as_double(uint32_t t, const uint8_t * p) {
    switch (t) {
    case GV_U8: return *p;
    case GV_I8: return (int8_t)*p;
    case GV_BOOL: return *p != 0;
    case GV_U16: { uint16_t v; memcpy(&v, p, 2); return v; }
    case GV_I16: { int16_t v; memcpy(&v, p, 2); return v; }
    case GV_U32: { uint32_t v; memcpy(&v, p, 4); return v; }
    case GV_I32: { int32_t v; memcpy(&v, p, 4); return v; }
    case GV_F32: { float v; memcpy(&v, p, 4); return v; }
    case GV_U64: { uint64_t v; memcpy(&v, p, 8); return (double)v; }
    case GV_I64: { int64_t v; memcpy(&v, p, 8); return (double)v; }
    case GV_F64: { double v; memcpy(&v, p, 8); return v; }
    }
    return 0;
}

}  // namespace

GgufFile::~GgufFile() {
    if (map_) munmap(map_, map_size_);
}

bool GgufFile::open(const char * path, std::string & err) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 24) { ::close(fd); err = "file too small to be GGUF"; return false; }
    map_size_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) { map_ = nullptr; err = "mmap failed"; return false; }
    const uint8_t * base = (const uint8_t *)map_;
    Cursor c{base, base + map_size_};
    if (memcmp(base, "GGUF", 4) != 0) { err = "bad magic (not a GGUF file)"; return false; }
    c.p += 4;
    version = c.rd<uint32_t>();
    if (version != 2 && version != 3) { err = "unsupported GGUF version " + std::to_string(version); return false; }
    const uint64_t n_tensors = c.rd<uint64_t>();
    const uint64_t n_kv = c.rd<uint64_t>();
    if (!c.ok || n_tensors > (1u << 20) || n_kv > (1u << 20)) { err = "corrupt GGUF header"; return false; }

    kv.reserve(n_kv);
    for (uint64_t i = 0; i < n_kv; i++) {
        std::string key = c.str();
        GgufValue v;
        v.type = c.rd<uint32_t>();
        const uint8_t * vstart = c.p;
        if (!c.ok) break;
        if (v.type == GV_STR) {
            v.str = c.str();
        } else if (v.type == GV_ARR) {
            v.elem_type = c.rd<uint32_t>();
            v.count = c.rd<uint64_t>();
            if (!c.ok) break;
            if (v.elem_type == GV_STR) {
                if (v.count > (1u << 26) || v.count > (uint64_t)(c.end - c.p) / 8) { c.ok = false; break; }   // every string costs >= 8 bytes (its length)
                v.strs.reserve((size_t)v.count);
                for (uint64_t j = 0; j < v.count && c.ok; j++) v.strs.push_back(c.str());
            } else {
                const size_t es = scalar_size(v.elem_type);
                if (es == 0 || v.count > (size_t)(c.end - c.p) / es) { c.ok = false; break; }
                v.raw.assign(c.p, c.p + es * v.count);
                c.skip(es * v.count);
            }
        } else {
            const size_t es = scalar_size(v.type);
            if (es == 0 || (size_t)(c.end - c.p) < es) { c.ok = false; break; }
            v.num = scalar_as_double(v.type, c.p);
            c.skip(es);
        }
        if (!c.ok) break;
        v.encoded.assign(vstart, c.p);
        kv_index_[key] = kv.size();
        kv.emplace_back(std::move(key), std::move(v));
    }
    if (!c.ok) { err = "corrupt GGUF key/value section"; return false; }
    if (const GgufValue * a = find("general.alignment")) alignment = (uint64_t)a->num;
    if (alignment == 0 || (alignment & (alignment - 1))) { err = "bad general.alignment"; return false; }

    tensors.resize(n_tensors);
    for (uint64_t i = 0; i < n_tensors; i++) {
        GgufTensorInfo & t = tensors[i];
        t.name = c.str();
        t.n_dims = (int)c.rd<uint32_t>();
        if (!c.ok || t.n_dims < 1 || t.n_dims > 4) { c.ok = false; break; }
        for (int d = 0; d < t.n_dims; d++) t.ne[d] = (int64_t)c.rd<uint64_t>();
        t.type = (int)c.rd<uint32_t>();
        t.offset = c.rd<uint64_t>();
        if (!c.ok) break;
        for (int d = 0; d < 4; d++)
            if (t.ne[d] <= 0 || t.ne[d] > (1ll << 32)) c.ok = false;
        if (!c.ok) break;
        // element / byte counts with overflow checks: four dims of up to 2^32 each would wrap int64 (a wrapped size passes the
        // bounds check below); no tensor of a real file is anywhere near the file size, so bound the running product by it
        unsigned __int128 nel = 1;
        for (int d = 0; d < 4; d++) {
            nel *= (unsigned __int128)t.ne[d];
            if (nel > ((unsigned __int128)map_size_ << 3)) { err = "tensor " + t.name + ": shape larger than the file"; return false; }   // >= 1 bit per element
        }
        const size_t rb = ggml_row_bytes(t.type, t.ne[0]);
        if (rb == 0) { err = "tensor " + t.name + ": unsupported type/shape"; return false; }
        const unsigned __int128 nb = (unsigned __int128)rb * (unsigned __int128)(nel / (unsigned __int128)t.ne[0]);
        if (nb > (unsigned __int128)map_size_) { err = "tensor " + t.name + " is larger than the file"; return false; }
        t.nbytes = (size_t)nb;
        tensor_index_[t.name] = i;
    }
    if (!c.ok) { err = "corrupt GGUF tensor-info section"; return false; }
    const uint64_t pos = (uint64_t)(c.p - base);
    data_offset = (pos + alignment - 1) / alignment * alignment;
    if (data_offset > map_size_) { err = "tensor data section starts past end of file"; return false; }
    for (auto & t : tensors) {
        // no wrap-around: offset and nbytes are each checked against what is left of the file
        if (t.offset > map_size_ - data_offset || t.nbytes > map_size_ - data_offset - t.offset) { err = "tensor " + t.name + " extends past end of file"; return false; }
        t.data = base + data_offset + t.offset;
    }
    return true;
}

const GgufValue * GgufFile::find(const std::string & key) const {
    auto it = kv_index_.find(key);
    return it == kv_index_.end() ? nullptr : &kv[it->second].second;
}
const GgufTensorInfo * GgufFile::tensor(const std::string & name) const {
    auto it = tensor_index_.find(name);
    return it == tensor_index_.end() ? nullptr : &tensors[it->second];
}
bool GgufFile::get_u32(const std::string & key, uint32_t & out) const {
    const GgufValue * v = find(key);
    if (!v || v->type == GV_STR || v->type == GV_ARR) return false;
    out = (uint32_t)v->num;
    return true;
}
bool GgufFile::get_f32(const std::string & key, float & out) const {
    const GgufValue * v = find(key);
    if (!v || v->type == GV_STR || v->type == GV_ARR) return false;
    out = (float)v->num;
    return true;
}
bool GgufFile::get_bool(const std::string & key, bool & out) const {
    const GgufValue * v = find(key);
    if (!v || v->type == GV_STR || v->type == GV_ARR) return false;
    out = v->num != 0;
    return true;
}

GgufValue gguf_make_u32(uint32_t x) {
    GgufValue v;
    v.type = GV_U32;
    v.num = x;
    v.encoded.resize(4);
    memcpy(v.encoded.data(), &x, 4);
    return v;
}

bool gguf_write(const char * path, uint32_t version, uint64_t alignment,
                const std::vector<std::pair<std::string, GgufValue>> & kv, const std::vector<GgufOutTensor> & tensors,
                std::string & err) {
    std::vector<uint8_t> meta;
    auto put = [&](const void * p, size_t n) { meta.insert(meta.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    auto put_str = [&](const std::string & s) { uint64_t n = s.size(); put(&n, 8); put(s.data(), s.size()); };
    put("GGUF", 4);
    put(&version, 4);
    uint64_t nt = tensors.size(), nkv = kv.size();
    put(&nt, 8);
    put(&nkv, 8);
    for (auto & e : kv) {
        put_str(e.first);
        put(&e.second.type, 4);
        put(e.second.encoded.data(), e.second.encoded.size());
    }
    uint64_t off = 0;
    for (auto & t : tensors) {
        put_str(t.name);
        uint32_t nd = (uint32_t)t.n_dims;
        put(&nd, 4);
        for (int d = 0; d < t.n_dims; d++) { uint64_t v = (uint64_t)t.ne[d]; put(&v, 8); }
        uint32_t ty = (uint32_t)t.type;
        put(&ty, 4);
        put(&off, 8);
        off += (t.nbytes + alignment - 1) / alignment * alignment;
    }
    while (meta.size() % alignment) meta.push_back(0);
    FILE * f = fopen(path, "wb");
    if (!f) { err = std::string("cannot create ") + path; return false; }
    bool ok = fwrite(meta.data(), 1, meta.size(), f) == meta.size();
    static const uint8_t zeros[256] = {0};
    for (auto & t : tensors) {
        if (!ok) break;
        ok = fwrite(t.data, 1, t.nbytes, f) == t.nbytes;
        size_t pad = (alignment - t.nbytes % alignment) % alignment;
        while (ok && pad) { size_t n = pad > 256 ? 256 : pad; ok = fwrite(zeros, 1, n, f) == n; pad -= n; }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) err = "short write";
    return ok;
}

}  // namespace clipamd
