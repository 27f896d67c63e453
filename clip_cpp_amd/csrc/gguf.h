// gguf.h — minimal GGUF v2/v3 reader/writer for CLIP model files (host side).
//
// Replaces the gguf_* API of the (absent) ggml submodule that the reference calls in
// clip_model_load (clip.cpp:338-361,381-390,442-457,476-483,537-542) and in
// clip_model_quantize (clip.cpp:1690-1820).  Container layout: SURVEY Appendix A.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace clipamd {

enum GgmlType : int { GT_F32 = 0, GT_F16 = 1, GT_Q4_0 = 2, GT_Q4_1 = 3, GT_Q5_0 = 6, GT_Q5_1 = 7, GT_Q8_0 = 8 };

// bytes of one row of `k` elements in ggml block layout; 0 if the type is unsupported
size_t ggml_row_bytes(int type, int64_t k);
const char * ggml_type_name(int type);

enum GgufValueType : uint32_t {
    GV_U8 = 0, GV_I8 = 1, GV_U16 = 2, GV_I16 = 3, GV_U32 = 4, GV_I32 = 5, GV_F32 = 6, GV_BOOL = 7,
    GV_STR = 8, GV_ARR = 9, GV_U64 = 10, GV_I64 = 11, GV_F64 = 12,
};

struct GgufValue {
    uint32_t type = GV_U32;
    uint32_t elem_type = 0;          // for arrays
    double num = 0;                  // scalar numeric / bool
    std::string str;                 // GV_STR
    std::vector<std::string> strs;   // array of strings
    std::vector<uint8_t> raw;        // array of scalars (packed)
    uint64_t count = 0;              // array length
    std::vector<uint8_t> encoded;    // the exact on-disk bytes of the value (after the type tag); used by the writer
};

struct GgufTensorInfo {
    std::string name;
    int type = 0;
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};    // ne[0] = contiguous dimension
    uint64_t offset = 0;             // relative to data section
    size_t nbytes = 0;
    const uint8_t * data = nullptr;  // into the mapping
    int64_t nrows() const { return ne[1] * ne[2] * ne[3]; }
};

class GgufFile {
public:
    GgufFile() = default;
    ~GgufFile();
    GgufFile(const GgufFile &) = delete;
    GgufFile & operator=(const GgufFile &) = delete;

    // returns false (with `err` set) on any malformed input; never throws
    bool open(const char * path, std::string & err);

    uint32_t version = 0;
    uint64_t alignment = 32;
    uint64_t data_offset = 0;
    std::vector<std::pair<std::string, GgufValue>> kv;   // file order
    std::vector<GgufTensorInfo> tensors;                 // file order

    const GgufValue * find(const std::string & key) const;
    const GgufTensorInfo * tensor(const std::string & name) const;
    bool get_u32(const std::string & key, uint32_t & out) const;
    bool get_f32(const std::string & key, float & out) const;
    bool get_bool(const std::string & key, bool & out) const;

    // the mapped file (identity of its content for the repacked-weight cache, load.cpp)
    const uint8_t * file_base() const { return (const uint8_t *)map_; }
    size_t file_size() const { return map_size_; }

private:
    void * map_ = nullptr;
    size_t map_size_ = 0;
    std::map<std::string, size_t> kv_index_, tensor_index_;
};

// Writer used by clip_model_quantize: re-emits KVs verbatim (plus overrides) and new tensor payloads.
struct GgufOutTensor {
    std::string name;
    int type;
    int n_dims;
    int64_t ne[4];
    const uint8_t * data;
    size_t nbytes;
};
bool gguf_write(const char * path, uint32_t version, uint64_t alignment,
                const std::vector<std::pair<std::string, GgufValue>> & kv, const std::vector<GgufOutTensor> & tensors,
                std::string & err);
GgufValue gguf_make_u32(uint32_t v);

}  // namespace clipamd
