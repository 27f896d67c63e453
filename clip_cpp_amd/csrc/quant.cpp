// quant.cpp — host codecs for the ggml block formats used by CLIP GGUF files.
//
// The reference obtains these from the ggml submodule (ggml_quantize_q4_0 ... q8_0 called at
// reference clip.cpp:1771-1791; dequantize_row_q* behind ggml_get_rows at :1059,1061,1331).  That
// source is not in the reference tree, so the formats are implemented from their published definition
// (QK = 32):
//   q4_0  {f16 d;          u8 qs[16]}          x = (q - 8) d        q in [0,15]
//   q4_1  {f16 d; f16 m;   u8 qs[16]}          x = q d + m
//   q5_0  {f16 d; u8 qh[4]; u8 qs[16]}         x = (q - 16) d       q in [0,31], bit 4 of elem j in qh bit j
//   q5_1  {f16 d; f16 m; u8 qh[4]; u8 qs[16]}  x = q d + m
//   q8_0  {f16 d; i8 qs[32]}                   x = q d
// low nibble of qs[j] = element j, high nibble = element j+16.
#include "model.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace clipamd {

uint16_t f32_to_f16_bits(float x) {
    uint32_t f;
    memcpy(&f, &x, 4);
    const uint32_t sign = (f >> 16) & 0x8000u;
    const uint32_t e8 = (f >> 23) & 0xFFu;
    uint32_t man = f & 0x7FFFFFu;
    if (e8 == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? (0x200u | (man >> 13)) : 0u));
    const int e = (int)e8 - 112;  // rebias 127 -> 15
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int sh = 14 - e;
        uint32_t q = man >> sh;
        const uint32_t rem = man & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t out = sign | ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (out & 1u))) out++;
    return (uint16_t)out;
}

float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e5 = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (e5 == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int sh = 0;
            while (!(man & 0x400u)) { man <<= 1; sh++; }
            bits = sign | (uint32_t)(113 - sh) << 23 | (man & 0x3FFu) << 13;
        }
    } else if (e5 == 31) {
        bits = sign | 0x7F800000u | man << 13;
    } else {
        bits = sign | (e5 + 112u) << 23 | man << 13;
    }
    float r;
    memcpy(&r, &bits, 4);
    return r;
}

namespace {

struct Fmt {
    int bytes;      // block bytes
    int off_m;      // offset of m (or -1)
    int off_qh;     // offset of qh (or -1)
    int off_qs;     // offset of quants
    int bits;       // 4, 5 or 8
    bool affine;    // has m
};

bool fmt_of(int type, Fmt & f) {
    switch (type) {
    case GT_Q4_0: f = {18, -1, -1, 2, 4, false}; return true;
    case GT_Q4_1: f = {20, 2, -1, 4, 4, true}; return true;
    case GT_Q5_0: f = {22, -1, 2, 6, 5, false}; return true;
    case GT_Q5_1: f = {24, 2, 4, 8, 5, true}; return true;
    case GT_Q8_0: f = {34, -1, -1, 2, 8, false}; return true;
    }
    return false;
}

// float -> low byte of the truncated 32-bit integer, i.e. what `(int8_t)t` / `(uint8_t)t` compile to on x86 (cvttss2si + byte move),
// spelled so that NaN / Inf / |t| >= 2^31 — weights of a corrupt f32 file — are defined too (the "integer indefinite" 0x80000000 -> 0)
inline int32_t trunc_i32(float t) { return t > -2147483648.0f && t < 2147483648.0f ? (int32_t)t : INT32_MIN; }
inline int low_i8(float t) { return (int)(int8_t)(uint8_t)(uint32_t)trunc_i32(t); }
inline int low_u8(float t) { return (int)(uint8_t)(uint32_t)trunc_i32(t); }

inline uint16_t rd16(const uint8_t * p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline void wr16(uint8_t * p, uint16_t v) { memcpy(p, &v, 2); }

}  // namespace

void dequantize_row(int type, const void * src, float * dst, int64_t k) {
    if (type == GT_F32) { memcpy(dst, src, (size_t)k * 4); return; }
    if (type == GT_F16) {
        const uint16_t * s = (const uint16_t *)src;
        for (int64_t i = 0; i < k; i++) dst[i] = f16_bits_to_f32(s[i]);
        return;
    }
    Fmt f;
    if (!fmt_of(type, f)) return;
    const uint8_t * blk = (const uint8_t *)src;
    for (int64_t ib = 0; ib < k / 32; ib++, blk += f.bytes, dst += 32) {
        const float d = f16_bits_to_f32(rd16(blk));
        const float m = f.affine ? f16_bits_to_f32(rd16(blk + f.off_m)) : 0.0f;
        if (f.bits == 8) {
            for (int j = 0; j < 32; j++) dst[j] = (float)(int8_t)blk[f.off_qs + j] * d;
            continue;
        }
        uint32_t qh = 0;
        if (f.off_qh >= 0) memcpy(&qh, blk + f.off_qh, 4);
        const int zero = f.affine ? 0 : (f.bits == 4 ? 8 : 16);
        for (int j = 0; j < 32; j++) {
            const uint8_t b = blk[f.off_qs + (j & 15)];
            int q = j < 16 ? (b & 0x0F) : (b >> 4);
            if (f.bits == 5) q |= (int)((qh >> j) & 1u) << 4;
            dst[j] = f.affine ? (float)q * d + m : (float)(q - zero) * d;
        }
    }
}

size_t quantize_rows(int type, const float * src, void * dst, int64_t nrows, int64_t k) {
    const size_t rb = ggml_row_bytes(type, k);
    if (rb == 0) return 0;
    if (type == GT_F32) { memcpy(dst, src, (size_t)nrows * k * 4); return rb * nrows; }
    if (type == GT_F16) {
        uint16_t * o = (uint16_t *)dst;
        for (int64_t i = 0; i < nrows * k; i++) o[i] = f32_to_f16_bits(src[i]);
        return rb * nrows;
    }
    Fmt f;
    if (!fmt_of(type, f)) return 0;
    const int64_t nblk = nrows * (k / 32);
    uint8_t * blk = (uint8_t *)dst;
    for (int64_t ib = 0; ib < nblk; ib++, blk += f.bytes, src += 32) {
        memset(blk, 0, f.bytes);
        if (f.bits == 8) {
            float amax = 0.0f;
            for (int j = 0; j < 32; j++) amax = std::max(amax, fabsf(src[j]));
            const float d = amax / 127.0f;
            const float id = d ? 1.0f / d : 0.0f;
            wr16(blk, f32_to_f16_bits(d));
            for (int j = 0; j < 32; j++) blk[f.off_qs + j] = (uint8_t)low_i8(roundf(src[j] * id));
            continue;
        }
        const int levels = 1 << f.bits;  // 16 or 32
        float d, lo = 0.0f;
        if (f.affine) {
            float mn = FLT_MAX, mx = -FLT_MAX;
            for (int j = 0; j < 32; j++) { mn = std::min(mn, src[j]); mx = std::max(mx, src[j]); }
            d = (mx - mn) / (float)(levels - 1);
            lo = mn;
            wr16(blk + f.off_m, f32_to_f16_bits(mn));
        } else {
            float amax = 0.0f, vmax = 0.0f;  // signed value of the largest magnitude
            for (int j = 0; j < 32; j++)
                if (amax < fabsf(src[j])) { amax = fabsf(src[j]); vmax = src[j]; }
            d = vmax / (float)(-(levels / 2));
        }
        const float id = d ? 1.0f / d : 0.0f;
        wr16(blk, f32_to_f16_bits(d));
        uint32_t qh = 0;
        for (int j = 0; j < 32; j++) {
            int q;
            if (f.affine) {
                const float t = (src[j] - lo) * id + 0.5f;
                q = f.bits == 4 ? std::min(15, low_i8(t)) : low_u8(t);
            } else {
                const float t = src[j] * id + ((float)(levels / 2) + 0.5f);  // single add of 8.5 / 16.5
                q = std::min(levels - 1, low_i8(t));
            }
            const int nib = q & 0x0F;
            blk[f.off_qs + (j & 15)] |= (uint8_t)(j < 16 ? nib : nib << 4);
            if (f.bits == 5) qh |= (uint32_t)((q >> 4) & 1) << j;
        }
        if (f.off_qh >= 0) memcpy(blk + f.off_qh, &qh, 4);
    }
    return rb * nrows;
}

}  // namespace clipamd
