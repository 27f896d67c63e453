// k_misc.hip — the memory-bound kernels of the hot path (HBM roofline): LayerNorm, im2col for the
// patch convolution, class-token rows, text embedding gather, L2 normalisation, dtype conversion.
// Each replaces a chain of materialised ggml temporaries (repeat/mul/add/acc/get_rows/cont,
// SURVEY §8a "where the CPU time goes") with ONE pass over the data.

#include "kernels.h"

namespace clipamd {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one 64-lane wave per row, row held in registers (h <= 64*4*MAXV), two-pass mean/variance
// (ggml_norm semantics: biased variance, y = (x-mean) * 1/sqrt(var+eps), then *w + b;
// reference clip.cpp:1350-1355, ggml_compute_forward_norm).  Wave-shuffle reductions only.
// ---------------------------------------------------------------------------------------------
// NV = float4 per lane (h <= 256*NV): a template parameter so the row stays in NV*4 registers.  RW = rows per wave: with RW = 2 the
// loads of both rows are in flight before the first reduction (twice the bytes in flight per CU: the kernel is a pure HBM stream —
// 59 MB per launch at ViT-B/32 batch 256, measured 4.4 TB/s with one row per wave)
template <int LN_MAXV, int RW>
__global__ void __launch_bounds__(256) layernorm_kernel(const float * __restrict__ x, int ldx, const int * __restrict__ in_rows, int in_row_mul,
                                                        const float * __restrict__ w, const float * __restrict__ b, float eps,
                                                        int rows, int h, half_t * __restrict__ out16, int ld16,
                                                        float * __restrict__ out32, int ld32) {
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (r0 >= rows) return;
    f4 v[RW][LN_MAXV];
    float sum[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) {
        const int r = r0 + q < rows ? r0 + q : rows - 1;
        const long src = in_rows ? (long)in_rows[r] : (long)r * in_row_mul;
        const float * xr = x + (size_t)src * ldx;
        sum[q] = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = (i * 64 + lane) * 4;
            if (c < h) {
                v[q][i] = *(const f4 *)(xr + c);
                sum[q] += v[q][i][0] + v[q][i][1] + v[q][i][2] + v[q][i][3];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < RW; q++) {
        const int r = r0 + q;
        if (r >= rows) break;
        const float mean = wave_sum(sum[q]) / (float)h;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = (i * 64 + lane) * 4;
            if (c < h) {
                v[q][i] = v[q][i] - mean;
                sq += v[q][i][0] * v[q][i][0] + v[q][i][1] * v[q][i][1] + v[q][i][2] * v[q][i][2] + v[q][i][3] * v[q][i][3];
            }
        }
        const float var = wave_sum(sq) / (float)h;
        const float scale = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = (i * 64 + lane) * 4;
            if (c < h) {
                const f4 ww = *(const f4 *)(w + c), bb = *(const f4 *)(b + c);
                const f4 y = (v[q][i] * scale) * ww + bb;
                if (out32) *(f4 *)(out32 + (size_t)r * ld32 + c) = y;
                if (out16) {
                    const h2 lo = (h2){(_Float16)y[0], (_Float16)y[1]}, hi = (h2){(_Float16)y[2], (_Float16)y[3]};
                    *(uint2 *)(out16 + (size_t)r * ld16 + c) =
                        make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
                }
            }
        }
    }
}

// Entry of the LayerNorm-folded layer chain (gemm_common.h): for every row, y = LayerNorm(x) w + b (w == nullptr: y = x), written to
// out32 (may alias x), plus what the first q/k/v GEMM consumes: xg = fp16(y gamma_next) and the whole-row statistics (sum, sum of
// squared deviations from the mean) of y as ONE slot of width h.  The row lives in registers: every pass is a register pass.
template <int NV>
__device__ __forceinline__ void row_fold_prep(const f4 (&y)[NV], int h, int lane, const float * __restrict__ gnext, half_t * __restrict__ xg_row,
                                              float2 * __restrict__ stat, float * __restrict__ mu_row) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
        if ((i * 64 + lane) * 4 < h) s += (y[i][0] + y[i][1]) + (y[i][2] + y[i][3]);
    s = wave_sum(s);
    const float mean = s / (float)h;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++)
        if ((i * 64 + lane) * 4 < h) {
            const f4 d = y[i] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    q = wave_sum(q);
    if (lane == 0) *stat = make_float2(s, q);
    // centred form (GemmParams::ln_mu): the operand is built about the row's own mean, which the first consumer gets as its offset
    const float off = mu_row ? mean : 0.f;
    if (mu_row && lane == 0) *mu_row = mean;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < h) {
            const f4 g = (y[i] - off) * *(const f4 *)(gnext + c);
            const h2 lo = (h2){(_Float16)g[0], (_Float16)g[1]}, hi = (h2){(_Float16)g[2], (_Float16)g[3]};
            *(uint2 *)(xg_row + c) = make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
        }
    }
}

template <int NV>
__global__ void __launch_bounds__(256) layernorm_prep_kernel(const float * x, int ldx, const float * __restrict__ w, const float * __restrict__ b,
                                                             float eps, int rows, int h, float * out32, int ld32,
                                                             const float * __restrict__ gnext, half_t * __restrict__ xg, int ldxg,
                                                             float2 * __restrict__ stats, float * __restrict__ mu_out,
                                                             const float * __restrict__ cls, const float * __restrict__ pos0, int T) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float * xr = x + (size_t)r * ldx;
    const bool cls_row = cls != nullptr && r % T == 0;    // class-token row (reference clip.cpp:1315-1331): class_embd + pos[0], the sum cls_rows_kernel stores — here it never touches memory
    f4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < h) {
            v[i] = cls_row ? *(const f4 *)(cls + c) + *(const f4 *)(pos0 + c) : *(const f4 *)(xr + c);
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else {
            v[i] = (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
    if (w) {        // same arithmetic as layernorm_kernel (two-pass mean / variance, y = (x - mean) scale w + b)
        const float mean = wave_sum(sum) / (float)h;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int c = (i * 64 + lane) * 4;
            if (c < h) {
                v[i] = v[i] - mean;
                sq += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
        }
        const float var = wave_sum(sq) / (float)h;
        const float scale = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int c = (i * 64 + lane) * 4;
            if (c < h) {
                const f4 ww = *(const f4 *)(w + c), bb = *(const f4 *)(b + c);
                v[i] = (v[i] * scale) * ww + bb;
                if (out32) *(f4 *)(out32 + (size_t)r * ld32 + c) = v[i];
            }
        }
    }
    row_fold_prep<NV>(v, h, lane, gnext, xg + (size_t)r * ldxg, stats + r, mu_out ? mu_out + r : nullptr);
}

// ---------------------------------------------------------------------------------------------
// im2col (fp16) for the stride-P, no-padding patch convolution: col[(b,oy,ox)][(c,ky,kx)] =
// fp16(img[b][oy*P+ky][ox*P+kx][c]); two k's per thread, padded columns are zero.
// ---------------------------------------------------------------------------------------------
// IT = float (preprocessed images as the API hands them over) or half_t (the host-pointer path converts to fp16 while packing its
// pinned buffer — the same round-to-nearest-even this kernel applies, so the column matrix is bit-identical — and ships half the bytes)
template <typename IT>
__global__ void __launch_bounds__(256) im2col_kernel(const IT * __restrict__ imgs, half_t * __restrict__ col, int B, int S,
                                                     int P, int Kpad, long total2) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total2) return;
    const int kp2 = Kpad >> 1;
    const long m = i / kp2;
    const int k0 = (int)(i % kp2) * 2;
    const int G = S / P, Np = G * G, K = 3 * P * P;
    const int b = (int)(m / Np), pp = (int)(m % Np);
    const int oy = pp / G, ox = pp % G;
    _Float16 v[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int k = k0 + e;
        _Float16 val = (_Float16)0.f;
        if (k < K) {
            const int c = k / (P * P), rem = k % (P * P);
            const int ky = rem / P, kx = rem % P;
            val = (_Float16)imgs[(size_t)b * S * S * 3 + 3 * ((size_t)(oy * P + ky) * S + (ox * P + kx)) + c];
        }
        v[e] = val;
    }
    const h2 o = (h2){v[0], v[1]};
    *(uint32_t *)(col + (size_t)m * Kpad + k0) = __builtin_bit_cast(uint32_t, o);
}


// Coalesced form for even P: one thread per (patch, ky, pixel pair).  It reads 6 consecutive elements of the interleaved
// image row (RGB RGB) and writes one half2 into each of the three channel planes of the patch row, so a wave reads
// 1.5 KB (0.75 KB for fp16 input) contiguous and writes three 256-byte runs (the scalar kernel above strides the reads by 12 bytes).
template <typename IT>
__global__ void __launch_bounds__(256) im2col_rows_kernel(const IT * __restrict__ imgs, half_t * __restrict__ col, int S, int P,
                                                          int Kpad, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int hp = P >> 1;
    const int xp = (int)(i % hp);
    const long r = i / hp;
    const int ky = (int)(r % P);
    const long m = r / P;
    const int G = S / P, Np = G * G;
    const int b = (int)(m / Np), pp = (int)(m % Np);
    const int oy = pp / G, ox = pp % G;
    const IT * src = imgs + (size_t)b * S * S * 3 + 3 * ((size_t)(oy * P + ky) * S + (ox * P + 2 * xp));
    h2 rr, gg, bb;
    if constexpr (sizeof(IT) == 4) {
        const float2 a = *(const float2 *)(src), c = *(const float2 *)(src + 2), e = *(const float2 *)(src + 4);   // r0 g0 | b0 r1 | g1 b1
        rr = (h2){(_Float16)a.x, (_Float16)c.y}; gg = (h2){(_Float16)a.y, (_Float16)e.x}; bb = (h2){(_Float16)c.x, (_Float16)e.y};
    } else {
        const h2 a = *(const h2 *)(src), c = *(const h2 *)(src + 2), e = *(const h2 *)(src + 4);
        rr = (h2){a[0], c[1]}; gg = (h2){a[1], e[0]}; bb = (h2){c[0], e[1]};
    }
    half_t * dst = col + (size_t)m * Kpad + ky * P + 2 * xp;
    *(uint32_t *)(dst) = __builtin_bit_cast(uint32_t, rr);
    *(uint32_t *)(dst + P * P) = __builtin_bit_cast(uint32_t, gg);
    *(uint32_t *)(dst + 2 * P * P) = __builtin_bit_cast(uint32_t, bb);
}

__global__ void __launch_bounds__(256) cls_rows_kernel(float * __restrict__ x, const float * __restrict__ cls,
                                                       const float * __restrict__ pos, int B, int T, int h) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * h) return;
    const int b = i / h, c = i % h;
    x[(size_t)b * T * h + c] = cls[c] + pos[c];
}

// ---------------------------------------------------------------------------------------------
// Text embedding: x[row] = dequantize_row(token_embd[id]) + pos[t]   (ggml_get_rows dequantises the
// embedding tables exactly; reference clip.cpp:1059-1061).  Raw ggml block layouts are read byte-wise
// (18/20/22/24/34-byte blocks are only 2-byte aligned).  One workgroup per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_h(const uint8_t * p) {
    const uint16_t u = (uint16_t)p[0] | ((uint16_t)p[1] << 8);
    return (float)__builtin_bit_cast(_Float16, u);
}

__device__ float dequant_elem(const uint8_t * row, int type, int k) {
    const int ib = k >> 5, j = k & 31;
    switch (type) {
    case 0: return ((const float *)row)[k];
    case 1: return ld_h(row + 2 * k);
    case 2: {  // q4_0
        const uint8_t * blk = row + ib * 18;
        const uint8_t q = blk[2 + (j & 15)];
        return (float)((j < 16 ? (q & 0x0F) : (q >> 4)) - 8) * ld_h(blk);
    }
    case 3: {  // q4_1
        const uint8_t * blk = row + ib * 20;
        const uint8_t q = blk[4 + (j & 15)];
        return (float)(j < 16 ? (q & 0x0F) : (q >> 4)) * ld_h(blk) + ld_h(blk + 2);
    }
    case 6: {  // q5_0
        const uint8_t * blk = row + ib * 22;
        const uint32_t qh = blk[2] | (blk[3] << 8) | (blk[4] << 16) | ((uint32_t)blk[5] << 24);
        const uint8_t q = blk[6 + (j & 15)];
        const int lo = j < 16 ? (q & 0x0F) : (q >> 4);
        const int hi = (qh >> j) & 1;
        return (float)((lo | (hi << 4)) - 16) * ld_h(blk);
    }
    case 7: {  // q5_1
        const uint8_t * blk = row + ib * 24;
        const uint32_t qh = blk[4] | (blk[5] << 8) | (blk[6] << 16) | ((uint32_t)blk[7] << 24);
        const uint8_t q = blk[8 + (j & 15)];
        const int lo = j < 16 ? (q & 0x0F) : (q >> 4);
        const int hi = (qh >> j) & 1;
        return (float)(lo | (hi << 4)) * ld_h(blk) + ld_h(blk + 2);
    }
    case 8: {  // q8_0
        const uint8_t * blk = row + ib * 34;
        return (float)(int8_t)blk[2 + j] * ld_h(blk);
    }
    }
    return 0.f;
}

// 4 consecutive elements k .. k + 3 (k % 4 == 0: inside one 32-weight block) of a raw ggml row: the block header is read once
// (dequant_elem re-reads scale / min / fifth bits per element: 3 dependent byte-wise loads each, 47 us per 10290 x 512 launch in r03z)
__device__ __forceinline__ f4 dequant4(const uint8_t * row, int type, int k) {
    const int ib = k >> 5, j = k & 31;
    f4 o;
    switch (type) {
    case 0: return *(const f4 *)((const float *)row + k);
    case 1: {
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = ld_h(row + 2 * (k + e));
        return o;
    }
    case 2: {  // q4_0
        const uint8_t * blk = row + ib * 18;
        const float d = ld_h(blk);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint8_t q = blk[2 + ((j + e) & 15)];
            o[e] = (float)((j < 16 ? (q & 0x0F) : (q >> 4)) - 8) * d;
        }
        return o;
    }
    case 3: {  // q4_1
        const uint8_t * blk = row + ib * 20;
        const float d = ld_h(blk), m = ld_h(blk + 2);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint8_t q = blk[4 + ((j + e) & 15)];
            o[e] = (float)(j < 16 ? (q & 0x0F) : (q >> 4)) * d + m;
        }
        return o;
    }
    case 6: {  // q5_0
        const uint8_t * blk = row + ib * 22;
        const float d = ld_h(blk);
        const uint32_t qh = blk[2] | (blk[3] << 8) | (blk[4] << 16) | ((uint32_t)blk[5] << 24);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint8_t q = blk[6 + ((j + e) & 15)];
            const int lo = j < 16 ? (q & 0x0F) : (q >> 4);
            o[e] = (float)((lo | (int)(((qh >> (j + e)) & 1u) << 4)) - 16) * d;   // (int): unsigned arithmetic would wrap below 16
        }
        return o;
    }
    case 7: {  // q5_1
        const uint8_t * blk = row + ib * 24;
        const float d = ld_h(blk), m = ld_h(blk + 2);
        const uint32_t qh = blk[4] | (blk[5] << 8) | (blk[6] << 16) | ((uint32_t)blk[7] << 24);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint8_t q = blk[8 + ((j + e) & 15)];
            const int lo = j < 16 ? (q & 0x0F) : (q >> 4);
            o[e] = (float)(lo | (int)(((qh >> (j + e)) & 1u) << 4)) * d + m;
        }
        return o;
    }
    case 8: {  // q8_0
        const uint8_t * blk = row + ib * 34;
        const float d = ld_h(blk);
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (float)(int8_t)blk[2 + j + e] * d;
        return o;
    }
    }
    return (f4){0.f, 0.f, 0.f, 0.f};
}

// One wave per row (4 rows per workgroup), the row in registers; with gnext != nullptr the kernel is also the entry of the
// LayerNorm-folded layer chain (row_fold_prep): xg = fp16(x gamma_next) and the whole-row statistics for the first q/k/v GEMM.
template <int NV>
__global__ void __launch_bounds__(256) text_embed_kernel(const int32_t * __restrict__ ids, const int * __restrict__ seq_start, int nseq, int rows,
                                                         const uint8_t * __restrict__ tok, int tok_type, size_t tok_row_bytes,
                                                         const float * __restrict__ pos, int h, float * __restrict__ x,
                                                         const float * __restrict__ gnext, half_t * __restrict__ xg, int ldxg,
                                                         float2 * __restrict__ stats, float * __restrict__ mu_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // position within its sequence: 64-ary search of seq_start (nseq + 1 entries, ascending) — the 64 lanes probe 64 pivots per step, so
    // 256 sequences take 2 dependent round trips (a binary search took 8, + 1 for the start: the launch was latency-, not bandwidth-bound)
    const int id = ids[row];
    int lo = 0, hi = nseq, start = seq_start[0];
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + (lane + 1) * step;
        const int val = idx < hi ? seq_start[idx] : 0x7fffffff;
        const int cnt = __popcll(__ballot(val <= row));               // pivots are ascending: the first cnt of them are <= row
        if (cnt > 0) start = __shfl(val, cnt - 1, 64);
        const int nlo = lo + cnt * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    const int t = row - start;
    const uint8_t * trow = tok + (size_t)id * tok_row_bytes;
    f4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < h) {
            const f4 pe = *(const f4 *)(pos + (size_t)t * h + c);
            v[i] = pe + dequant4(trow, tok_type, c);          // (same products and sums as dequant_elem, element by element)
            *(f4 *)(x + (size_t)row * h + c) = v[i];
        } else {
            v[i] = (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
    if (gnext) row_fold_prep<NV>(v, h, lane, gnext, xg + (size_t)row * ldxg, stats + row, mu_out ? mu_out + row : nullptr);
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) l2norm_kernel(const float * __restrict__ v, float * __restrict__ out, int rows, int n,
                                                     int normalize) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float * vr = v + (size_t)r * n;
    float sq = 0.f;
    for (int c = lane; c < n; c += 64) sq += vr[c] * vr[c];
    sq = wave_sum(sq);
    const float inv = normalize ? 1.0f / sqrtf(sq) : 1.0f;
    for (int c = lane; c < n; c += 64) out[(size_t)r * n + c] = vr[c] * inv;
}

__global__ void __launch_bounds__(256) f32_to_f16_kernel(const float * __restrict__ src, int lds, half_t * __restrict__ dst, int ldd,
                                                         int rows, int cols, int cols_pad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols_pad) return;
    const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
    dst[(size_t)r * ldd + c] = c < cols ? (_Float16)src[(size_t)r * lds + c] : (_Float16)0.f;
}
__global__ void __launch_bounds__(256) f16_to_f32_kernel(const half_t * __restrict__ src, int lds, float * __restrict__ dst, int ldd,
                                                         int rows, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)r * ldd + c] = (float)src[(size_t)r * lds + c];
}

}  // namespace

void launch_layernorm(const float * x, int ldx, const int * in_rows, int in_row_mul, const float * w, const float * b, float eps,
                      int rows, int h, half_t * out16, int ld16, float * out32, int ld32, hipStream_t stream) {
    if (rows <= 0) return;
    const dim3 block(256);
    const bool two = rows >= 4096;       // large row counts: two rows per wave (more bytes in flight); small ones keep the wider grid
    const dim3 grid(two ? (rows + 7) / 8 : (rows + 3) / 4);
#define CLIPAMD_LN(NV)                                                                                                               \
    if (two) hipLaunchKernelGGL((layernorm_kernel<NV, 2>), grid, block, 0, stream, x, ldx, in_rows, in_row_mul, w, b, eps, rows, h, out16, ld16, out32, ld32); \
    else hipLaunchKernelGGL((layernorm_kernel<NV, 1>), grid, block, 0, stream, x, ldx, in_rows, in_row_mul, w, b, eps, rows, h, out16, ld16, out32, ld32)
    if (h <= 256) { CLIPAMD_LN(1); }
    else if (h <= 512) { CLIPAMD_LN(2); }
    else if (h <= 768) { CLIPAMD_LN(3); }
    else if (h <= 1024) { CLIPAMD_LN(4); }
    else if (h <= 1280) { CLIPAMD_LN(5); }
    else { CLIPAMD_LN(8); }   // h <= 2048
#undef CLIPAMD_LN
}

template <typename IT>
static void launch_im2col_t(const IT * imgs, half_t * col, int B, int S, int P, int Kpad, hipStream_t stream) {
    const int G = S / P;
    const long total2 = (long)B * G * G * (Kpad / 2);
    if (total2 <= 0) return;
    if (P % 2 == 0 && Kpad == 3 * P * P && S % 2 == 0) {   // no K padding to zero-fill, aligned pixel pairs
        const long total = (long)B * G * G * P * (P / 2);
        hipLaunchKernelGGL(im2col_rows_kernel<IT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, imgs, col, S, P, Kpad, total);
        return;
    }
    hipLaunchKernelGGL(im2col_kernel<IT>, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, stream, imgs, col, B, S, P, Kpad, total2);
}

void launch_im2col(const void * imgs, bool imgs_f16, half_t * col, int B, int S, int P, int Kpad, hipStream_t stream) {
    if (imgs_f16) launch_im2col_t<half_t>((const half_t *)imgs, col, B, S, P, Kpad, stream);
    else launch_im2col_t<float>((const float *)imgs, col, B, S, P, Kpad, stream);
}

void launch_cls_rows(float * x, const float * class_embd, const float * pos, int B, int T, int h, hipStream_t stream) {
    if (B <= 0) return;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * h + 255) / 256), dim3(256), 0, stream, x, class_embd, pos, B, T, h);
}


// ---------------------------------------------------------------------------------------------
// Zero-shot scoring (reference clip_similarity_score + softmax_with_sorting, clip.cpp:1525-1532,1591-1622,
// as composed by clip_zero_shot_label_image :1624-1659 and tests/benchmark.cpp:114-160): one workgroup per image.
//   sims[j]  = sum_i img[i]*txt[j][i]      sequential fp32 accumulation in i order (bit-identical to the host loop)
//   e[j]     = (float)(exp((double)sims[j]) + 1e-9);  p[j] = (float)(e[j] / sum_j e[j])   (sum in double)
//   output   = p sorted descending (ties: lower label index first = the host's stable sort) + the label indices.
// Sort: bitonic network over (p, index) pairs in LDS, n padded to a power of two (<= 8192).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) zero_shot_kernel(const float * img, const float * txt, int n, int dim, int npad, float * scores, int * indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zs_smem[];
    float * key = (float *)zs_smem;            // [npad]
    int * idx = (int *)(key + npad);           // [npad]
    float * iv = (float *)(idx + npad);        // [dim]
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < dim; i += 256) iv[i] = img[(size_t)b * dim + i];
    __syncthreads();
    double part = 0.0;
    for (int j = tid; j < npad; j += 256) {
        float e = -1.0f;                       // padding sorts last (every real probability is > 0)
        if (j < n) {
            const float * t = txt + (size_t)j * dim;
            float dot = 0.0f;
            for (int i = 0; i < dim; i++) dot += iv[i] * t[i];
            e = (float)(exp((double)dot) + 1e-9);
            part += (double)e;
        }
        key[j] = e;
        idx[j] = j;
    }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double sum = red[0];
    for (int j = tid; j < n; j += 256) key[j] = (float)((double)key[j] / sum);
    __syncthreads();
    // descending bitonic sort; order relation: (a before b) <=> key_a > key_b || (key_a == key_b && idx_a < idx_b)
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npad; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const float ka = key[i], kb = key[l];
                    const int ia = idx[i], ib = idx[l];
                    const bool a_first = ka > kb || (ka == kb && ia < ib);
                    const bool up = (i & k) == 0;          // this sub-sequence is sorted "first elements first"
                    if (up ? !a_first : a_first) { key[i] = kb; key[l] = ka; idx[i] = ib; idx[l] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < n; j += 256) {
        scores[(size_t)b * n + j] = key[j];
        indices[(size_t)b * n + j] = idx[j];
    }
}

static size_t raw_row_bytes(int type, int k) {
    switch (type) {
    case 0: return (size_t)k * 4;
    case 1: return (size_t)k * 2;
    case 2: return (size_t)(k / 32) * 18;
    case 3: return (size_t)(k / 32) * 20;
    case 6: return (size_t)(k / 32) * 22;
    case 7: return (size_t)(k / 32) * 24;
    case 8: return (size_t)(k / 32) * 34;
    }
    return 0;
}

void launch_text_embed(const int32_t * ids, const int * seq_start, int nseq, int rows, const void * tok_raw, int tok_type,
                       const float * pos, int h, float * x, hipStream_t stream, const float * gamma_next, half_t * xg, int ldxg, float2 * stats, float * mu_out) {
    if (rows <= 0) return;
    const dim3 grid((rows + 3) / 4), block(256);
#define CLIPAMD_TE(NV)                                                                                                              \
    hipLaunchKernelGGL((text_embed_kernel<NV>), grid, block, 0, stream, ids, seq_start, nseq, rows, (const uint8_t *)tok_raw, tok_type, \
                       raw_row_bytes(tok_type, h), pos, h, x, gamma_next, xg, ldxg, stats, mu_out)
    if (h <= 256) { CLIPAMD_TE(1); }
    else if (h <= 512) { CLIPAMD_TE(2); }
    else if (h <= 768) { CLIPAMD_TE(3); }
    else if (h <= 1024) { CLIPAMD_TE(4); }
    else if (h <= 1280) { CLIPAMD_TE(5); }
    else { CLIPAMD_TE(8); }   // h <= 2048
#undef CLIPAMD_TE
}

void launch_layernorm_prep(const float * x, int ldx, const float * w, const float * b, float eps, int rows, int h, float * out32, int ld32,
                           const float * gamma_next, half_t * xg, int ldxg, float2 * stats, hipStream_t stream, float * mu_out,
                           const float * class_embd, const float * pos0, int T) {
    if (rows <= 0) return;
    const dim3 grid((rows + 3) / 4), block(256);
#define CLIPAMD_LNP(NV)                                                                                                             \
    hipLaunchKernelGGL((layernorm_prep_kernel<NV>), grid, block, 0, stream, x, ldx, w, b, eps, rows, h, out32, ld32, gamma_next, xg, ldxg, stats, mu_out, class_embd, pos0, T > 0 ? T : 1)
    if (h <= 256) { CLIPAMD_LNP(1); }
    else if (h <= 512) { CLIPAMD_LNP(2); }
    else if (h <= 768) { CLIPAMD_LNP(3); }
    else if (h <= 1024) { CLIPAMD_LNP(4); }
    else if (h <= 1280) { CLIPAMD_LNP(5); }
    else { CLIPAMD_LNP(8); }   // h <= 2048
#undef CLIPAMD_LNP
}

// Pooled rows out of the last layer (forward.cpp pooled_tail): xp[r] = x[src(r)] (f32) and ap[r] = a[src(r)] (fp16), src(r) = in_rows[r] or
// r * in_row_mul — the CLS rows b * T of the vision tower (reference clip.cpp:1426-1431), the last-token rows of the text tower (:1154-1155).
namespace {
__global__ void __launch_bounds__(256) gather_rows_kernel(const float * __restrict__ x, const half_t * __restrict__ a, const int * __restrict__ in_rows,
                                                          int in_row_mul, int rows, int h, float * __restrict__ xp, half_t * __restrict__ ap) {
    const int r = blockIdx.x;
    const long src = in_rows ? (long)in_rows[r] : (long)r * in_row_mul;
    for (int c = threadIdx.x * 4; c < h; c += 256 * 4) {
        *(f4 *)(xp + (size_t)r * h + c) = *(const f4 *)(x + (size_t)src * h + c);
        if (a) *(uint2 *)(ap + (size_t)r * h + c) = *(const uint2 *)(a + (size_t)src * h + c);      // (null: only the f32 rows)
    }
}
}  // namespace
void launch_gather_rows(const float * x, const half_t * a, const int * in_rows, int in_row_mul, int rows, int h, float * xp, half_t * ap, hipStream_t stream) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, stream, x, a, in_rows, in_row_mul, rows, h, xp, ap);
}

// The ragged-batch metadata of a text call (sequence starts, last-token rows) from a pinned, device-mapped host slot into the workspace,
// by ONE workgroup; when the slot has been read it stamps `done` (mapped host memory) so that the host can re-use the slot without a HIP
// event: hipEventRecord per call made the runtime stall once for ~35 ms after a few hundred to ~1000 calls (profiles/r03_step_spikes.txt).
namespace {
__global__ void __launch_bounds__(256) meta_upload_kernel(const int * __restrict__ src, int * __restrict__ seq, int * __restrict__ last, int n_texts,
                                                          unsigned * done, unsigned stamp) {
    for (int i = threadIdx.x; i < 2 * n_texts + 1; i += 256) {
        const int v = src[i];
        if (i <= n_texts) seq[i] = v; else last[i - n_texts - 1] = v;
    }
    __syncthreads();                      // every read of the slot has returned
    if (threadIdx.x == 0) __hip_atomic_store(done, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
void launch_meta_upload(const int * src_mapped, int * seq, int * last, int n_texts, unsigned * done_mapped, unsigned stamp, hipStream_t stream) {
    hipLaunchKernelGGL(meta_upload_kernel, dim3(1), dim3(256), 0, stream, src_mapped, seq, last, n_texts, done_mapped, stamp);
}

void launch_l2norm(const float * v, float * out, int rows, int n, bool normalize, hipStream_t stream) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, v, out, rows, n, normalize ? 1 : 0);
}

bool launch_zero_shot(const float * img, int B, const float * txt, int n, int dim, float * scores, int * indices, hipStream_t stream) {
    if (B <= 0 || n <= 0) return true;
    int npad = 1;
    while (npad < n) npad <<= 1;
    if (npad > 8192) return false;
    const size_t smem = (size_t)npad * 8 + (size_t)dim * 4;
    static unsigned long long lds_ok = 0;
    opt_in_dynamic_lds(zero_shot_kernel, 8192 * 8 + 4096 * 4, lds_ok);
    if (dim > 4096) return false;
    hipLaunchKernelGGL(zero_shot_kernel, dim3(B), dim3(256), smem, stream, img, txt, n, dim, npad, scores, indices);
    return true;
}

void launch_f32_to_f16(const float * src, int lds, half_t * dst, int ldd, int rows, int cols, int cols_pad,
                       hipStream_t stream) {
    const long total = (long)rows * cols_pad;
    if (total <= 0) return;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, lds, dst, ldd, rows,
                       cols, cols_pad);
}
void launch_f16_to_f32(const half_t * src, int lds, float * dst, int ldd, int rows, int cols, hipStream_t stream) {
    const long total = (long)rows * cols;
    if (total <= 0) return;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, lds, dst, ldd, rows,
                       cols);
}

}  // namespace clipamd
