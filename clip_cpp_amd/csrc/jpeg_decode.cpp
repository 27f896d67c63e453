// jpeg_decode.cpp — baseline + progressive Huffman JPEG decoder for clip_image_load_from_file.
//
// The reference decodes images with its vendored stb_image (reference clip.cpp:709-726 -> stbi_load(..., 3)).
// stb_image is third-party source and is not copied; this is an independent implementation of ITU-T T.81 that
// reproduces the choices that affect PIXEL VALUES the way the reference's decoder makes them, so that the pixels
// handed to clip_image_preprocess are identical (checked against the reference's own stb build in
// tests/test_image_io.py via oracle/_ref/libstb_ref.so, when /root/reference is present):
//   * integer "slow" IDCT of Loeffler/Ligtenberg/Moschytz with 12-bit constants, two passes (>>10, >>17),
//   * chroma up-sampling: 2x horizontal / vertical / both with the 3:1 "triangle" filter, nearest for other factors,
//   * YCbCr -> RGB in 20-bit fixed point (1.40200, 0.34414 (truncated to 16 bits), 0.71414, 1.77200),
//   * grey-scale -> RGB replication; four-component files: CMYK / YCCK per the Adobe APP14 transform (else YCbCr + ignored channel).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace clipamd {

namespace {

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                   6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                   // tail so that corrupt run lengths stay in bounds
                                   63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool present = false;
    uint8_t sym[256];
    uint8_t len[256];     // code length of symbol i (in code order)
    uint16_t code[256];
    int count = 0;
    // canonical decode tables
    int maxcode[18];      // largest code of length l, left-aligned to 16 bits, +1
    int delta[17];        // first symbol index - first code
    uint8_t fast[512];    // 9-bit lookahead -> symbol index, 255 = miss

    bool build(const uint8_t * counts, const uint8_t * symbols) {
        int k = 0;
        for (int l = 1; l <= 16; l++)
            for (int i = 0; i < counts[l - 1]; i++) {
                if (k >= 256) return false;
                len[k++] = (uint8_t)l;
            }
        count = k;
        memcpy(sym, symbols, (size_t)k);
        int c = 0;
        k = 0;
        for (int l = 1; l <= 16; l++) {
            delta[l] = k - c;
            if (k < count && len[k] == l) {
                while (k < count && len[k] == l) code[k++] = (uint16_t)c++;
                if (c - 1 >= (1 << l)) return false;
            }
            maxcode[l] = c << (16 - l);
            c <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memset(fast, 255, sizeof fast);
        for (int i = 0; i < count; i++) {
            if (len[i] <= 9) {
                const int base = code[i] << (9 - len[i]);
                for (int j = 0; j < (1 << (9 - len[i])); j++) fast[base + j] = (uint8_t)i;
            }
        }
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;           // current scan's tables
    int dc_pred = 0;
    int bw = 0, bh = 0;           // blocks per row / column (padded to MCU)
    int pw = 0, ph = 0;           // plane size in pixels (bw*8, bh*8)
    std::vector<int16_t> coef;    // progressive: bw*bh*64
    std::vector<uint8_t> plane;   // decoded samples
};

struct Decoder {
    const uint8_t * p;
    const uint8_t * end;
    std::string err;

    int width = 0, height = 0, ncomp = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    Component comp[4];
    uint16_t qt[4][64];
    Huff hdc[4], hac[4];
    int restart_interval = 0;
    bool saw_adobe = false, saw_jfif = false;
    int adobe_transform = -1;   // APP14 colour transform: 0 = samples stored as they are (RGB / CMYK), 1 = YCbCr, 2 = YCCK
    int rgb_ids = 0;            // components whose id spells 'R', 'G', 'B' in that order

    // bit reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    int marker = 0;      // pending marker hit inside entropy data
    bool nomore = false;
    int eobrun = 0;
    // scan params
    int ss = 0, se = 63, ah = 0, al = 0;

    bool fail(const char * m) { if (err.empty()) err = m; return false; }

    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    void fill() {
        while (bitcnt <= 24) {
            int b = nomore ? 0 : get8();
            if (b == 0xFF) {
                int c = get8();
                while (c == 0xFF) c = get8();
                if (c != 0) {
                    marker = c;
                    nomore = true;
                    b = 0;
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) fill();
        const uint32_t v = bitbuf >> (32 - n);
        bitbuf <<= n;
        bitcnt -= n;
        return (int)v;
    }
    int getbit() { return getbits(1); }
    // receive + extend
    int receive_extend(int n) {
        if (n == 0) return 0;
        const int v = getbits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode_sym(const Huff & h) {
        if (bitcnt < 16) fill();
        const int look = (int)(bitbuf >> 23);
        const int f = h.fast[look];
        if (f != 255) {
            const int l = h.len[f];
            if (l > bitcnt) return -1;
            bitbuf <<= l;
            bitcnt -= l;
            return h.sym[f];
        }
        const int top = (int)(bitbuf >> 16);
        int l = 10;
        for (; l <= 16; l++)
            if (top < h.maxcode[l]) break;
        if (l > 16 || l > bitcnt) return -1;
        const int idx = (int)(bitbuf >> (32 - l)) + h.delta[l];
        if (idx < 0 || idx >= h.count) return -1;
        bitbuf <<= l;
        bitcnt -= l;
        return h.sym[idx];
    }
    void reset_entropy() {
        bitbuf = 0;
        bitcnt = 0;
        nomore = false;
        marker = 0;
        eobrun = 0;
        for (int i = 0; i < 4; i++) comp[i].dc_pred = 0;
    }

    // ---- block decoders ----
    // the DC term has to fit 16 bits after scaling (both factors taken as 16-bit values, as the reference's check takes them); a predictor
    // that would leave the int range is refused as well
    static bool dc_fits(int dc, int scale) {
        const int a = (int16_t)dc, b = (int16_t)scale;
        if (b == 0 || b == -1) return true;
        if ((a >= 0) == (b >= 0)) return a <= 32767 / b;
        return b < 0 ? a <= -32768 / b : a >= -32768 / b;
    }
    static bool pred_fits(int pred, int diff) {
        if ((pred >= 0) != (diff >= 0)) return true;
        return pred < 0 ? pred >= INT32_MIN - diff : pred <= INT32_MAX - diff;
    }
    bool block_baseline(int16_t * data, Component & c) {
        memset(data, 0, 64 * sizeof(int16_t));
        const int t = decode_sym(hdc[c.td]);
        if (t < 0 || t > 15) return fail("bad huffman code");
        const int diff = t ? receive_extend(t) : 0;
        if (!pred_fits(c.dc_pred, diff)) return fail("bad DC delta");
        c.dc_pred += diff;
        if (!dc_fits(c.dc_pred, qt[c.tq][0])) return fail("DC term out of range");
        data[0] = (int16_t)c.dc_pred;
        const Huff & ac = hac[c.ta];
        int k = 1;
        while (k < 64) {
            const int rs = decode_sym(ac);
            if (rs < 0) return fail("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                data[kZigzag[k++]] = (int16_t)receive_extend(s);
            }
        }
        return true;
    }
    bool block_prog_dc(int16_t * data, Component & c) {
        if (ah == 0) {
            memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode_sym(hdc[c.td]);
            if (t < 0 || t > 15) return fail("bad huffman code");
            const int diff = t ? receive_extend(t) : 0;
            if (!pred_fits(c.dc_pred, diff)) return fail("bad DC delta");
            c.dc_pred += diff;
            if (!dc_fits(c.dc_pred, 1 << al)) return fail("DC term out of range");
            data[0] = (int16_t)(c.dc_pred * (1 << al));
        } else {
            if (getbit()) data[0] = (int16_t)(data[0] + (1 << al));
        }
        return true;
    }
    bool block_prog_ac(int16_t * data, const Huff & ac) {
        if (ah == 0) {
            if (eobrun) { eobrun--; return true; }
            int k = ss;
            while (k <= se) {
                const int rs = decode_sym(ac);
                if (rs < 0) return fail("bad huffman code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1 << r);
                        if (r) eobrun += getbits(r);
                        eobrun--;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    data[kZigzag[k++]] = (int16_t)(receive_extend(s) * (1 << al));
                }
            }
        } else {
            const int bit = 1 << al;
            if (eobrun) {
                eobrun--;
                for (int k = ss; k <= se; k++) {
                    int16_t * q = &data[kZigzag[k]];
                    if (*q != 0 && getbit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
                }
            } else {
                int k = ss;
                do {
                    const int rs = decode_sym(ac);
                    if (rs < 0) return fail("bad huffman code");
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eobrun = (1 << r) - 1;
                            if (r) eobrun += getbits(r);
                            r = 64;   // force end of block
                        }
                    } else {
                        if (s != 1) return fail("bad huffman code");
                        s = getbit() ? bit : -bit;
                    }
                    while (k <= se) {
                        int16_t * q = &data[kZigzag[k++]];
                        if (*q != 0) {
                            if (getbit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
                        } else {
                            if (r == 0) { *q = (int16_t)s; break; }
                            r--;
                        }
                    }
                } while (k <= se);
            }
        }
        return true;
    }

    // ---- IDCT (LL&M, 12-bit constants) ----
    static inline uint8_t clamp8(int x) { return (uint8_t)((unsigned)x > 255 ? (x < 0 ? 0 : 255) : x); }
    static inline int fx(double v) { return (int)(v * 4096 + 0.5); }

    // One 1-D pass over eight 16-bit inputs.  The arithmetic is the scalar LL&M flow regrouped the way the reference's SSE2 kernel groups it
    // (the same integers for every valid stream): the four input sums s0 +- s4, s1 + s7, s3 + s5 are formed in 16 bits and WRAP, every
    // product and sum behind them is 32-bit (wrapping), and the eight results are shifted and SATURATED back to 16 bits.  Only corrupt
    // streams (coefficients x quantisers beyond 16 bits) ever reach the wrap / saturation; with them the pixels still equal that decoder's.
    static inline int sat16(int x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }
    static inline int32_t wrap_mul(int a, int b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
    static inline int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
    static inline int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
    static void idct1d(const int s[8], int32_t bias, int shift, int out[8]) {
        const int c0541 = fx(0.5411961), cm1847 = fx(-1.847759065), c0765 = fx(0.765366865), c1175 = fx(1.175875602), cm0899 = fx(-0.899976223),
                  cm2562 = fx(-2.562915447), cm1961 = fx(-1.961570560), c0298 = fx(0.298631336), c3072 = fx(3.072711026), cm0390 = fx(-0.390180644),
                  c2053 = fx(2.053119869), c1501 = fx(1.501321110);
        const int e04 = (int16_t)(s[0] + s[4]), d04 = (int16_t)(s[0] - s[4]), a17 = (int16_t)(s[1] + s[7]), a35 = (int16_t)(s[3] + s[5]);
        // even part
        const int32_t t2e = wrap_add(wrap_mul(s[2], c0541), wrap_mul(s[6], c0541 + cm1847));
        const int32_t t3e = wrap_add(wrap_mul(s[2], c0541 + c0765), wrap_mul(s[6], c0541));
        const int32_t t0e = wrap_mul(e04, 4096), t1e = wrap_mul(d04, 4096);
        const int32_t x0 = wrap_add(t0e, t3e), x3 = wrap_sub(t0e, t3e), x1 = wrap_add(t1e, t2e), x2 = wrap_sub(t1e, t2e);
        // odd part
        const int32_t y0 = wrap_add(wrap_mul(s[7], cm1961 + c0298), wrap_mul(s[3], cm1961));
        const int32_t y2 = wrap_add(wrap_mul(s[7], cm1961), wrap_mul(s[3], cm1961 + c3072));
        const int32_t y1 = wrap_add(wrap_mul(s[5], cm0390 + c2053), wrap_mul(s[1], cm0390));
        const int32_t y3 = wrap_add(wrap_mul(s[5], cm0390), wrap_mul(s[1], cm0390 + c1501));
        const int32_t y4 = wrap_add(wrap_mul(a17, c1175 + cm0899), wrap_mul(a35, c1175));
        const int32_t y5 = wrap_add(wrap_mul(a17, c1175), wrap_mul(a35, c1175 + cm2562));
        const int32_t x4 = wrap_add(y0, y4), x5 = wrap_add(y1, y5), x6 = wrap_add(y2, y5), x7 = wrap_add(y3, y4);
        const int32_t xe[4] = {x0, x1, x2, x3}, xo[4] = {x7, x6, x5, x4};
        for (int k = 0; k < 4; k++) {
            const int32_t a = wrap_add(xe[k], bias);
            out[k] = sat16(wrap_add(a, xo[k]) >> shift);
            out[7 - k] = sat16(wrap_sub(a, xo[k]) >> shift);
        }
    }

    static void idct_block(uint8_t * out, int stride, const int16_t * d) {
        int val[64];
        for (int i = 0; i < 8; i++) {                      // columns: 2 extra bits of precision kept (>> 10 of 12)
            const int col[8] = {d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]};
            int o[8];
            idct1d(col, 512, 10, o);
            for (int k = 0; k < 8; k++) val[k * 8 + i] = o[k];
        }
        for (int i = 0; i < 8; i++) {                      // rows: >> 17 with the rounding and the +128 level shift in the bias
            int o[8];
            idct1d(val + i * 8, 65536 + (128 << 17), 17, o);
            uint8_t * q = out + i * stride;
            for (int k = 0; k < 8; k++) q[k] = clamp8(o[k]);
        }
    }

    // ---- segments ----
    bool read_dqt(int len) {
        while (len > 0) {
            const int q = get8();
            const int prec = q >> 4, t = q & 15;
            if (t > 3 || prec > 1) return fail("bad DQT");
            for (int i = 0; i < 64; i++) qt[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
            len -= prec ? 129 : 65;
        }
        return len == 0;
    }
    bool read_dht(int len) {
        while (len > 0) {
            const int q = get8();
            const int tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) return fail("bad DHT");
            uint8_t counts[16], syms[256];
            int n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); n += counts[i]; }
            if (n > 256) return fail("bad DHT");
            for (int i = 0; i < n; i++) syms[i] = (uint8_t)get8();
            Huff & h = tc ? hac[th] : hdc[th];
            if (!h.build(counts, syms)) return fail("bad code lengths");
            len -= 17 + n;
        }
        return len == 0;
    }
    bool read_sof(int len) {
        if (len < 6) return fail("bad SOF");
        if (get8() != 8) return fail("only 8-bit JPEG is supported");
        height = get16();
        width = get16();
        ncomp = get8();
        if (height <= 0 || width <= 0) return fail("bad image size");
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) return fail("unsupported component count");
        if (len != 6 + 3 * ncomp) return fail("bad SOF length");
        if ((size_t)width * height > ((size_t)1 << 28)) return fail("image too large");
        hmax = vmax = 1;                                   // a second SOF must not inherit the first one's maxima
        rgb_ids = 0;
        for (int i = 0; i < ncomp; i++) {
            Component & c = comp[i];
            c.id = get8();
            if (ncomp == 3 && c.id == "RGB"[i]) rgb_ids++;
            const int q = get8();
            c.h = q >> 4;
            c.v = q & 15;
            c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        for (int i = 0; i < ncomp; i++)
            if (hmax % comp[i].h || vmax % comp[i].v) return fail("unsupported sampling factors");
        mcux = (width + 8 * hmax - 1) / (8 * hmax);
        mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Component & c = comp[i];
            c.bw = mcux * c.h;
            c.bh = mcuy * c.v;
            c.pw = c.bw * 8;
            c.ph = c.bh * 8;
            c.plane.assign((size_t)c.pw * c.ph, 0);
            if (progressive) c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        return true;
    }

    // number of blocks a NON-interleaved scan covers for component c
    static int ceil_div(int a, int b) { return (a + b - 1) / b; }

    bool handle_restart(int & todo) {
        if (restart_interval == 0) return true;
        if (--todo > 0) return true;
        if (bitcnt < 24) fill();
        if (marker >= 0xD0 && marker <= 0xD7) {
            reset_entropy();
        } else if (marker != 0) {
            return true;   // some other marker: scan ends, caller sees it
        } else {
            // look for the RSTn marker in the byte stream
            if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; reset_entropy(); }
            else return true;
        }
        todo = restart_interval;
        return true;
    }

    bool read_sos() {
        const int len = get16();
        const int ns = get8();
        if (ns < 1 || ns > ncomp || len != 6 + 2 * ns) return fail("bad SOS");
        int order[4];
        for (int i = 0; i < ns; i++) {
            const int id = get8(), q = get8();
            int which = -1;
            for (int k = 0; k < ncomp; k++)
                if (comp[k].id == id) which = k;
            if (which < 0) return fail("bad SOS component");
            comp[which].td = q >> 4;
            comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return fail("bad SOS table");
            order[i] = which;
        }
        ss = get8();
        se = get8();
        const int a = get8();
        ah = a >> 4;
        al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return fail("bad SOS");
        } else {
            if (ss != 0 || ah != 0 || al != 0) return fail("bad SOS");
            se = 63;
        }
        reset_entropy();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        int16_t tmp[64];
        if (ns == 1) {
            Component & c = comp[order[0]];
            const int w = ceil_div(ceil_div(width * c.h, hmax), 8), h = ceil_div(ceil_div(height * c.v, vmax), 8);
            for (int by = 0; by < h; by++)
                for (int bx = 0; bx < w; bx++) {
                    if (progressive) {
                        int16_t * d = &c.coef[((size_t)by * c.bw + bx) * 64];
                        if (ss == 0) { if (!block_prog_dc(d, c)) return false; }
                        else { if (!hac[c.ta].present) return fail("missing AC table"); if (!block_prog_ac(d, hac[c.ta])) return false; }
                    } else {
                        if (!hdc[c.td].present || !hac[c.ta].present) return fail("missing huffman table");
                        if (!block_baseline(tmp, c)) return false;
                        for (int i = 0; i < 64; i++) tmp[i] = (int16_t)(tmp[i] * qt[c.tq][i]);
                        idct_block(&c.plane[(size_t)by * 8 * c.pw + bx * 8], c.pw, tmp);
                    }
                    if (restart_interval) {
                        if (--todo <= 0) {
                            if (bitcnt < 24) fill();
                            if (!(marker >= 0xD0 && marker <= 0xD7)) return true;
                            reset_entropy();
                            todo = restart_interval;
                        }
                    }
                }
        } else {
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    for (int i = 0; i < ns; i++) {
                        Component & c = comp[order[i]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                const int bx = mx * c.h + x, by = my * c.v + y;
                                if (progressive) {
                                    if (ss != 0) return fail("interleaved AC scan");
                                    if (!block_prog_dc(&c.coef[((size_t)by * c.bw + bx) * 64], c)) return false;
                                } else {
                                    if (!hdc[c.td].present || !hac[c.ta].present) return fail("missing huffman table");
                                    if (!block_baseline(tmp, c)) return false;
                                    for (int k = 0; k < 64; k++) tmp[k] = (int16_t)(tmp[k] * qt[c.tq][k]);
                                    idct_block(&c.plane[(size_t)by * 8 * c.pw + bx * 8], c.pw, tmp);
                                }
                            }
                    }
                    if (restart_interval) {
                        if (--todo <= 0) {
                            if (bitcnt < 24) fill();
                            if (!(marker >= 0xD0 && marker <= 0xD7)) return true;
                            reset_entropy();
                            todo = restart_interval;
                        }
                    }
                }
        }
        return true;
    }

    void finish_progressive() {
        int16_t tmp[64];
        for (int i = 0; i < ncomp; i++) {
            Component & c = comp[i];
            const int w = ceil_div(ceil_div(width * c.h, hmax), 8), h = ceil_div(ceil_div(height * c.v, vmax), 8);
            for (int by = 0; by < h; by++)
                for (int bx = 0; bx < w; bx++) {
                    const int16_t * d = &c.coef[((size_t)by * c.bw + bx) * 64];
                    for (int k = 0; k < 64; k++) tmp[k] = (int16_t)(d[k] * qt[c.tq][k]);
                    idct_block(&c.plane[(size_t)by * 8 * c.pw + bx * 8], c.pw, tmp);
                }
        }
    }

    bool decode() {
        if (end - p < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG");
        p += 2;
        bool have_sof = false;
        int m = 0;
        for (;;) {
            if (marker) {
                m = marker;
                marker = 0;
            } else {
                // find next marker
                int c = get8();
                while (c != 0xFF && p < end) c = get8();
                while (c == 0xFF && p < end) c = get8();
                m = c;
                if (p >= end && m != 0xD9) return have_sof ? true : fail("truncated JPEG");
            }
            if (m == 0xD9) break;
            if (m == 0xDA) {
                if (!have_sof) return fail("SOS before SOF");
                if (!read_sos()) return false;
                if (!marker) {
                    // skip to the next marker after the entropy-coded segment
                    nomore = false;
                    while (p < end) {
                        if (p[0] == 0xFF && p + 1 < end && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7) && p[1] != 0xFF) break;
                        p++;
                    }
                    if (p + 1 < end) { marker = p[1]; p += 2; }
                    else break;
                }
                continue;
            }
            if (m >= 0xD0 && m <= 0xD7) continue;
            const int len = get16() - 2;
            if (len < 0 || p + len > end) return fail("bad segment length");
            const uint8_t * seg_end = p + len;
            switch (m) {
            case 0xDB: if (!read_dqt(len)) return fail("bad DQT"); break;
            case 0xC4: if (!read_dht(len)) return fail("bad DHT"); break;
            case 0xC0: case 0xC1: progressive = false; if (!read_sof(len)) return false; have_sof = true; break;
            case 0xC2: progressive = true; if (!read_sof(len)) return false; have_sof = true; break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("unsupported JPEG process (lossless / arithmetic)");
            case 0xDD: if (len != 2) return fail("bad DRI"); restart_interval = get16(); break;     // (len excludes the length field itself)
            case 0xE0: if (len >= 5 && !memcmp(p, "JFIF", 5)) saw_jfif = true; break;
            case 0xEE:
                if (len >= 12 && !memcmp(p, "Adobe", 6)) { saw_adobe = true; adobe_transform = p[11]; }
                break;
            default: break;
            }
            p = seg_end;
        }
        if (!have_sof) return fail("no SOF");
        if (progressive) finish_progressive();
        return true;
    }

    // ---- up-sampling + colour conversion ----
    static inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
    static inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }

    static void row_h2(uint8_t * out, const uint8_t * in, int w) {
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0];
        out[1] = div4(in[0] * 3 + in[1] + 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            const int n = 3 * in[i] + 2;
            out[i * 2 + 0] = div4(n + in[i - 1]);
            out[i * 2 + 1] = div4(n + in[i + 1]);
        }
        out[i * 2 + 0] = div4(in[w - 2] * 3 + in[w - 1] + 2);
        out[i * 2 + 1] = in[w - 1];
    }
    static void row_v2(uint8_t * out, const uint8_t * near, const uint8_t * far, int w) {
        for (int i = 0; i < w; i++) out[i] = div4(3 * near[i] + far[i] + 2);
    }
    static void row_hv2(uint8_t * out, const uint8_t * near, const uint8_t * far, int w) {
        if (w == 1) { out[0] = out[1] = div4(3 * near[0] + far[0] + 2); return; }
        int t1 = 3 * near[0] + far[0];
        out[0] = div4(t1 + 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
            out[i * 2] = div16(3 * t1 + t0 + 8);
        }
        out[w * 2 - 1] = div4(t1 + 2);
    }
    static void row_generic(uint8_t * out, const uint8_t * near, int w, int hs) {
        for (int i = 0; i < w; i++)
            for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    }

    void to_rgb(std::vector<uint8_t> & rgb) {
        rgb.resize((size_t)width * height * 3);
        struct Res { int hs, vs, ystep, ypos, wl; const uint8_t *l0, *l1; std::vector<uint8_t> buf; };
        Res rs[4];
        for (int k = 0; k < ncomp; k++) {
            Component & c = comp[k];
            Res & r = rs[k];
            r.hs = hmax / c.h;
            r.vs = vmax / c.v;
            r.ystep = r.vs >> 1;
            r.wl = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.l0 = r.l1 = c.plane.data();
            r.buf.resize((size_t)width + 8 * hmax + 16);
        }
        const int cy = (height * 1);
        (void)cy;
        for (int j = 0; j < height; j++) {
            const uint8_t * line[4];
            for (int k = 0; k < ncomp; k++) {
                Component & c = comp[k];
                Res & r = rs[k];
                const bool bot = r.ystep >= (r.vs >> 1);
                const uint8_t * near = bot ? r.l1 : r.l0;
                const uint8_t * far = bot ? r.l0 : r.l1;
                if (r.hs == 1 && r.vs == 1) line[k] = near;
                else if (r.hs == 1 && r.vs == 2) { row_v2(r.buf.data(), near, far, r.wl); line[k] = r.buf.data(); }
                else if (r.hs == 2 && r.vs == 1) { row_h2(r.buf.data(), near, r.wl); line[k] = r.buf.data(); }
                else if (r.hs == 2 && r.vs == 2) { row_hv2(r.buf.data(), near, far, r.wl); line[k] = r.buf.data(); }
                else { row_generic(r.buf.data(), near, r.wl, r.hs); line[k] = r.buf.data(); }
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.l0 = r.l1;
                    const int rows = (height * c.v + vmax - 1) / vmax;   // component height in samples
                    if (++r.ypos < rows) r.l1 += c.pw;
                }
            }
            uint8_t * o = &rgb[(size_t)j * width * 3];
            // x * y / 255 rounded, for 0 <= x, y <= 255 (the multiply the reference's decoder uses for the K channel)
            auto mul255 = [](int x, int y) { const unsigned t = (unsigned)(x * y) + 128u; return (uint8_t)((t + (t >> 8)) >> 8); };
            auto ycc = [&](int i, uint8_t * px) {
                const int yf = (line[0][i] << 20) + (1 << 19);
                const int cb = line[1][i] - 128, cr = line[2][i] - 128;
                int r = yf + cr * (((int)(1.40200f * 4096.0f + 0.5f)) << 8);
                int g = yf + cr * -(((int)(0.71414f * 4096.0f + 0.5f)) << 8) + ((cb * -(((int)(0.34414f * 4096.0f + 0.5f)) << 8)) & 0xffff0000);
                int b = yf + cb * (((int)(1.77200f * 4096.0f + 0.5f)) << 8);
                r >>= 20; g >>= 20; b >>= 20;
                px[0] = clamp8(r); px[1] = clamp8(g); px[2] = clamp8(b);
            };
            if (ncomp == 1) {
                for (int i = 0; i < width; i++) o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = line[0][i];
            } else if (ncomp == 3) {
                // stored as R, G, B when the component ids say so, or under an Adobe marker with transform 0 in a file that is not JFIF
                const bool is_rgb = rgb_ids == 3 || (saw_adobe && adobe_transform == 0 && !saw_jfif);
                for (int i = 0; i < width; i++) {
                    if (is_rgb) { o[3 * i] = line[0][i]; o[3 * i + 1] = line[1][i]; o[3 * i + 2] = line[2][i]; }
                    else ycc(i, o + 3 * i);
                }
            } else {
                // four components: Adobe transform 0 = CMYK (stored inverted: sample * K / 255), 2 = YCCK ((255 - RGB) * K / 255),
                // anything else: YCbCr + an ignored fourth channel
                const int tr = saw_adobe ? adobe_transform : -1;
                for (int i = 0; i < width; i++) {
                    uint8_t * px = o + 3 * i;
                    const int k = line[3][i];
                    if (tr == 0) { px[0] = mul255(line[0][i], k); px[1] = mul255(line[1][i], k); px[2] = mul255(line[2][i], k); }
                    else {
                        ycc(i, px);
                        if (tr == 2) { px[0] = mul255(255 - px[0], k); px[1] = mul255(255 - px[1], k); px[2] = mul255(255 - px[2], k); }
                    }
                }
            }
        }
    }
};

}  // namespace

bool decode_jpeg(const uint8_t * data, size_t size, std::vector<uint8_t> & rgb, int & nx, int & ny, std::string & err) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
    Decoder * d = new Decoder();
    d->p = data;
    d->end = data + size;
    bool ok = d->decode();
    if (ok) {
        d->to_rgb(rgb);
        nx = d->width;
        ny = d->height;
    } else {
        err = d->err.empty() ? "JPEG decode failed" : d->err;
    }
    delete d;
    return ok;
}

}  // namespace clipamd
