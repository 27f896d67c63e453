// image_formats.cpp — the remaining file formats clip_image_load_from_file accepts in the reference (clip.cpp:709-726 hands the
// file to its vendored stb_image: stbi_load(fname, &nx, &ny, &nc, 3)): every BMP flavour stb_image reads (1 / 4 / 8-bit paletted,
// 16 / 32-bit with default or BI_BITFIELDS masks, OS/2 and V4 / V5 headers), TGA (true colour, grey, colour-mapped, 15 / 16-bit, RLE),
// GIF (first frame), PSD (RGB, 8 / 16 bit, raw / PackBits) and binary PNM with any maxval.  Host side, outside the GPU hot path
// (SURVEY §2 #7); own decoders.  The pixels feed the bit-exact preprocessing, so wherever a format leaves room (5-bit channel
// scaling, un-matting of PSD alpha, what a GIF's undrawn pixels are) the arithmetic follows what the reference's decoder does for
// the same bytes — tests/test_image_io.py compares every case with that decoder (oracle/_ref/libstb_ref.so).  Also here, for
// completeness of that list: Softimage PIC and Radiance RGBE (.hdr, tone-mapped to 8 bits with gamma 2.2 as the reference does).
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace clipamd {

namespace {

constexpr int kMaxDim = 1 << 24;                      // per-side limit of the reference's decoder
constexpr size_t kMaxPixels = (size_t)1 << 28;        // this loader's own cap (image_io.cpp)

// Sequential reader over the file image.  Past the end every byte reads as 0 and skips saturate (a negative skip jumps to the
// end): truncated files then decode the way the reference's reader makes them decode, instead of being rejected half way.
struct Reader {
    const uint8_t * base, * p, * end;
    explicit Reader(const std::vector<uint8_t> & d) : base(d.data()), p(d.data()), end(d.data() + d.size()) {}
    int u8() { return p < end ? *p++ : 0; }
    int u16le() { const int a = u8(); return a | (u8() << 8); }
    uint32_t u32le() { const uint32_t a = (uint32_t)u16le(); return a | ((uint32_t)u16le() << 16); }
    int u16be() { const int a = u8(); return (a << 8) | u8(); }
    uint32_t u32be() { const uint32_t a = (uint32_t)u16be(); return (a << 16) | (uint32_t)u16be(); }
    void skip(long n) {
        if (n == 0) return;
        if (n < 0 || n > end - p) { p = end; return; }
        p += n;
    }
    long consumed() const { return (long)(p - base); }
    bool at_end() const { return p >= end; }
};

bool pixels_fit(long w, long h) { return w > 0 && h > 0 && w <= kMaxDim && h <= kMaxDim && (size_t)w * (size_t)h <= kMaxPixels; }

// ------------------------------------------------------------------------------------------------------------------
// BMP
// ------------------------------------------------------------------------------------------------------------------
int top_bit(uint32_t v) { int n = -1; while (v) { n++; v >>= 1; } return n; }
int bit_count(uint32_t v) { int n = 0; while (v) { n += (int)(v & 1); v >>= 1; } return n; }

// channel of `bits` significant bits (the mask's population count) -> 8 bits by bit replication: the masked value is first moved so that
// the mask's highest bit sits at bit 7, its top `bits` bits are then repeated downwards (5 bits abcde -> abcdeabc)
uint8_t widen_channel(uint32_t masked, int shift, int bits) {
    uint32_t v = shift < 0 ? masked << -shift : masked >> shift;
    if (bits <= 0) return 0;
    v = (v & 0xff) >> (8 - bits);
    uint32_t r = 0;
    int have = 0;
    while (have < 8) { r = (r << bits) | v; have += bits; }
    return (uint8_t)(r >> (have - 8));
}

bool decode_bmp_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    Reader r(d);
    if (r.u8() != 'B' || r.u8() != 'M') return false;
    r.skip(8);                                           // file size, two reserved words
    const int32_t data_off = (int32_t)r.u32le();
    const int32_t hsz = (int32_t)r.u32le();
    if (data_off < 0) return false;
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return false;
    int32_t w, h;
    if (hsz == 12) { w = r.u16le(); h = r.u16le(); }
    else { w = (int32_t)r.u32le(); h = (int32_t)r.u32le(); }
    if (r.u16le() != 1) return false;                   // planes
    const int bpp = r.u16le();
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
    long fixed = 14;                                     // bytes in front of the colour table that are not counted in hsz (file header, loose masks)
    auto default_masks = [&]() {                         // BI_RGB: 16 bpp = x1r5g5b5, 32 bpp = x8r8g8b8 (+ alpha byte), anything else: no masks
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; ma = 0; }
        else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; }
        else mr = mg = mb = ma = 0;
    };
    if (hsz != 12) {
        const int32_t comp = (int32_t)r.u32le();
        if (comp == 1 || comp == 2) return false;       // RLE4 / RLE8: not read by the reference either
        if (comp >= 4) return false;                    // embedded JPEG / PNG (a NEGATIVE field passes in the reference: treated below as "neither 0 nor 3")
        if (comp == 3 && bpp != 16 && bpp != 32) return false;
        r.skip(20);                                      // image size, resolution x / y, colours used / important
        if (hsz == 40 || hsz == 56) {
            if (hsz == 56) r.skip(16);
            if (bpp == 16 || bpp == 32) {
                if (comp == 0) default_masks();
                else if (comp != 3) return false;
                else {                                   // BI_BITFIELDS: three masks behind the header
                    mr = r.u32le(); mg = r.u32le(); mb = r.u32le();
                    fixed += 12;
                    if (mr == mg && mg == mb) return false;
                }
            }
        } else {                                         // V4 / V5
            mr = r.u32le(); mg = r.u32le(); mb = r.u32le(); ma = r.u32le();
            if (comp == 0) default_masks();          // (BI_BITFIELDS, and the odd negative value, keep the header's masks)
            r.skip(4 + 48);                              // colour space + endpoints / gamma
            if (hsz == 124) r.skip(16);
        }
    }
    const bool bottom_up = h > 0;
    if (h == INT32_MIN) return false;
    if (h < 0) h = -h;
    if (!pixels_fit(w, h)) return false;

    long n_pal = 0;
    if (hsz == 12) { if (bpp < 24) n_pal = (data_off - fixed - 24) / 3; }
    else if (bpp < 16) n_pal = (data_off - fixed - hsz) >> 2;
    if (n_pal == 0) {
        // no colour table: the pixel offset may leave a small gap behind the header, nothing else
        const long seen = r.consumed();
        if (seen <= 0 || seen > 1024) return false;
        if (data_off < seen || data_off - seen > 1024) return false;
        r.skip(data_off - seen);
    }
    nx = w; ny = h;
    rgb.assign((size_t)w * h * 3, 0);
    size_t z = 0;
    if (bpp < 16) {
        // (an OS/2 header makes the reference count 4 entries too few — a 2-colour table comes out as a negative count, which it takes
        // for "no entries"; the same count is used here so that the pixel offset agrees, entries that are not read are black)
        if (n_pal == 0 || n_pal > 256) return false;
        uint8_t pal[256][3];
        for (long i = 0; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = 0;
        for (long i = 0; i < n_pal; i++) {
            pal[i][2] = (uint8_t)r.u8(); pal[i][1] = (uint8_t)r.u8(); pal[i][0] = (uint8_t)r.u8();
            if (hsz != 12) (void)r.u8();
        }
        r.skip(data_off - fixed - hsz - n_pal * (hsz == 12 ? 3 : 4));
        long row_bytes;
        if (bpp == 1) row_bytes = ((long)w + 7) >> 3;
        else if (bpp == 4) row_bytes = ((long)w + 1) >> 1;
        else if (bpp == 8) row_bytes = w;
        else return false;
        const long pad = (-row_bytes) & 3;
        const int per_byte = 8 / bpp, mask = (1 << bpp) - 1;
        for (int y = 0; y < h; y++) {
            int byte = 0;
            for (int x = 0; x < w; x++) {
                if (x % per_byte == 0) byte = r.u8();
                const int idx = (byte >> (8 - bpp - (x % per_byte) * bpp)) & mask;     // leftmost pixel in the high bits
                rgb[z++] = pal[idx][0]; rgb[z++] = pal[idx][1]; rgb[z++] = pal[idx][2];
            }
            r.skip(pad);
        }
    } else {
        r.skip(data_off - fixed - hsz);
        long row_bytes = 0;
        if (bpp == 24) row_bytes = 3L * w;
        else if (bpp == 16) row_bytes = 2L * w;
        else if (bpp != 32) return false;
        const long pad = (-row_bytes) & 3;
        const bool bytes_bgr = bpp == 24 || (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0xff0000u && ma == 0xff000000u);
        int rs = 0, gs = 0, bs = 0, rc = 0, gc = 0, bc = 0;
        if (!bytes_bgr) {
            if (!mr || !mg || !mb) return false;
            rs = top_bit(mr) - 7; rc = bit_count(mr);
            gs = top_bit(mg) - 7; gc = bit_count(mg);
            bs = top_bit(mb) - 7; bc = bit_count(mb);
            if (rc > 8 || gc > 8 || bc > 8 || bit_count(ma) > 8) return false;
        }
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                if (bytes_bgr) {
                    const int b = r.u8(), g = r.u8(), rr = r.u8();
                    if (bpp == 32) (void)r.u8();
                    rgb[z++] = (uint8_t)rr; rgb[z++] = (uint8_t)g; rgb[z++] = (uint8_t)b;
                } else {
                    const uint32_t v = bpp == 16 ? (uint32_t)r.u16le() : r.u32le();
                    rgb[z++] = widen_channel(v & mr, rs, rc);
                    rgb[z++] = widen_channel(v & mg, gs, gc);
                    rgb[z++] = widen_channel(v & mb, bs, bc);
                }
            }
            r.skip(pad);
        }
    }
    if (bottom_up) {
        const size_t rb = (size_t)w * 3;
        std::vector<uint8_t> tmp(rb);
        for (int y = 0; y < h / 2; y++) {
            uint8_t * a = &rgb[rb * y], * b = &rgb[rb * (h - 1 - y)];
            memcpy(tmp.data(), a, rb); memcpy(a, b, rb); memcpy(b, tmp.data(), rb);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// TGA
// ------------------------------------------------------------------------------------------------------------------
// channels a TGA sample of `bits` bits expands to: 1 grey, 2 grey + alpha, 3 / 4 colour; 15- and 16-bit colour = 5:5:5
int tga_channels(int bits, bool grey, bool & packed555) {
    packed555 = false;
    switch (bits) {
    case 8: return 1;
    case 16: if (grey) return 2; packed555 = true; return 3;
    case 15: packed555 = true; return 3;
    case 24: return 3;
    case 32: return 4;
    default: return 0;
    }
}

bool looks_like_tga(const std::vector<uint8_t> & d) {     // TGA has no magic: plausibility of the 18-byte header, tried after every other format
    Reader r(d);
    (void)r.u8();
    const int cmap = r.u8();
    if (cmap > 1) return false;
    const int type = r.u8();
    if (cmap == 1) {
        if (type != 1 && type != 9) return false;
        r.skip(4);
        const int eb = r.u8();
        if (eb != 8 && eb != 15 && eb != 16 && eb != 24 && eb != 32) return false;
        r.skip(4);
    } else {
        if (type != 2 && type != 3 && type != 10 && type != 11) return false;
        r.skip(9);
    }
    if (r.u16le() < 1 || r.u16le() < 1) return false;
    const int bits = r.u8();
    if (cmap == 1 && bits != 8 && bits != 16) return false;
    return bits == 8 || bits == 15 || bits == 16 || bits == 24 || bits == 32;
}

void tga_555(int px, uint8_t * out) {                       // 5-bit channels to 8 bits by v * 255 / 31 (not bit replication), stored R, G, B
    out[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31);
    out[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31);
    out[2] = (uint8_t)(((px & 31) * 255) / 31);
}

bool decode_tga_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    if (!looks_like_tga(d)) return false;
    Reader r(d);
    const int id_len = r.u8();
    const int indexed = r.u8();
    int type = r.u8();
    const int pal_first = r.u16le();
    const int pal_len = r.u16le();
    const int pal_bits = r.u8();
    r.skip(4);                                              // x / y origin
    const int w = r.u16le(), h = r.u16le();
    const int bits = r.u8();
    const int desc = r.u8();
    const bool rle = type >= 8;
    if (rle) type -= 8;
    const bool bottom_up = ((desc >> 5) & 1) == 0;          // (bit 4, right-to-left, is not honoured by the reference either)
    bool packed = false;
    const int nc = indexed ? tga_channels(pal_bits, false, packed) : tga_channels(bits, type == 3, packed);
    if (!nc || !pixels_fit(w, h)) return false;
    r.skip(id_len);
    std::vector<uint8_t> pal;
    if (indexed) {
        if (pal_len == 0) return false;
        r.skip(pal_first);                                  // (bytes, as the reference skips)
        pal.assign((size_t)pal_len * nc, 0);
        if (packed) { for (int i = 0; i < pal_len; i++) tga_555(r.u16le(), &pal[(size_t)i * nc]); }
        else {
            if ((long)(r.end - r.p) < (long)pal.size()) return false;
            memcpy(pal.data(), r.p, pal.size());
            r.skip((long)pal.size());
        }
    }
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> px(n * nc);
    uint8_t cur[4] = {0, 0, 0, 0};
    int run = 0;
    bool repeating = false;
    for (size_t i = 0; i < n; i++) {
        bool fetch = true;
        if (rle) {
            if (run == 0) { const int c = r.u8(); run = 1 + (c & 127); repeating = (c >> 7) != 0; }
            else if (repeating) fetch = false;
        }
        if (fetch) {
            if (indexed) {
                int idx = bits == 8 ? r.u8() : r.u16le();
                if (idx >= pal_len) idx = 0;
                memcpy(cur, &pal[(size_t)idx * nc], nc);
            } else if (packed) tga_555(r.u16le(), cur);
            else for (int j = 0; j < nc; j++) cur[j] = (uint8_t)r.u8();
        }
        memcpy(&px[i * nc], cur, nc);
        run--;
    }
    nx = w; ny = h;
    rgb.resize(n * 3);
    for (int y = 0; y < h; y++) {
        const uint8_t * src = &px[(size_t)(bottom_up ? h - 1 - y : y) * w * nc];
        uint8_t * o = &rgb[(size_t)y * w * 3];
        for (int x = 0; x < w; x++, src += nc, o += 3) {
            if (nc <= 2) o[0] = o[1] = o[2] = src[0];                         // grey (+ alpha)
            else if (packed) { o[0] = src[0]; o[1] = src[1]; o[2] = src[2]; }
            else { o[0] = src[2]; o[1] = src[1]; o[2] = src[0]; }             // stored B, G, R (, A)
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// GIF (the first frame, on the logical screen)
// ------------------------------------------------------------------------------------------------------------------
struct GifCanvas {
    int W = 0, H = 0;
    std::vector<uint8_t> rgba;       // logical screen, starts all zero (transparent black)
    std::vector<uint8_t> touched;    // pixels the frame's raster visited (drawn or transparent)
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0, cx = 0, cy = 0, pass_left = 0, step = 1;     // frame rectangle, cursor, interlace state (rows)
    const uint8_t (*table)[4] = nullptr;                                             // active colour table, entries {B, G, R, A}
    void emit(int colour) {
        if (cy >= y1) return;
        const size_t at = (size_t)cy * W + cx;
        touched[at] = 1;
        const uint8_t * c = table[colour];
        if (c[3] > 128) { uint8_t * o = &rgba[at * 4]; o[0] = c[2]; o[1] = c[1]; o[2] = c[0]; o[3] = c[3]; }
        if (++cx >= x1) {
            cx = x0;
            cy += step;
            while (cy >= y1 && pass_left > 0) {           // interlace: rows 0, 8, ... then 4, 12, ... then 2, 6, ... then 1, 3, ...
                step = 1 << pass_left;
                cy = y0 + (step >> 1);
                pass_left--;
            }
        }
    }
};

bool decode_gif_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    if (d.size() < 6 || memcmp(d.data(), "GIF8", 4) != 0 || (d[4] != '7' && d[4] != '9') || d[5] != 'a') return false;
    Reader r(d);
    r.skip(6);
    GifCanvas g;
    g.W = r.u16le(); g.H = r.u16le();
    const int flags = r.u8();
    const int bg_index = r.u8();
    (void)r.u8();                                          // aspect ratio
    if (!pixels_fit(g.W, g.H)) return false;
    static thread_local uint8_t gpal[256][4], lpal[256][4];
    memset(gpal, 0, sizeof gpal); memset(lpal, 0, sizeof lpal);
    auto read_table = [&](uint8_t pal[256][4], int n, int transparent) {
        for (int i = 0; i < n; i++) {
            pal[i][2] = (uint8_t)r.u8(); pal[i][1] = (uint8_t)r.u8(); pal[i][0] = (uint8_t)r.u8();
            pal[i][3] = i == transparent ? 0 : 255;
        }
    };
    if (flags & 0x80) read_table(gpal, 2 << (flags & 7), -1);
    g.rgba.assign((size_t)g.W * g.H * 4, 0);
    g.touched.assign((size_t)g.W * g.H, 0);
    int gce_flags = 0, transparent = -1;
    for (;;) {
        const int tag = r.u8();
        if (tag == 0x21) {                                  // extension
            const int label = r.u8();
            int len;
            if (label == 0xF9) {                            // graphic control: transparency of the frame that follows
                len = r.u8();
                if (len == 4) {
                    gce_flags = r.u8();
                    (void)r.u16le();                        // delay
                    if (transparent >= 0) gpal[transparent][3] = 255;
                    if (gce_flags & 1) { transparent = r.u8(); gpal[transparent][3] = 0; }
                    else { r.skip(1); transparent = -1; }
                } else { r.skip(len); continue; }
            }
            while ((len = r.u8()) != 0) r.skip(len);
            continue;
        }
        if (tag != 0x2C) return false;                      // trailer before any image, or garbage
        const int fx = r.u16le(), fy = r.u16le(), fw = r.u16le(), fh = r.u16le();
        if (fx + fw > g.W || fy + fh > g.H) return false;
        g.x0 = fx; g.x1 = fx + fw; g.y0 = fy; g.y1 = fy + fh; g.cx = fx; g.cy = fw == 0 ? g.y1 : fy;
        const int lflags = r.u8();
        if (lflags & 0x40) { g.step = 8; g.pass_left = 3; } else { g.step = 1; g.pass_left = 0; }
        if (lflags & 0x80) { read_table(lpal, 2 << (lflags & 7), (gce_flags & 1) ? transparent : -1); g.table = lpal; }
        else if (flags & 0x80) g.table = gpal;
        else return false;
        // LZW raster
        const int min_bits = r.u8();
        if (min_bits > 12) return false;
        const int clear = 1 << min_bits;
        struct Entry { int16_t prefix; uint8_t first, suffix; };
        std::vector<Entry> tab(8192);
        for (int i = 0; i < clear; i++) tab[i] = {(int16_t)-1, (uint8_t)i, (uint8_t)i};
        int width = min_bits + 1, mask = (1 << width) - 1, avail = clear + 2, old = -1;
        bool cleared = false;
        uint32_t acc = 0;
        int have = 0, block = 0;
        std::vector<uint8_t> stack;
        stack.reserve(8192);
        bool done = false;
        while (!done) {
            if (have < width) {
                if (block == 0) { block = r.u8(); if (block == 0) break; }       // block terminator (or the end of the file)
                block--;
                acc |= (uint32_t)r.u8() << have;
                have += 8;
                continue;
            }
            const int code = (int)(acc & (uint32_t)mask);
            acc >>= width; have -= width;
            if (code == clear) { width = min_bits + 1; mask = (1 << width) - 1; avail = clear + 2; old = -1; cleared = true; }
            else if (code == clear + 1) done = true;                            // end of information
            else if (code <= avail) {
                if (!cleared) return false;                                      // a stream has to open with a clear code
                if (old >= 0) {
                    if (avail + 1 > 8192) return false;
                    Entry & e = tab[avail++];
                    e.prefix = (int16_t)old;
                    e.first = tab[old].first;
                    e.suffix = code == avail ? e.first : tab[code].first;
                } else if (code == avail) return false;
                stack.clear();
                for (int c = code; c >= 0; c = tab[c].prefix) stack.push_back(tab[c].suffix);
                for (size_t i = stack.size(); i-- > 0;) g.emit(stack[i]);
                if ((avail & mask) == 0 && avail <= 0x0FFF) { width++; mask = (1 << width) - 1; }
                old = code;
            } else return false;
        }
        break;
    }
    // what the raster did not visit takes the background colour — when its index is not 0, and with the table entry's bytes in
    // their stored order (B, G, R: the reference copies the entry as it lies); index 0 leaves those pixels black
    if (bg_index > 0) {
        for (size_t i = 0; i < g.touched.size(); i++)
            if (!g.touched[i]) { uint8_t * o = &g.rgba[i * 4]; o[0] = gpal[bg_index][0]; o[1] = gpal[bg_index][1]; o[2] = gpal[bg_index][2]; }
    }
    nx = g.W; ny = g.H;
    rgb.resize((size_t)g.W * g.H * 3);
    for (size_t i = 0; i < (size_t)g.W * g.H; i++) { rgb[i * 3] = g.rgba[i * 4]; rgb[i * 3 + 1] = g.rgba[i * 4 + 1]; rgb[i * 3 + 2] = g.rgba[i * 4 + 2]; }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// PSD (the flattened composite; RGB mode only, as the reference)
// ------------------------------------------------------------------------------------------------------------------
bool decode_psd_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    Reader r(d);
    if (r.u32be() != 0x38425053u) return false;            // "8BPS"
    if (r.u16be() != 1) return false;
    r.skip(6);
    const int channels = r.u16be();
    if (channels > 16) return false;
    const uint32_t hh = r.u32be(), ww = r.u32be();
    if (hh > (uint32_t)kMaxDim || ww > (uint32_t)kMaxDim) return false;
    const int depth = r.u16be();
    if (depth != 8 && depth != 16) return false;
    if (r.u16be() != 3) return false;                       // colour mode: RGB
    r.skip((long)r.u32be());                                // mode data
    r.skip((long)r.u32be());                                // image resources
    r.skip((long)r.u32be());                                // layer and mask information
    const int compression = r.u16be();
    if (compression > 1) return false;
    const int w = (int)ww, h = (int)hh;
    if (!pixels_fit(w, h)) return false;
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> px(n * 4);
    if (compression) {
        r.skip((long)h * channels * 2);                     // per-row byte counts
        for (int c = 0; c < 4; c++) {
            uint8_t * o = px.data() + c;
            if (c >= channels) { for (size_t i = 0; i < n; i++) o[i * 4] = c == 3 ? 255 : 0; continue; }
            size_t got = 0;
            while (got < n) {                               // PackBits over the whole plane
                int len = r.u8();
                if (len == 128) continue;
                if (len < 128) {
                    len++;
                    if ((size_t)len > n - got) return false;
                    for (int k = 0; k < len; k++) o[(got + k) * 4] = (uint8_t)r.u8();
                } else {
                    len = 257 - len;
                    if ((size_t)len > n - got) return false;
                    const uint8_t v = (uint8_t)r.u8();
                    for (int k = 0; k < len; k++) o[(got + k) * 4] = v;
                }
                got += len;
            }
        }
    } else {
        for (int c = 0; c < 4; c++) {
            uint8_t * o = px.data() + c;
            if (c >= channels) { for (size_t i = 0; i < n; i++) o[i * 4] = c == 3 ? 255 : 0; continue; }
            if (depth == 16) for (size_t i = 0; i < n; i++) o[i * 4] = (uint8_t)(r.u16be() >> 8);
            else for (size_t i = 0; i < n; i++) o[i * 4] = (uint8_t)r.u8();
        }
    }
    if (channels >= 4) {
        // the composite is stored matted on white: taken back out for partly transparent pixels, in float and truncated to a byte like
        // the reference does (value * (1 / a) + 255 * (1 - 1 / a))
        for (size_t i = 0; i < n; i++) {
            uint8_t * p = &px[i * 4];
            if (p[3] != 0 && p[3] != 255) {
                const float a = p[3] / 255.0f;
                const float ra = 1.0f / a;
                const float inv_a = 255.0f * (1 - ra);
                for (int c = 0; c < 3; c++) {
                    volatile float prod = p[c] * ra;         // product and sum rounded separately (no fused multiply-add)
                    const float v = prod + inv_a;
                    p[c] = (uint8_t)(int)v;
                }
            }
        }
    }
    nx = w; ny = h;
    rgb.resize(n * 3);
    for (size_t i = 0; i < n; i++) { rgb[i * 3] = px[i * 4]; rgb[i * 3 + 1] = px[i * 4 + 1]; rgb[i * 3 + 2] = px[i * 4 + 2]; }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// binary PNM (P5 / P6), any maxval up to 65535
// ------------------------------------------------------------------------------------------------------------------
bool decode_pnm_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    if (d.size() < 3 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return false;
    const int ch = d[1] == '6' ? 3 : 1;
    Reader r(d);
    r.skip(2);
    int c = r.u8();
    auto is_space = [](int ch_) { return ch_ == ' ' || ch_ == '\t' || ch_ == '\n' || ch_ == '\v' || ch_ == '\f' || ch_ == '\r'; };
    auto skip_blank = [&]() {
        for (;;) {
            while (!r.at_end() && is_space(c)) c = r.u8();
            if (r.at_end() || c != '#') break;
            while (!r.at_end() && c != '\n' && c != '\r') c = r.u8();
        }
    };
    bool overflow = false;
    auto number = [&]() {
        long v = 0;
        while (!r.at_end() && c >= '0' && c <= '9') {
            v = v * 10 + (c - '0');
            c = r.u8();
            if (v > 214748364 || (v == 214748364 && c > '7')) { overflow = true; return 0L; }
        }
        return v;
    };
    skip_blank();
    const long w = number();
    if (overflow || w == 0) return false;
    skip_blank();
    const long h = number();
    if (overflow || h == 0) return false;
    skip_blank();
    const long maxv = number();
    if (overflow || maxv > 65535) return false;
    // (the single blank behind maxval is the byte `c` already holds; the samples start at the reader's position)
    if (!pixels_fit(w, h)) return false;
    const int bytes = maxv > 255 ? 2 : 1;
    const size_t n = (size_t)w * h, need = n * ch * bytes;
    if ((size_t)(r.end - r.p) < need) return false;
    nx = (int)w; ny = (int)h;
    rgb.resize(n * 3);
    const uint8_t * s = r.p;
    // samples are taken as they are (no scaling by maxval); of a two-byte sample the SECOND byte is kept — the reference reads the
    // big-endian pair as a little-endian word and keeps its high byte
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) rgb[i * 3 + k] = s[(i * ch + (ch == 3 ? k : 0)) * bytes + (bytes - 1)];
    return true;
}


// ------------------------------------------------------------------------------------------------------------------
// Softimage PIC
// ------------------------------------------------------------------------------------------------------------------
bool decode_pic_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    static const uint8_t magic[4] = {0x53, 0x80, 0xF6, 0x34};
    if (d.size() < 92 || memcmp(d.data(), magic, 4) != 0 || memcmp(d.data() + 88, "PICT", 4) != 0) return false;
    Reader r(d);
    r.skip(92);
    const int w = r.u16be(), h = r.u16be();
    if (r.at_end() || !pixels_fit(w, h)) return false;
    r.skip(8);                                              // ratio, fields, pad
    struct Packet { int type, channels; } pk[10];
    int n_pk = 0, chained;
    do {                                                    // channel packets: which of R G B A (bits 7..4) a run of the scanline data carries, and how it is packed
        if (n_pk == 10) return false;
        chained = r.u8();
        const int bits = r.u8();
        pk[n_pk].type = r.u8();
        pk[n_pk].channels = r.u8();
        n_pk++;
        if (r.at_end() || bits != 8) return false;
    } while (chained);
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> px(n * 4, 0xff);
    auto read_px = [&](int channels, uint8_t * dst) {       // false: the file ends inside a pixel
        for (int i = 0, m = 0x80; i < 4; i++, m >>= 1)
            if (channels & m) { if (r.at_end()) return false; dst[i] = (uint8_t)r.u8(); }
        return true;
    };
    auto copy_px = [](int channels, uint8_t * dst, const uint8_t * src) {
        for (int i = 0, m = 0x80; i < 4; i++, m >>= 1) if (channels & m) dst[i] = src[i];
    };
    for (int y = 0; y < h; y++)
        for (int k = 0; k < n_pk; k++) {
            uint8_t * dst = &px[(size_t)y * w * 4];
            const int ch = pk[k].channels;
            if (pk[k].type == 0) {                          // raw
                for (int x = 0; x < w; x++, dst += 4) if (!read_px(ch, dst)) return false;
            } else if (pk[k].type == 1) {                   // runs only: count, value
                int left = w;
                while (left > 0) {
                    int count = r.u8();
                    if (r.at_end()) return false;
                    if (count > left) count = left;
                    uint8_t v[4];
                    if (!read_px(ch, v)) return false;
                    for (int i = 0; i < count; i++, dst += 4) copy_px(ch, dst, v);
                    left -= count;
                }
            } else if (pk[k].type == 2) {                   // mixed: >= 128 a run (128: 16-bit length follows), < 128 count + 1 raw pixels
                int left = w;
                while (left > 0) {
                    int count = r.u8();
                    if (r.at_end()) return false;
                    if (count >= 128) {
                        count = count == 128 ? r.u16be() : count - 127;
                        if (count > left) return false;
                        uint8_t v[4];
                        if (!read_px(ch, v)) return false;
                        for (int i = 0; i < count; i++, dst += 4) copy_px(ch, dst, v);
                    } else {
                        count++;
                        if (count > left) return false;
                        for (int i = 0; i < count; i++, dst += 4) if (!read_px(ch, dst)) return false;
                    }
                    left -= count;
                }
            } else return false;
        }
    nx = w; ny = h;
    rgb.resize(n * 3);
    for (size_t i = 0; i < n; i++) { rgb[i * 3] = px[i * 4]; rgb[i * 3 + 1] = px[i * 4 + 1]; rgb[i * 3 + 2] = px[i * 4 + 2]; }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Radiance RGBE (.hdr): decoded to float, then to 8 bits the way the reference's loader does for an 8-bit request
// (value ^ (1 / 2.2) * 255 + 0.5, clamped, truncated)
// ------------------------------------------------------------------------------------------------------------------
bool decode_hdr_impl(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    if (!((d.size() >= 11 && !memcmp(d.data(), "#?RADIANCE\n", 11)) || (d.size() >= 7 && !memcmp(d.data(), "#?RGBE\n", 7)))) return false;
    Reader r(d);
    auto line = [&]() {                                     // up to the next newline; over-long lines are cut at 1022 characters
        std::string t;
        int c = r.u8();
        while (!r.at_end() && c != '\n') {
            t.push_back((char)c);
            if (t.size() == 1023) { while (!r.at_end() && r.u8() != '\n') {} break; }
            c = r.u8();
        }
        const size_t z = t.find('\0');                      // (the reference compares C strings)
        if (z != std::string::npos) t.resize(z);
        return t;
    };
    const std::string id = line();
    if (id != "#?RADIANCE" && id != "#?RGBE") return false;
    bool rgbe = false;
    for (;;) {
        const std::string t = line();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") rgbe = true;
    }
    if (!rgbe) return false;
    const std::string res = line();
    if (res.compare(0, 3, "-Y ") != 0) return false;
    const char * q = res.c_str() + 3;
    char * e = nullptr;
    const long h = strtol(q, &e, 10);
    while (*e == ' ') e++;
    if (strncmp(e, "+X ", 3) != 0) return false;
    const long w = strtol(e + 3, nullptr, 10);
    if (!pixels_fit(w, h)) return false;
    const size_t n = (size_t)w * h;
    std::vector<float> f(n * 3);
    auto to_float = [](float * o, const uint8_t * in) {
        if (in[3] != 0) {
            const float s = (float)ldexp(1.0f, (int)in[3] - (128 + 8));
            o[0] = in[0] * s; o[1] = in[1] * s; o[2] = in[2] * s;
        } else o[0] = o[1] = o[2] = 0.f;
    };
    uint8_t held[4] = {0, 0, 0, 0};
    auto flat_from = [&](size_t first) {                    // four bytes per pixel, no packing
        for (size_t i = first; i < n; i++) {
            // a file that ends early: the reference's 4-byte read fails without consuming anything and the pixel it decoded last is
            // converted again — the rest of the image repeats it
            if ((size_t)(r.end - r.p) >= 4) { memcpy(held, r.p, 4); r.p += 4; }
            to_float(&f[i * 3], held);
        }
    };
    if (w < 8 || w >= 32768) flat_from(0);
    else {
        std::vector<uint8_t> scan((size_t)w * 4);
        for (long y = 0; y < h; y++) {
            const int c1 = r.u8(), c2 = r.u8();
            int len = r.u8();
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                // not a packed scanline: the reference takes these four bytes for pixel 0 and reads the REST OF THE IMAGE flat from
                // here, whichever scanline it was in
                held[0] = (uint8_t)c1; held[1] = (uint8_t)c2; held[2] = (uint8_t)len; held[3] = (uint8_t)r.u8();
                to_float(&f[0], held);
                flat_from(1);
                break;
            }
            len = (len << 8) | r.u8();
            if (len != w) return false;
            for (int k = 0; k < 4; k++) {
                long i = 0;
                while (i < w) {
                    int count = r.u8();
                    if (count > 128) {
                        const int v = r.u8();
                        count -= 128;
                        if (count > w - i) return false;
                        for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = (uint8_t)v;
                    } else {
                        if (count == 0 || count > w - i) return false;
                        for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = (uint8_t)r.u8();
                    }
                }
            }
            for (long i = 0; i < w; i++) to_float(&f[((size_t)y * w + i) * 3], &scan[(size_t)i * 4]);
        }
    }
    nx = (int)w; ny = (int)h;
    rgb.resize(n * 3);
    const float scale = 1.0f, inv_gamma = 1.0f / 2.2f;
    for (size_t i = 0; i < n * 3; i++) {
        const float lin = f[i] * scale;
        float z = (float)pow((double)lin, (double)inv_gamma) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        rgb[i] = (uint8_t)(int)z;
    }
    return true;
}

}  // namespace

bool decode_bmp(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_bmp_impl(d, rgb, nx, ny); }
bool decode_tga(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_tga_impl(d, rgb, nx, ny); }
bool decode_gif(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_gif_impl(d, rgb, nx, ny); }
bool decode_psd(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_psd_impl(d, rgb, nx, ny); }
bool decode_pnm(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_pnm_impl(d, rgb, nx, ny); }
bool decode_pic(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_pic_impl(d, rgb, nx, ny); }
bool decode_hdr(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) { return decode_hdr_impl(d, rgb, nx, ny); }

}  // namespace clipamd
