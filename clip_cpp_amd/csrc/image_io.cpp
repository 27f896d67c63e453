// image_io.cpp — clip_image_load_from_file (reference clip.cpp:709-726, which delegates to the vendored
// stb_image).  Host-side and outside the GPU hot path.  Own decoders, 3-channel RGB
// output like stbi_load(..., 3): PNG here (8/16-bit, all colour types, non-interlaced and Adam7, inflate
// via zlib), JPEG in jpeg_decode.cpp (baseline + progressive Huffman, grey / YCbCr / RGB / CMYK / YCCK),
// BMP / GIF / PSD / PIC / PNM / HDR / TGA in image_formats.cpp.  Formats are tried in the reference decoder's order.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

#include "model.h"

namespace clipamd {

bool decode_jpeg(const uint8_t * data, size_t size, std::vector<uint8_t> & rgb, int & nx, int & ny, std::string & err);  // jpeg_decode.cpp
// image_formats.cpp
bool decode_bmp(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);
bool decode_gif(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);
bool decode_psd(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);
bool decode_pnm(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);
bool decode_tga(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);
bool decode_pic(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);
bool decode_hdr(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny);

namespace {

bool read_file(const char * fname, std::vector<uint8_t> & out) {
    FILE * f = fopen(fname, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) { fclose(f); return false; }
    out.resize((size_t)n);
    const bool ok = fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// ---- PNG ----
inline uint32_t be32(const uint8_t * p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool unfilter(uint8_t * data, size_t & pos, size_t avail, int w, int h, int bpp_bits, std::vector<uint8_t> & out) {
    // data at pos: h scanlines of (1 + rowbytes); writes raw pixels rows into out (rowbytes each)
    const size_t rowbytes = ((size_t)w * bpp_bits + 7) / 8;
    const int bpp = std::max(1, bpp_bits / 8);
    out.assign(rowbytes * h, 0);
    std::vector<uint8_t> zero(rowbytes, 0);
    for (int y = 0; y < h; y++) {
        if (pos + 1 + rowbytes > avail) return false;
        const int ft = data[pos];
        const uint8_t * src = data + pos + 1;
        uint8_t * cur = &out[rowbytes * y];
        const uint8_t * up = y ? &out[rowbytes * (y - 1)] : zero.data();
        for (size_t i = 0; i < rowbytes; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up[i], c = i >= (size_t)bpp ? up[i - bpp] : 0;
            int v = src[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            cur[i] = (uint8_t)v;
        }
        pos += 1 + rowbytes;
    }
    return true;
}

bool decode_png(const std::vector<uint8_t> & d, std::vector<uint8_t> & rgb, int & nx, int & ny) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) return false;
    size_t p = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (p + 12 <= d.size()) {
        const uint32_t len = be32(&d[p]);
        const uint8_t * type = &d[p + 4];
        if (!memcmp(type, "IEND", 4)) break;                                          // (whatever length it claims)
        if (p + 12 + len > d.size()) return false;
        const uint8_t * body = &d[p + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) return false;      // compression / filter method 0 only, interlace 0 or 1
        } else if (!memcmp(type, "PLTE", 4)) {
            if (len > 256 * 3 || len % 3) return false;
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if ((type[0] & 0x20) == 0 && memcmp(type, "tRNS", 4) != 0) {
            return false;     // an unknown CRITICAL chunk (upper-case first letter).  Apple's CgBI flavour lands here too: the reference's inflate refuses a
                              // headerless stream that ends without spare bytes, i.e. practically every such file
        }
        p += 12 + len;
    }
    if (w <= 0 || h <= 0 || w > (1 << 15) || h > (1 << 15)) return false;
    int chans;
    switch (ctype) {
    case 0: chans = 1; break;
    case 2: chans = 3; break;
    case 3: chans = 1; break;
    case 4: chans = 2; break;
    case 6: chans = 4; break;
    default: return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && plte.empty()) return false;                                    // indexed colour needs its table
    const int bpp_bits = chans * depth;
    // inflate
    size_t raw_cap = 0;
    if (!interlace) raw_cap = (((size_t)w * bpp_bits + 7) / 8 + 1) * h;
    else raw_cap = (((size_t)w * bpp_bits + 7) / 8 + 8) * (h + 8) * 2;
    std::vector<uint8_t> raw(raw_cap);
    uLongf dl = 0;
    {
        // streamed inflate instead of uncompress(): what matters is that the scanlines come out — data behind them (a file whose IDAT
        // holds more than the image needs) and a wrong Adler-32 trailer are not errors for the reference's decoder either
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) return false;
        zs.next_in = idat.data(); zs.avail_in = (uInt)std::min<size_t>(idat.size(), 0xffffffffu);
        zs.next_out = raw.data(); zs.avail_out = (uInt)std::min<size_t>(raw.size(), 0xffffffffu);
        const int zr = inflate(&zs, Z_FINISH);
        dl = zs.total_out;
        const bool trailer_only = zr == Z_DATA_ERROR && zs.msg && strstr(zs.msg, "incorrect data check");
        inflateEnd(&zs);
        if (zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR && !trailer_only) return false;
    }

    nx = w; ny = h;
    rgb.assign((size_t)w * h * 3, 0);
    auto put_pixel = [&](const std::vector<uint8_t> & rows, int pw, int px, int py, int ox, int oy) {
        // fetch pixel (px,py) of a (sub)image with width pw from unfiltered rows -> rgb[oy][ox]
        const size_t rowbytes = ((size_t)pw * bpp_bits + 7) / 8;
        const uint8_t * row = &rows[rowbytes * py];
        int v[4] = {0, 0, 0, 255};
        if (depth < 8) {
            const int bit = px * depth;
            const int s = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
            v[0] = ctype == 3 ? s : s * 255 / ((1 << depth) - 1);
        } else {
            const int bs = depth / 8;
            for (int c = 0; c < chans; c++) v[c] = row[((size_t)px * chans + c) * bs];  // 16-bit: high byte
        }
        uint8_t * o = &rgb[((size_t)oy * w + ox) * 3];
        if (ctype == 3) {
            const size_t idx = (size_t)v[0] * 3;
            if (idx + 2 < plte.size()) { o[0] = plte[idx]; o[1] = plte[idx + 1]; o[2] = plte[idx + 2]; }
        } else if (ctype == 0 || ctype == 4) {
            o[0] = o[1] = o[2] = (uint8_t)v[0];
        } else {
            o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2];
        }
    };
    size_t pos = 0;
    std::vector<uint8_t> rows;
    if (!interlace) {
        if (!unfilter(raw.data(), pos, dl, w, h, bpp_bits, rows)) return false;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) put_pixel(rows, w, x, y, x, y);
    } else {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int pass = 0; pass < 7; pass++) {
            const int pw = (w - xs[pass] + dx[pass] - 1) / dx[pass], ph = (h - ys[pass] + dy[pass] - 1) / dy[pass];
            if (pw <= 0 || ph <= 0) continue;
            if (!unfilter(raw.data(), pos, dl, pw, ph, bpp_bits, rows)) return false;
            for (int y = 0; y < ph; y++)
                for (int x = 0; x < pw; x++) put_pixel(rows, pw, x, y, xs[pass] + x * dx[pass], ys[pass] + y * dy[pass]);
        }
    }
    return true;
}

}  // namespace

bool load_image_file(const char * fname, clip_image_u8 * img) {
    std::vector<uint8_t> d, rgb;
    int nx = 0, ny = 0;
    std::string err;
    bool ok = read_file(fname, d);
    if (ok) {
        // (the order of the reference's decoder: formats with a magic number first, TGA — which has none — last)
        ok = decode_png(d, rgb, nx, ny) || decode_bmp(d, rgb, nx, ny) || decode_gif(d, rgb, nx, ny) || decode_psd(d, rgb, nx, ny) ||
             decode_pic(d, rgb, nx, ny) || decode_jpeg(d.data(), d.size(), rgb, nx, ny, err) || decode_pnm(d, rgb, nx, ny) || decode_hdr(d, rgb, nx, ny) ||
             decode_tga(d, rgb, nx, ny);
    }
    if (!ok) {
        fprintf(stderr, "%s: failed to load '%s'%s%s\n", "clip_image_load_from_file", fname, err.empty() ? "" : ": ", err.c_str());
        return false;
    }
    img->nx = nx;
    img->ny = ny;
    img->size = (size_t)nx * ny * 3;
    img->data = new uint8_t[img->size]();
    memcpy(img->data, rgb.data(), img->size);
    return true;
}

}  // namespace clipamd
