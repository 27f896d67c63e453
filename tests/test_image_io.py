"""clip_image_load_from_file: own decoders for what the reference's decoder reads (PNG, BMP, GIF, PSD, JPEG, PNM, TGA).  The pixels
must equal what the REFERENCE's decoder (its vendored stb_image, built from /root/reference into oracle/_ref/ by `make -C oracle ref`)
produces, because they feed the bit-exact preprocessing; lossless formats are also checked against PIL / numpy."""
import ctypes as C
import io
import os
import subprocess

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")


def load_ours(clip_lib, path):
    L = clip_lib.lib()
    img = L.clip_image_u8_make()
    try:
        if not L.clip_image_load_from_file(os.fsencode(path), img):
            return None
        c = img.contents
        assert c.size == c.nx * c.ny * 3
        return np.ctypeslib.as_array(c.data, shape=(c.ny, c.nx, 3)).copy()
    finally:
        L.clip_image_u8_free(img)


@pytest.fixture(scope="module")
def stb():
    if not os.path.exists(REF_SO):
        if not os.path.isdir("/root/reference"):
            pytest.skip("reference tree (stb_image.h) not available on this machine")
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(REF_SO)
    L.stbref_load_from_memory.restype = C.POINTER(C.c_ubyte)
    L.stbref_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.stbref_free.argtypes = [C.c_void_p]

    def load(data):
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        p = L.stbref_load_from_memory(data, len(data), C.byref(x), C.byref(y), C.byref(c))
        if not p:
            return None
        a = np.ctypeslib.as_array(p, shape=(y.value, x.value, 3)).copy()
        L.stbref_free(p)
        return a
    return load


def _photo(h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(np.sin(xx / 17.0 + yy / 31.0) * 0.5 + 0.5) * 255, (np.cos(yy / 13.0) * 0.5 + 0.5) * 255, (xx * 3 + yy * 5) % 256], -1)
    return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)


JPEG_CASES = [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True),
              dict(subsampling=0, progressive=True), dict(subsampling=1, progressive=True, quality=35), dict(quality=95, optimize=True),
              dict(gray=True), dict(gray=True, progressive=True),
              # restart intervals (DRI + RSTn markers: cameras and many encoders write them; PIL only on request)
              dict(subsampling=2, restart_marker_blocks=1), dict(subsampling=0, restart_marker_blocks=3),
              dict(subsampling=1, progressive=True, restart_marker_blocks=2), dict(gray=True, restart_marker_blocks=7),
              dict(subsampling=2, restart_marker_rows=1)]


@pytest.mark.parametrize("size", [(64, 64), (57, 83), (1, 1), (200, 3), (17, 250)])
@pytest.mark.parametrize("case", range(len(JPEG_CASES)))
def test_jpeg_pixels_equal_reference_decoder(clip_lib, stb, tmp_path, size, case):
    kw = dict(JPEG_CASES[case])
    gray = kw.pop("gray", False)
    img = _photo(size[0], size[1], seed=case)
    pim = PIL.fromarray(img).convert("L") if gray else PIL.fromarray(img)
    buf = io.BytesIO()
    pim.save(buf, "JPEG", **{"quality": 80, **kw})
    data = buf.getvalue()
    assert (b"\xff\xdd" in data) == any(k.startswith("restart") for k in kw)
    path = str(tmp_path / "t.jpg")
    open(path, "wb").write(data)
    want = stb(data)
    got = load_ours(clip_lib, path)
    assert want is not None and got is not None
    assert got.shape == want.shape
    assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="reference tree only exists in the dev container")
@pytest.mark.parametrize("name", ["red_apple.jpg", "white.jpg"])
def test_reference_test_images(clip_lib, stb, name):
    path = os.path.join("/root/reference/tests", name)
    want = stb(open(path, "rb").read())
    got = load_ours(clip_lib, path)
    assert got is not None and np.array_equal(got, want)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1", "I;16"])
@pytest.mark.parametrize("interlace", [False, True])
def test_png_lossless(clip_lib, tmp_path, mode, interlace):
    img = _photo(37, 53, seed=5)
    pim = PIL.fromarray(img)
    if mode == "I;16":
        pim = PIL.fromarray((img[:, :, 0].astype(np.uint16) << 8) | img[:, :, 1])
    elif mode != "RGB":
        pim = pim.convert(mode)
    path = str(tmp_path / "t.png")
    if interlace:
        # PIL cannot write Adam7; interlaced decoding is covered through the optimize flag only when pngcrush-like tools exist
        pim.save(path, "PNG", optimize=True)
    else:
        pim.save(path, "PNG")
    got = load_ours(clip_lib, path)
    assert got is not None
    if mode == "I;16":
        want = np.repeat((np.asarray(pim) >> 8).astype(np.uint8)[:, :, None], 3, axis=2)
    elif mode in ("LA",):
        want = np.repeat(np.asarray(pim)[:, :, :1], 3, axis=2)
    elif mode == "RGBA":
        want = np.asarray(pim)[:, :, :3]
    else:
        want = np.asarray(pim.convert("RGB"))
    assert np.array_equal(got, want)


def test_bmp_and_pnm(clip_lib, tmp_path):
    img = _photo(21, 34, seed=9)
    p = str(tmp_path / "t.bmp")
    PIL.fromarray(img).save(p, "BMP")
    assert np.array_equal(load_ours(clip_lib, p), img)
    p = str(tmp_path / "t.ppm")
    PIL.fromarray(img).save(p, "PPM")
    assert np.array_equal(load_ours(clip_lib, p), img)
    p = str(tmp_path / "t.pgm")
    PIL.fromarray(img[:, :, 0]).save(p, "PPM")
    assert np.array_equal(load_ours(clip_lib, p), np.repeat(img[:, :, :1], 3, axis=2))


def test_unreadable_files_fail_cleanly(clip_lib, tmp_path):
    assert load_ours(clip_lib, str(tmp_path / "missing.jpg")) is None
    p = str(tmp_path / "junk.jpg")
    open(p, "wb").write(b"\xff\xd8\xff\xe0" + os.urandom(200))
    assert load_ours(clip_lib, p) is None
    p = str(tmp_path / "trunc.png")
    buf = io.BytesIO()
    PIL.fromarray(_photo(20, 20)).save(buf, "PNG")
    open(p, "wb").write(buf.getvalue()[:60])
    assert load_ours(clip_lib, p) is None


# ---------------------------------------------------------------------------------------------------------------------
# The other formats the reference's decoder reads (image_formats.cpp): every case is compared with that decoder itself.
# ---------------------------------------------------------------------------------------------------------------------
import struct


def _same_as_reference(clip_lib, stb, tmp_path, data, name="t.bin", must_load=True):
    path = str(tmp_path / name)
    open(path, "wb").write(data)
    want, got = stb(data), load_ours(clip_lib, path)
    if want is None:
        assert not must_load, "the reference decoder rejects this fixture"
        assert got is None, "the reference rejects this file"
        return None
    assert got is not None, "the reference reads this file"
    assert got.shape == want.shape
    assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()
    return got


def _pil_bytes(pim, fmt, **kw):
    buf = io.BytesIO()
    pim.save(buf, fmt, **kw)
    return buf.getvalue()


def _bmp(w, h, bpp, rows, palette=None, hsz=40, comp=0, masks=None, top_down=False, gap=0):
    """A BMP file from raw row bytes (un-padded, top row first)."""
    stride = (len(rows[0]) + 3) & ~3
    body = b"".join(r + b"\0" * (stride - len(r)) for r in (rows if top_down else rows[::-1]))
    if hsz == 12:
        hdr = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        hdr = struct.pack("<IiiHHIIiiII", hsz, w, -h if top_down else h, 1, bpp, comp, len(body), 2835, 2835, 0, 0)
        if hsz == 56:
            hdr += b"\0" * 16
        if hsz in (108, 124):
            m = masks or (0, 0, 0, 0)
            hdr += struct.pack("<IIII", *(tuple(m) + (0,) * (4 - len(m)))) + b"\0" * (4 + 48) + (b"\0" * 16 if hsz == 124 else b"")
    extra = b""
    if hsz in (40, 56) and comp == 3:
        extra = struct.pack("<III", *masks[:3])
    pal = b""
    if palette is not None:
        pal = b"".join(bytes((b, g, r)) + (b"" if hsz == 12 else b"\0") for r, g, b in palette)
    off = 14 + len(hdr) + len(extra) + len(pal) + gap
    return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + hdr + extra + pal + b"\0" * gap + body


def _pack_bits(idx, bpp):
    per = 8 // bpp
    out = bytearray()
    for i in range(0, len(idx), per):
        v = 0
        for k in range(per):
            v = (v << bpp) | (int(idx[i + k]) if i + k < len(idx) else 0)
        out.append(v)
    return bytes(out)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (8, 3), (33, 9)])
def test_bmp_flavours_equal_reference_decoder(clip_lib, stb, tmp_path, w, h):
    rng = np.random.default_rng(w * 100 + h)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.bmp", **kw)
    # what PIL writes: 1-bit, 8-bit grey palette, 8-bit colour palette, 24-bit, 32-bit
    pim = PIL.fromarray(img)
    for mode in ("1", "L", "P", "RGB", "RGBA"):
        same(_pil_bytes(pim.convert(mode), "BMP"))
    # paletted 1 / 4 / 8 bpp with an arbitrary colour table, bottom-up and top-down, a short table, OS/2 and V4 / V5 headers
    for bpp in (1, 4, 8):
        n = 1 << bpp
        pal = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(n)]
        idx = rng.integers(0, n, (h, w))
        rows = [_pack_bits(idx[y], bpp) for y in range(h)]
        got = same(_bmp(w, h, bpp, rows, palette=pal))
        assert np.array_equal(got, np.array(pal, dtype=np.uint8)[idx])
        same(_bmp(w, h, bpp, rows, palette=pal, top_down=True))
        same(_bmp(w, h, bpp, rows, palette=pal, hsz=108))
        same(_bmp(w, h, bpp, rows, palette=pal, hsz=124))
        # OS/2 header: the reference reads 4 colour-table entries too few (pixels that use the last four come out of uninitialised
        # memory there: only the others are compared; with a 2-colour table nothing is defined but the size)
        if bpp > 1:
            lo = np.minimum(idx, n - 5)
            same(_bmp(w, h, bpp, [_pack_bits(lo[y], bpp) for y in range(h)], palette=pal, hsz=12))
        else:
            path = str(tmp_path / "os2.bmp")
            open(path, "wb").write(_bmp(w, h, bpp, rows, palette=pal, hsz=12))
            assert load_ours(clip_lib, path).shape == (h, w, 3)
        same(_bmp(w, h, bpp, [_pack_bits(np.minimum(idx[y], 1), bpp) for y in range(h)], palette=pal[:2]))
    # 16 bpp: default x1r5g5b5, and BI_BITFIELDS 5:6:5 / 4:4:4:4 / odd masks; 32 bpp: default, bitfields in another order, 10:10:10
    v16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    rows16 = [v16[y].astype("<u2").tobytes() for y in range(h)]
    got = same(_bmp(w, h, 16, rows16))
    want = np.stack([(v16 >> 10) & 31, (v16 >> 5) & 31, v16 & 31], -1).astype(np.uint32)
    assert np.array_equal(got, ((want << 3) | (want >> 2)).astype(np.uint8))
    for masks in ((0xF800, 0x07E0, 0x001F), (0x0F00, 0x00F0, 0x000F), (0x7000, 0x0180, 0x0007), (0x00FF, 0xFF00, 0x0001)):
        same(_bmp(w, h, 16, rows16, comp=3, masks=masks))
        same(_bmp(w, h, 16, rows16, comp=3, masks=masks + (0x8000,), hsz=108))
    v32 = rng.integers(0, 1 << 32, (h, w), dtype=np.uint32)
    rows32 = [v32[y].astype("<u4").tobytes() for y in range(h)]
    same(_bmp(w, h, 32, rows32))
    same(_bmp(w, h, 32, rows32, top_down=True, gap=16))
    for masks in ((0x000000FF, 0x0000FF00, 0x00FF0000), (0x3FF00000, 0x000FFC00, 0x000003FF), (0xFF000000, 0x00FF0000, 0x0000FF00)):
        same(_bmp(w, h, 32, rows32, comp=3, masks=masks), must_load=False)       # (10-bit masks: refused by both)
        same(_bmp(w, h, 32, rows32, comp=3, masks=masks + (0x000000FF,), hsz=124), must_load=False)
    same(_bmp(w, h, 32, rows32, hsz=108))                                          # BI_RGB under a V4 header: the header's masks are ignored
    rows24 = [img[y, :, ::-1].tobytes() for y in range(h)]
    for hsz in (12, 40, 56, 108, 124):
        assert np.array_equal(same(_bmp(w, h, 24, rows24, hsz=hsz)), img)
    # refused by both: RLE, embedded PNG, equal masks, a pixel offset far behind the header
    same(_bmp(w, h, 8, [bytes(w)] * h, palette=[(0, 0, 0)] * 256, comp=1), must_load=False)
    same(_bmp(w, h, 24, rows24, comp=5), must_load=False)
    same(_bmp(w, h, 16, rows16, comp=3, masks=(0x1F, 0x1F, 0x1F)), must_load=False)
    same(_bmp(w, h, 24, rows24, gap=2000), must_load=False)


def _tga(w, h, bits, pixels, type_, cmap=None, cmap_bits=0, desc=0, ident=b"", rle=False, cmap_first=0):
    """pixels: list of per-pixel byte strings in file order."""
    if rle:
        body, i = bytearray(), 0
        while i < len(pixels):
            run = 1
            while i + run < len(pixels) and run < 128 and pixels[i + run] == pixels[i]:
                run += 1
            if run > 1:
                body += bytes((0x80 | (run - 1),)) + pixels[i]
                i += run
            else:
                lit = 1
                while i + lit < len(pixels) and lit < 128 and pixels[i + lit] != pixels[i + lit - 1]:
                    lit += 1
                body += bytes((lit - 1,)) + b"".join(pixels[i:i + lit])
                i += lit
        body = bytes(body)
    else:
        body = b"".join(pixels)
    n_cmap = len(cmap) if cmap else 0
    hdr = struct.pack("<BBBHHBHHHHBB", len(ident), 1 if cmap else 0, type_ + (8 if rle else 0), cmap_first, n_cmap, cmap_bits, 0, 0, w, h, bits, desc)
    return hdr + ident + b"\0" * cmap_first + (b"".join(cmap) if cmap else b"") + body


@pytest.mark.parametrize("w,h", [(1, 1), (6, 4), (131, 3)])
@pytest.mark.parametrize("rle", [False, True])
def test_tga_flavours_equal_reference_decoder(clip_lib, stb, tmp_path, w, h, rle):
    rng = np.random.default_rng(w + h)
    # long runs and literals mixed, so that RLE packets straddle rows
    base = rng.integers(0, 256, (h * w, 4), dtype=np.uint8)
    base[rng.random(h * w) < 0.6] = base[0]
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.tga", **kw)
    px = lambda n: [bytes(base[i, :n]) for i in range(h * w)]
    for desc in (0, 0x20, 0x10, 0x28):                                              # bottom-up, top-down, (right-to-left: ignored), alpha bits
        same(_tga(w, h, 24, px(3), 2, desc=desc, rle=rle))
        same(_tga(w, h, 32, px(4), 2, desc=desc, rle=rle))
    same(_tga(w, h, 8, px(1), 3, rle=rle))                                          # grey
    same(_tga(w, h, 16, px(2), 3, rle=rle))                                         # grey + alpha
    same(_tga(w, h, 16, px(2), 2, rle=rle))                                         # 5:5:5 (+ attribute bit)
    same(_tga(w, h, 15, px(2), 2, rle=rle))
    same(_tga(w, h, 24, px(3), 2, ident=b"made by a test", rle=rle))
    for cmap_bits, n in ((24, 3), (32, 4), (16, 2), (15, 2), (8, 1)):               # colour-mapped, 8- and 16-bit indices, indices past the table
        cmap = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for _ in range(200)]
        same(_tga(w, h, 8, px(1), 1, cmap=cmap, cmap_bits=cmap_bits, rle=rle))
        same(_tga(w, h, 16, [bytes((p[0], p[1] & 1)) for p in px(2)], 1, cmap=cmap, cmap_bits=cmap_bits, rle=rle))
    same(_tga(w, h, 8, px(1), 1, cmap=[bytes((i, 255 - i, i // 2)) for i in range(256)], cmap_bits=24, cmap_first=5, rle=rle))
    if not rle:
        pim = PIL.fromarray(base[:, :3].reshape(h, w, 3))
        for mode in ("L", "LA", "P", "RGB", "RGBA"):
            same(_pil_bytes(pim.convert(mode), "TGA"))
            same(_pil_bytes(pim.convert(mode), "TGA", compression="tga_rle"))
        # truncated pixel data: the reference returns rows it never filled; only the size is defined
        path = str(tmp_path / "trunc.tga")
        open(path, "wb").write(_tga(w, h, 24, px(3), 2)[:18 + (3 * w * h) // 2])
        assert load_ours(clip_lib, path).shape == (h, w, 3)


def _gif_lzw(indices, min_bits):
    """Variable-width LZW, the real algorithm (string table, KwKwK codes occur on runs)."""
    clear, end = 1 << min_bits, (1 << min_bits) + 1
    out, acc, nacc = bytearray(), 0, 0

    def put(code, width):
        nonlocal acc, nacc
        acc |= code << nacc
        nacc += width
        while nacc >= 8:
            out.append(acc & 255)
            acc >>= 8
            nacc -= 8
    table = {(i,): i for i in range(clear)}
    width, nxt = min_bits + 1, end + 1
    put(clear, width)
    cur = ()
    for v in indices:
        v = int(v)
        if cur + (v,) in table:
            cur = cur + (v,)
            continue
        put(table[cur], width)
        if nxt < 4096:
            table[cur + (v,)] = nxt
            nxt += 1
            if nxt - 1 == (1 << width) and width < 12:
                width += 1
        else:
            put(clear, width)
            table = {(i,): i for i in range(clear)}
            width, nxt = min_bits + 1, end + 1
        cur = (v,)
    if cur:
        put(table[cur], width)
    put(end, width)
    if nacc:
        out.append(acc & 255)
    blocks = b"".join(bytes((len(out[i:i + 255]),)) + bytes(out[i:i + 255]) for i in range(0, len(out), 255))
    return bytes((min_bits,)) + blocks + b"\0"


def _gif(W, H, frame, gpal=None, lpal=None, bg=0, transparent=None, interlace=False, ext=b"", version=b"89a"):
    fx, fy, fw, fh, idx = frame
    bits = lambda pal: max(1, int(np.ceil(np.log2(len(pal))))) - 1
    tab = lambda pal: b"".join(bytes(c) for c in pal) + b"\0\0\0" * ((2 << bits(pal)) - len(pal))
    out = b"GIF" + version + struct.pack("<HHBBB", W, H, (0x80 | bits(gpal)) if gpal else 0, bg, 0)
    if gpal:
        out += tab(gpal)
    out += ext
    if transparent is not None:
        out += b"\x21\xf9\x04" + struct.pack("<BHB", 1, 0, transparent) + b"\0"
    out += b"\x2c" + struct.pack("<HHHHB", fx, fy, fw, fh, (0x80 | bits(lpal) if lpal else 0) | (0x40 if interlace else 0))
    if lpal:
        out += tab(lpal)
    pal = lpal or gpal
    rows = idx.reshape(fh, fw)
    if interlace:
        order = [y for s, st in ((0, 8), (4, 8), (2, 4), (1, 2)) for y in range(s, fh, st)]
        rows = rows[order]
    return out + _gif_lzw(rows.reshape(-1), max(2, bits(pal) + 1)) + b"\x3b"


@pytest.mark.parametrize("W,H", [(1, 1), (9, 7), (40, 23), (130, 70)])
def test_gif_first_frame_equals_reference_decoder(clip_lib, stb, tmp_path, W, H):
    rng = np.random.default_rng(W * H)
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.gif", **kw)
    for n_col in (2, 4, 16, 200, 256):
        pal = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in range(n_col)]
        idx = rng.integers(0, n_col, W * H)
        idx[rng.random(W * H) < 0.5] = idx[0]                                       # runs -> long table strings, KwKwK codes
        got = same(_gif(W, H, (0, 0, W, H, idx), gpal=pal))
        assert np.array_equal(got, np.array(pal, dtype=np.uint8)[idx].reshape(H, W, 3))
        same(_gif(W, H, (0, 0, W, H, idx), gpal=pal, interlace=True))
        same(_gif(W, H, (0, 0, W, H, idx), gpal=pal, transparent=int(idx[0])))     # transparent pixels stay black
        same(_gif(W, H, (0, 0, W, H, idx), gpal=pal, transparent=int(idx[0]), bg=1))
        same(_gif(W, H, (0, 0, W, H, idx), lpal=pal, version=b"87a"))              # local table only
        same(_gif(W, H, (0, 0, W, H, idx), gpal=pal[::-1], lpal=pal, transparent=0, bg=n_col - 1, interlace=True))
        same(_gif(W, H, (0, 0, W, H, idx), gpal=pal, ext=b"\x21\xfe\x05hello\0" + b"\x21\xff\x0bNETSCAPE2.0\x03\x01\0\0\0"))
        if W > 2 and H > 2:
            # a frame smaller than the screen: the rest takes the background colour (index != 0) or stays black (index 0)
            sub = rng.integers(0, n_col, (W - 2) * (H - 2))
            for bg in (0, 1, n_col - 1):
                same(_gif(W, H, (1, 1, W - 2, H - 2, sub), gpal=pal, bg=bg))
                same(_gif(W, H, (2, 1, W - 2, H - 2, sub), gpal=pal, bg=bg, transparent=int(sub[0]), interlace=True))
            same(_gif(W, H, (2, 2, W - 1, H - 1, rng.integers(0, n_col, (W - 1) * (H - 1))), gpal=pal), must_load=False)   # frame outside the screen
    # what PIL writes (its own LZW encoder, optimised palettes, interlace flag)
    pim = PIL.fromarray(_photo(H, W, seed=3))
    same(_pil_bytes(pim.convert("P"), "GIF"))
    same(_pil_bytes(pim.convert("P"), "GIF", interlace=True))
    same(_pil_bytes(pim.convert("L"), "GIF"))
    same(_pil_bytes(pim.convert("P", palette=1, colors=17), "GIF", transparency=3))
    # no image at all / no colour table / truncated raster
    same(b"GIF89a" + struct.pack("<HHBBB", W, H, 0, 0, 0) + b"\x3b", must_load=False)
    same(b"GIF89a" + struct.pack("<HHBBB", W, H, 0, 0, 0) + b"\x2c" + struct.pack("<HHHHB", 0, 0, W, H, 0) + b"\x02\x02\x4c\x01\0\x3b", must_load=False)
    data = _gif(W, H, (0, 0, W, H, rng.integers(0, 16, W * H)), gpal=[(i * 16, 255 - i * 16, i) for i in range(16)])
    same(data[:len(data) * 2 // 3], must_load=False)


def _packbits(row):
    out, i = bytearray(), 0
    while i < len(row):
        run = 1
        while i + run < len(row) and run < 128 and row[i + run] == row[i]:
            run += 1
        if run > 1:
            out += bytes((257 - run, row[i]))
            i += run
        else:
            lit = 1
            while i + lit < len(row) and lit < 128 and row[i + lit] != row[i + lit - 1]:
                lit += 1
            out += bytes((lit - 1,)) + bytes(row[i:i + lit])
            i += lit
    return bytes(out)


def _psd(planes, depth=8, rle=False, mode=3, resources=b"\x01\x02\x03\x04"):
    """planes: [channels][h][w] uint8 / uint16."""
    ch, h, w = planes.shape
    hdr = b"8BPS" + struct.pack(">H6xHIIHH", 1, ch, h, w, depth, mode)
    hdr += struct.pack(">I", 0) + struct.pack(">I", len(resources)) + resources + struct.pack(">I", 0)
    if not rle:
        return hdr + struct.pack(">H", 0) + planes.astype(">u2" if depth == 16 else "u1").tobytes()
    rows = [_packbits(bytes(planes[c, y])) for c in range(ch) for y in range(h)]
    return hdr + struct.pack(">H", 1) + b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows)


@pytest.mark.parametrize("w,h", [(1, 1), (13, 5), (64, 40)])
def test_psd_composite_equals_reference_decoder(clip_lib, stb, tmp_path, w, h):
    rng = np.random.default_rng(w * h + 1)
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.psd", **kw)
    p8 = rng.integers(0, 256, (5, h, w), dtype=np.uint8)
    p8[:, rng.random((h, w)) < 0.5] = 200                                           # runs for PackBits
    p8[3, rng.random((h, w)) < 0.3] = 255
    p8[3, rng.random((h, w)) < 0.2] = 0
    for ch in (1, 2, 3, 4, 5):                                                      # missing channels default to 0 (alpha: opaque); a fourth one un-mattes
        for rle in (False, True):
            got = same(_psd(p8[:ch], rle=rle))
            if ch == 3:
                assert np.array_equal(got, np.moveaxis(p8[:3], 0, -1))
    p16 = rng.integers(0, 65536, (4, h, w), dtype=np.uint16)
    same(_psd(p16[:3], depth=16))
    same(_psd(p16, depth=16))
    same(_psd(p8[:3], mode=1), must_load=False)                                    # grey-scale mode: refused by both
    same(_psd(p8[:3], depth=1), must_load=False)
    same(_psd(p8[:3])[:-(w * h) // 2 - 1], must_load=False)                         # truncated planes read as zeros


@pytest.mark.parametrize("w,h", [(1, 1), (9, 4), (40, 11)])
def test_pnm_any_maxval_equals_reference_decoder(clip_lib, stb, tmp_path, w, h):
    rng = np.random.default_rng(w * h)
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.pnm", **kw)
    for ch, magic in ((1, b"P5"), (3, b"P6")):
        v8 = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        v16 = rng.integers(0, 65536, (h, w, ch), dtype=np.uint16)
        for maxval, body in ((255, v8.tobytes()), (15, (v8 & 15).tobytes()), (1, (v8 & 1).tobytes()), (65535, v16.astype(">u2").tobytes()), (1023, (v16 & 1023).astype(">u2").tobytes())):
            same(magic + b"\n%d %d\n%d\n" % (w, h, maxval) + body)
            same(magic + b" # a comment\n%d\t%d # another\r\n%d " % (w, h, maxval) + body)
        same(magic + b"\n%d %d\n255\n" % (w, h) + v8.tobytes()[:-1], must_load=False)
        same(magic + b"\n%d %d\n65536\n" % (w, h) + v16.astype(">u2").tobytes(), must_load=False)
        same(magic + b"\n0 %d\n255\n" % h, must_load=False)
        same(magic + b"\n99999999999 %d\n255\n" % h, must_load=False)


def _jpeg_patch(data, drop_app0=False, adobe_transform=None, ids=None):
    """Rewrite markers of a JPEG: remove the JFIF APP0, insert / change the Adobe APP14 transform, rename the component ids."""
    out, i = bytearray(data[:2]), 2
    if adobe_transform is not None and b"Adobe" not in data:
        out += b"\xff\xee" + struct.pack(">H", 14) + b"Adobe" + struct.pack(">HHHB", 100, 0, 0, adobe_transform)
    while i < len(data):
        assert data[i] == 0xFF
        m = data[i + 1]
        if m == 0xDA:
            seg = bytearray(data[i:])
            j = 0
            while ids and j >= 0:                      # every scan header (progressive files have several; FF DA cannot occur inside entropy-coded data)
                for k in range(seg[j + 4]):
                    seg[j + 5 + 2 * k] = ids[seg[j + 5 + 2 * k] - 1]
                j = seg.find(b"\xff\xda", j + 2)
            out += seg
            break
        ln = struct.unpack(">H", data[i + 2:i + 4])[0]
        seg = bytearray(data[i:i + 2 + ln])
        if m == 0xE0 and drop_app0:
            seg = b""
        if m == 0xEE and adobe_transform is not None and seg[4:9] == b"Adobe":
            seg[15] = adobe_transform
        if m in (0xC0, 0xC2) and ids:
            for k in range(seg[9]):
                seg[10 + 3 * k] = ids[seg[10 + 3 * k] - 1]
        out += seg
        i += 2 + ln
    return bytes(out)


@pytest.mark.parametrize("size", [(16, 16), (41, 29), (3, 70)])
@pytest.mark.parametrize("kw", [dict(), dict(progressive=True), dict(subsampling=0), dict(quality=30, restart_marker_blocks=2)])
def test_jpeg_four_components_and_rgb_equal_reference_decoder(clip_lib, stb, tmp_path, size, kw):
    same = lambda data, **k: _same_as_reference(clip_lib, stb, tmp_path, data, "t.jpg", **k)
    img = _photo(size[0], size[1], seed=11)
    cmyk = _pil_bytes(PIL.fromarray(img).convert("CMYK"), "JPEG", **{"quality": 85, **kw})
    assert b"Adobe" in cmyk
    same(cmyk)                                                                      # CMYK as PIL / libjpeg write it
    for tr in (0, 1, 2):                                                            # Adobe transform: CMYK / (YCbCr + ignored channel) / YCCK
        same(_jpeg_patch(cmyk, adobe_transform=tr))
    rgb = _pil_bytes(PIL.fromarray(img), "JPEG", **{"quality": 85, **kw})
    same(_jpeg_patch(rgb, ids=b"RGB"))                                              # component ids 'R' 'G' 'B': samples are RGB
    same(_jpeg_patch(rgb, ids=b"RGb"))
    same(_jpeg_patch(rgb, adobe_transform=0))                                       # Adobe transform 0 in a JFIF file: still YCbCr
    same(_jpeg_patch(rgb, adobe_transform=0, drop_app0=True))                       # ... without JFIF: RGB
    same(_jpeg_patch(rgb, adobe_transform=1, drop_app0=True))
    same(_jpeg_patch(rgb, drop_app0=True))


def _png(w, h, depth, ctype, filtered_rows, plte=None, trns=None, interlace=0):
    import zlib
    chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if plte is not None:
        out += chunk(b"PLTE", plte)
    if trns is not None:
        out += chunk(b"tRNS", trns)
    return out + chunk(b"IDAT", zlib.compress(filtered_rows)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (5, 3), (8, 8), (17, 9), (33, 13)])
def test_png_pixels_equal_reference_decoder(clip_lib, stb, tmp_path, w, h):
    """Random filtered scanlines of every colour type x bit depth x filter type, with and without tRNS, plain and Adam7 (which PIL cannot write)."""
    rng = np.random.default_rng(w * 64 + h)
    same = lambda data: _same_as_reference(clip_lib, stb, tmp_path, data, "t.png")
    xs, ys, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    for ctype, chans in ((0, 1), (2, 3), (3, 1), (4, 2), (6, 4)):
        for depth in (1, 2, 4, 8, 16):
            if (ctype in (2, 4, 6) and depth < 8) or (ctype == 3 and depth == 16):
                continue
            plte = bytes(rng.integers(0, 256, 3 << min(depth, 8), dtype=np.uint8)) if ctype == 3 else None
            trns = {0: b"\x00\x01", 2: b"\x00\x01\x00\x02\x00\x03", 3: b"\x00\x80"}.get(ctype)
            row = lambda pw, f: bytes((f,)) + bytes(rng.integers(0, 256, (pw * chans * depth + 7) // 8, dtype=np.uint8))
            for f in range(5):
                rows = b"".join(row(w, f) for _ in range(h))
                same(_png(w, h, depth, ctype, rows, plte))
                if trns:
                    same(_png(w, h, depth, ctype, rows, plte, trns))
            raw = b""
            for p in range(7):
                pw, ph = (w - xs[p] + dx[p] - 1) // dx[p], (h - ys[p] + dy[p] - 1) // dy[p]
                if pw > 0 and ph > 0:
                    raw += b"".join(row(pw, int(rng.integers(0, 5))) for _ in range(ph))
            same(_png(w, h, depth, ctype, raw, plte, interlace=1))


def _pic(w, h, packets, rows):
    """packets: [(type, channel mask)]; rows[y][k] = the bytes of packet k on scanline y."""
    hdr = b"\x53\x80\xf6\x34" + struct.pack(">f", 3.71) + b"made by a test".ljust(80, b"\0") + b"PICT" + struct.pack(">HHfHH", w, h, 1.0, 3, 0)
    pk = b"".join(struct.pack("BBBB", 1 if i + 1 < len(packets) else 0, 8, t, ch) for i, (t, ch) in enumerate(packets))
    return hdr + pk + b"".join(b"".join(r) for r in rows)


def _pic_encode(vals, type_, rng):
    """vals: [w][n channels] uint8 of one packet on one scanline."""
    w = len(vals)
    if type_ == 0:
        return vals.tobytes()
    out, i = bytearray(), 0
    while i < w:
        run = 1
        while i + run < w and run < 255 and np.array_equal(vals[i + run], vals[i]):
            run += 1
        if type_ == 1:
            out += bytes((run,)) + vals[i].tobytes()
            i += run
        elif run > 1 and rng.random() < 0.8:
            if rng.random() < 0.3:
                out += b"\x80" + struct.pack(">H", run) + vals[i].tobytes()        # 16-bit run length
            else:
                run = min(run, 128)
                out += bytes((127 + run,)) + vals[i].tobytes()
            i += run
        else:
            lit = int(min(w - i, rng.integers(1, 129)))
            out += bytes((lit - 1,)) + vals[i:i + lit].tobytes()
            i += lit
    return bytes(out)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (300, 5)])
def test_softimage_pic_equals_reference_decoder(clip_lib, stb, tmp_path, w, h):
    rng = np.random.default_rng(w + 7 * h)
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.pic", **kw)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[:, rng.random(w) < 0.6] = img[0, 0]
    for layout in ([(0, 0xE0)], [(1, 0xE0)], [(2, 0xE0)], [(2, 0xE0), (2, 0x10)], [(0, 0x80), (1, 0x40), (2, 0x20)], [(2, 0xF0)], [(2, 0x60)]):
        sel = lambda ch: [i for i, m in enumerate((0x80, 0x40, 0x20, 0x10)) if ch & m]
        rows = [[_pic_encode(np.ascontiguousarray(img[y][:, sel(ch)]), t, rng) for t, ch in layout] for y in range(h)]
        got = same(_pic(w, h, layout, rows))
        if layout[0][1] == 0xE0:
            assert np.array_equal(got, img[:, :, :3])
    # refused by ours; the reference crashes on a corrupt PIC (it converts a NULL image), so these are not handed to it
    for bad in (_pic(w, h, [(3, 0xE0)], [[b"\0" * 3 * w]] * h), _pic(w, h, [(0, 0xE0)], [[img[y, :, :3].tobytes()] for y in range(h)])[:-2]):
        path = str(tmp_path / "bad.pic")
        open(path, "wb").write(bad)
        assert load_ours(clip_lib, path) is None


def _rgbe(rgbf):
    """float RGB [n][3] -> RGBE bytes [n][4] (Ward's shared-exponent encoding)."""
    m = rgbf.max(axis=1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0).astype(int)
    out = np.zeros((len(rgbf), 4), dtype=np.uint8)
    ok = m > 1e-32
    out[ok, :3] = np.clip(rgbf[ok] * (256.0 / (2.0 ** e[ok]))[:, None], 0, 255).astype(np.uint8)
    out[ok, 3] = (e[ok] + 128).astype(np.uint8)
    return out


def _hdr(w, h, px, rle, ident=b"#?RADIANCE", extra=b"EXPOSURE=1.0\n"):
    head = ident + b"\n" + extra + b"FORMAT=32-bit_rle_rgbe\n\n" + b"-Y %d +X %d\n" % (h, w)
    if not rle:
        return head + px.tobytes()
    body = bytearray()
    for y in range(h):
        row = px[y * w:(y + 1) * w]
        body += bytes((2, 2, w >> 8, w & 255))
        for k in range(4):
            col, i = row[:, k], 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and col[i + run] == col[i]:
                    run += 1
                if run > 2:
                    body += bytes((128 + run, col[i]))
                    i += run
                else:
                    lit = 1
                    while i + lit < w and lit < 128 and not (i + lit + 2 < w and col[i + lit] == col[i + lit + 1] == col[i + lit + 2]):
                        lit += 1
                    body += bytes((lit,)) + col[i:i + lit].tobytes()
                    i += lit
    return head + bytes(body)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 4), (8, 3), (200, 6)])
def test_radiance_hdr_equals_reference_decoder(clip_lib, stb, tmp_path, w, h):
    rng = np.random.default_rng(w * 3 + h)
    same = lambda data, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, "t.hdr", **kw)
    rgbf = np.exp(rng.normal(-1.0, 2.0, (w * h, 3)))                                  # 0.001 ... 50: under-, mid- and over-exposed
    rgbf[rng.random(w * h) < 0.1] = 0.0
    rgbf[rng.random(w * h) < 0.4] = rgbf[0]
    px = _rgbe(rgbf)
    same(_hdr(w, h, px, rle=False))                                                  # flat (also what a wide file may hold: "not a packed scanline")
    same(_hdr(w, h, px, rle=False, ident=b"#?RGBE", extra=b""))
    if w >= 8:
        same(_hdr(w, h, px, rle=True))
        same(_hdr(w, h, px, rle=True, extra=b"# comment line\nGAMMA=1\nPRIMARIES=0.640 0.330 0.290 0.600 0.150 0.060 0.333 0.333\n"))
        # a later scanline that is not packed: the reference restarts flat at pixel 0 from there
        first = _hdr(w, 1, px[:w], rle=True)
        head = _hdr(w, h, px, rle=False)[:-(w * h * 4)]
        one_packed_row = first[len(_hdr(w, 1, px[:w], rle=False)) - w * 4:]
        same(head + one_packed_row + px[w:].tobytes() + bytes(4 * w), must_load=False)
    same(_hdr(w, h, px, rle=False).replace(b"FORMAT=32-bit_rle_rgbe", b"FORMAT=32-bit_rle_xyze"), must_load=False)
    same(_hdr(w, h, px, rle=False).replace(b"-Y ", b"+Y "), must_load=False)


def test_out_of_spec_files_decode_like_the_reference_decoder(clip_lib, stb, tmp_path):
    """Files a careless writer or a flipped bit produces, which the reference still decodes — to well-defined pixels: quantisation tables that
    push the IDCT past 16 bits (its SSE2 kernel wraps / saturates there), a DC term out of range (refused), PNGs with a wrong Adler-32 or with
    more IDAT data than the image needs or an IEND of any length, a BMP whose compression field is negative, an HDR that ends early."""
    import zlib
    same = lambda data, name, **kw: _same_as_reference(clip_lib, stb, tmp_path, data, name, **kw)
    img = _photo(23, 31, seed=2)
    for kw in (dict(), dict(progressive=True), dict(subsampling=0)):
        for pim in (PIL.fromarray(img), PIL.fromarray(img).convert("L")):
            s = _pil_bytes(pim, "JPEG", quality=85, **kw)
            i = s.index(b"\xff\xdb")
            for off in (5, 6, 20, 68):
                for v in (0, 1, 127, 128, 200, 255):
                    d = bytearray(s)
                    d[i + off] = v
                    same(bytes(d), "q.jpg", must_load=False)
    rows = b"".join(b"\0" + bytes(img[y].tobytes()) for y in range(23))
    good = _png(31, 23, 8, 2, rows)
    z0 = good.index(b"IDAT") + 4
    zlen = struct.unpack(">I", good[z0 - 8:z0 - 4])[0]
    bad_adler = bytearray(good)
    bad_adler[z0 + zlen - 1] ^= 0x55
    assert np.array_equal(same(bytes(bad_adler), "a.png"), img)
    assert np.array_equal(same(_png(31, 23, 8, 2, rows + bytes(500)), "s.png"), img)                     # surplus scanline data
    assert np.array_equal(same(good[:-12] + struct.pack(">I", 77) + b"IEND" + b"\0\0\0\0", "e.png"), img)
    same(good.replace(b"IEND", b"JUNK"), "u.png", must_load=False)                                        # unknown critical chunk
    bmp = bytearray(_bmp(31, 23, 24, [img[y, :, ::-1].tobytes() for y in range(23)]))
    bmp[30:34] = struct.pack("<i", -5)
    assert np.array_equal(same(bytes(bmp), "n.bmp"), img)
    px = _rgbe(np.exp(np.random.default_rng(1).normal(-1.0, 2.0, (5 * 6, 3))))
    hdr = _hdr(5, 6, px, rle=False)
    same(hdr[:-4 * 7 - 2], "short.hdr")                                                                   # the last full pixel repeats
