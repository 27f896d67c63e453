"""clip_image_load_from_file: own PNM/BMP/PNG/JPEG decoders.  JPEG pixels must equal what the REFERENCE's decoder
(its vendored stb_image, built from /root/reference into oracle/_ref/ by `make -C oracle ref`) produces, because
they feed the bit-exact preprocessing; PNG/BMP/PNM are lossless and are checked against PIL / numpy."""
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
