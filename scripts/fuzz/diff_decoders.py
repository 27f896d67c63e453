#!/usr/bin/env python3
"""Differential mutation run of clip_image_load_from_file against the REFERENCE's decoder (oracle/_ref/libstb_ref.so, built from the
reference tree by `make -C oracle ref`): a few header-biased byte mutations per file; both decoders either refuse the file or must return
the same pixels.  Dev-container tool (needs the reference build); CPU only.
    python scripts/fuzz/diff_decoders.py [format ...] [--seed N] [--iters N]
Classes that remain by construction and are not findings: corrupt entropy-coded data (the two error recoveries differ), files the
reference returns UNINITIALISED memory for (JPEG cut inside its headers or at a missing restart marker, truncated raw TGA rows, colour
table entries a BMP never defines), this loader's 2^28-pixel cap, zero-sized HDR images."""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import PIL.Image as I                                                                   # noqa: E402
import clip_cpp_amd                                                                     # noqa: E402
import test_image_io as T                                                               # noqa: E402  (fixture builders)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("formats", nargs="*", default=["png", "jpg", "bmp", "gif", "tga", "psd", "pnm", "hdr"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so"))
    ref.stbref_load_from_memory.restype = C.POINTER(C.c_ubyte)
    ref.stbref_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    ref.stbref_free.argtypes = [C.c_void_p]
    lib = clip_cpp_amd.lib()
    tmp = "/tmp/diff_decoders_%d.bin" % os.getpid()

    def theirs(data):
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        p = ref.stbref_load_from_memory(data, len(data), C.byref(x), C.byref(y), C.byref(c))
        if not p:
            return None
        a = np.ctypeslib.as_array(p, shape=(y.value, x.value, 3)).copy()
        ref.stbref_free(p)
        return a

    def ours(data):
        open(tmp, "wb").write(data)
        img = lib.clip_image_u8_make()
        try:
            if not lib.clip_image_load_from_file(tmp.encode(), img):
                return None
            c = img.contents
            return np.ctypeslib.as_array(c.data, shape=(c.ny, c.nx, 3)).copy()
        finally:
            lib.clip_image_u8_free(img)

    rng = np.random.default_rng(args.seed)
    im = T._photo(23, 31, seed=2)
    P = I.fromarray(im)
    pb = T._pil_bytes
    seeds = {
        "jpg": [pb(P, "JPEG", quality=85), pb(P, "JPEG", quality=60, progressive=True), pb(P.convert("CMYK"), "JPEG"), pb(P.convert("L"), "JPEG"),
                pb(P, "JPEG", subsampling=0, restart_marker_blocks=2)],
        "png": [pb(P, "PNG"), pb(P.convert("P"), "PNG"), pb(P.convert("LA"), "PNG"), pb(P.convert("1"), "PNG")],
        "bmp": [pb(P, "BMP"), pb(P.convert("P"), "BMP"), pb(P.convert("1"), "BMP"), pb(P.convert("RGBA"), "BMP"),
                T._bmp(31, 23, 16, [bytes(rng.integers(0, 256, 62, dtype=np.uint8)) for _ in range(23)], comp=3, masks=(0xF800, 0x7E0, 0x1F))],
        "gif": [pb(P.convert("P"), "GIF"), pb(P.convert("P"), "GIF", interlace=True), pb(P.convert("P", palette=1, colors=17), "GIF", transparency=3)],
        "tga": [pb(P, "TGA"), pb(P.convert("RGBA"), "TGA", compression="tga_rle"), pb(P.convert("P"), "TGA"), pb(P.convert("LA"), "TGA")],
        "psd": [T._psd(np.moveaxis(np.concatenate([im, im[:, :, :1] // 2 + 60], -1), -1, 0), rle=True), T._psd(np.moveaxis(im, -1, 0))],
        "pnm": [b"P6\n31 23\n255\n" + im.tobytes(), b"P5\n31 23\n65535\n" + im[:, :, :2].tobytes()],
        "hdr": [T._hdr(31, 23, T._rgbe(np.exp(rng.normal(-1, 2, (31 * 23, 3)))), rle=True), T._hdr(5, 4, T._rgbe(np.exp(rng.normal(-1, 2, (20, 3)))), rle=False)],
    }   # (no PIC seeds: the reference dereferences NULL on a corrupt PIC)
    for fmt in args.formats:
        stats = collections.Counter()
        for s in seeds[fmt]:
            for _ in range(args.iters):
                d = bytearray(s)
                for _ in range(int(rng.integers(1, 4))):
                    pos = int(rng.integers(0, min(len(d), 120 if rng.random() < 0.7 else len(d))))
                    k = int(rng.integers(0, 3))
                    d[pos] = d[pos] ^ (1 << int(rng.integers(0, 8))) if k == 0 else int(rng.integers(0, 256)) if k == 1 else int(rng.choice([0, 1, 2, 3, 4, 8, 16, 24, 32, 127, 128, 255]))
                a, b = theirs(bytes(d)), ours(bytes(d))
                stats["both refuse" if a is None and b is None else "only ours reads" if a is None else "only the reference reads" if b is None else
                      "equal" if a.shape == b.shape and np.array_equal(a, b) else "pixels differ"] += 1
        print(fmt, dict(stats))
    os.remove(tmp)


if __name__ == "__main__":
    main()
