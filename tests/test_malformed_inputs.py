"""CPU tier: the two parsers of untrusted files — the image decoders behind clip_image_load_from_file and the GGUF reader / loader /
quantizer behind clip_model_load and clip_model_quantize — take a few hundred deterministically mutated files without crashing (they
return false / NULL or a decoded result).  The sanitizer form of the same mutations is scripts/fuzz/run.sh (ASan + UBSan; round 4: 336 k
mutated images and 13.5 k mutated GGUF files, findings fixed: the DRI length check that rejected every JPEG with restart intervals, the
float -> byte conversions of the quantizer on non-finite weights, 32-bit IDCT products of corrupt coefficients)."""
import os
import re
import subprocess
import sys

import pytest

from oracle import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, io, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
os.environ["CLIP_AMD_ALLOW_NO_DEVICE"] = "1"
os.environ["CLIP_AMD_WEIGHT_CACHE"] = "0"
import clip_cpp_amd
L = clip_cpp_amd.lib()
rng = np.random.default_rng(%(seed)d)
tmp = %(tmp)r

def mutate(d):
    d = bytearray(d)
    kind = int(rng.integers(0, 6))
    for _ in range(int(rng.integers(1, 6))):
        if not d:
            break
        pos = int(rng.integers(0, min(len(d), 700 if rng.integers(0, 3) == 0 else len(d))))
        if kind == 0:
            d[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            d[pos] = int(rng.integers(0, 256))
        elif kind == 2:
            d[pos] = 0xFF if rng.integers(0, 2) else 0
        elif kind == 3:
            del d[pos:]
        elif kind == 4:
            del d[pos:pos + int(rng.integers(1, 17))]
        else:
            d[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 17)), dtype=np.uint8))
    return bytes(d)

mode = %(mode)r
done = ok = 0
if mode == "images":
    import PIL.Image as I
    yy, xx = np.mgrid[0:29, 0:41]
    im = np.clip(np.stack([(np.sin(xx / 9.0) * .5 + .5) * 255, (np.cos(yy / 7.0) * .5 + .5) * 255, (3 * xx + yy) %% 256], -1)
                 + rng.normal(0, 8, (29, 41, 3)), 0, 255).astype(np.uint8)
    P = I.fromarray(im)
    seeds = []
    for fmt, kw in (("JPEG", dict(quality=85)), ("JPEG", dict(quality=70, progressive=True, restart_marker_blocks=2)), ("JPEG", dict(subsampling=0, restart_marker_blocks=1)),
                    ("PNG", {}), ("BMP", {}), ("PPM", {})):
        b = io.BytesIO(); P.save(b, fmt, **kw); seeds.append(b.getvalue())
    b = io.BytesIO(); P.convert("P").save(b, "PNG"); seeds.append(b.getvalue())
    # image_formats.cpp: paletted BMP, TGA (RLE), GIF, CMYK JPEG, and hand-made PSD / HDR / PIC
    for mode_, fmt, kw in (("P", "BMP", {}), ("RGBA", "TGA", dict(compression="tga_rle")), ("P", "GIF", {}), ("CMYK", "JPEG", {})):
        b = io.BytesIO(); P.convert(mode_).save(b, fmt, **kw); seeds.append(b.getvalue())
    import struct
    seeds.append(b"8BPS" + struct.pack(">H6xHIIHH", 1, 4, 29, 41, 8, 3) + struct.pack(">III", 0, 0, 0) + struct.pack(">H", 0) + np.moveaxis(np.concatenate([im, im[:, :, :1] // 2 + 90], -1), -1, 0).tobytes())
    seeds.append(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 29 +X 41\n" + b"".join(bytes((2, 2, 0, 41)) + b"".join(bytes((41,)) + np.ascontiguousarray(im[y, :, k]).tobytes() for k in range(3)) + bytes((128 + 41, 129)) for y in range(29)))
    seeds.append(b"\x53\x80\xf6\x34" + bytes(84) + b"PICT" + struct.pack(">HHfHH", 41, 29, 1.0, 3, 0) + bytes((0, 8, 0, 0xE0)) + im.tobytes())
    for s in seeds:
        for it in range(%(iters)d):
            open(tmp, "wb").write(s if it == 0 else mutate(s))
            img = L.clip_image_u8_make()
            if L.clip_image_load_from_file(os.fsencode(tmp), img):
                c = img.contents
                assert c.nx > 0 and c.ny > 0 and c.size == 3 * c.nx * c.ny
                int(np.ctypeslib.as_array(c.data, shape=(c.size,)).sum())          # every byte readable
                ok += 1
            L.clip_image_u8_free(img)
            done += 1
else:
    for path in %(models)r:
        s = open(path, "rb").read()
        for it in range(%(iters)d):
            d = s if it == 0 else mutate(s[:200000]) + s[200000:]
            open(tmp, "wb").write(d)
            ctx = L.clip_model_load(os.fsencode(tmp), 0)
            if ctx:
                ok += 1
                L.clip_free(ctx)
            if it %% 5 == 0:
                L.clip_model_quantize(os.fsencode(tmp), os.fsencode(tmp + ".q"), 2)
            done += 1
print("DONE %%d %%d" %% (done, ok))
"""


def _run(mode, tmp_path, iters, models=()):
    code = CHILD % dict(root=ROOT, seed=20250925, tmp=str(tmp_path / "mut.bin"), mode=mode, iters=iters, models=list(models))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, errors="replace", timeout=900)   # (corrupt tensor names reach stderr verbatim)
    last = re.findall(r"DONE (\d+) (\d+)", r.stdout)            # (the library's own printf lines share the pipe)
    assert r.returncode == 0 and last, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return int(last[-1][0]), int(last[-1][1])


def test_mutated_image_files_never_crash_the_decoders(clip_lib, tmp_path):
    pytest.importorskip("PIL.Image")
    done, ok = _run("images", tmp_path, 120)
    assert done == 14 * 120 and 14 <= ok < done        # the unmutated seeds decode, some mutations do not


def test_mutated_gguf_files_never_crash_load_or_quantize(clip_lib, tmp_path, fixture_cache):
    models = [fixtures.cached_model(fixture_cache, "tiny", "q4_1", text=False, vision=True),
              fixtures.cached_model(fixture_cache, "tiny", "f32", text=False, vision=True)]
    done, ok = _run("gguf", tmp_path, 100, models)
    assert done == 200 and 2 <= ok < done


def test_block_count_larger_than_the_file_is_refused(clip_lib, tmp_path, fixture_cache):
    """A block count the file's tensors cannot cover (hostile metadata: round-5 sanitizer finding, a 343 GB std::vector) fails the load cleanly."""
    import struct
    path = fixtures.cached_model(fixture_cache, "tiny", "q4_1", text=False, vision=True)
    d = bytearray(open(path, "rb").read())
    key = b"clip.vision.block_count"
    at = d.index(key) + len(key)
    assert struct.unpack("<I", d[at:at + 4])[0] == 4                     # GGUF value type: u32
    n = struct.unpack("<I", d[at + 4:at + 8])[0]
    code = """
import os, sys
sys.path.insert(0, %r)
os.environ["CLIP_AMD_ALLOW_NO_DEVICE"] = "1"
os.environ["CLIP_AMD_WEIGHT_CACHE"] = "0"
import clip_cpp_amd
L = clip_cpp_amd.lib()
for v in (%d, 0x7fffffff, 0x50000000):
    d = bytearray(open(%r, "rb").read())
    d[%d:%d] = v.to_bytes(4, "little")
    open(%r, "wb").write(d)
    ctx = L.clip_model_load(os.fsencode(%r), 0)
    print("LOADED" if ctx else "REFUSED", v)
    if ctx: L.clip_free(ctx)
""" % (ROOT, n, path, at + 4, at + 8, str(tmp_path / "bc.gguf"), str(tmp_path / "bc.gguf"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, errors="replace", timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = re.findall(r"(LOADED|REFUSED) (\d+)", r.stdout)
    assert out == [("LOADED", str(n)), ("REFUSED", str(0x7fffffff)), ("REFUSED", str(0x50000000))], r.stdout[-500:] + r.stderr[-500:]
