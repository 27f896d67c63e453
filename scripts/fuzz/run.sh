#!/bin/bash
# ASan + UBSan mutation runs over the two parsers of untrusted files: the image decoders (clip_image_load_from_file: PNG / BMP / GIF / PSD /
# PIC / JPEG / PNM / HDR / TGA) and the GGUF reader + loader + quantizer (clip_model_load on a host-only context, clip_model_quantize).  CPU only.
#   scripts/fuzz/run.sh [iterations-per-seed] [rng-seed]
# Host sources are compiled as C++ with clang's sanitizers and linked against the kernel objects of the normal build
# (python -m clip_cpp_amd.build first).  Seeds: small PIL-written images, the tiny fixtures of oracle/fixtures.py.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
W=${FUZZ_DIR:-/tmp/clip_amd_fuzz}; IT=${1:-2000}; SEED=${2:-1}
CL=/opt/rocm/lib/llvm/bin/clang++
FL="-std=c++17 -g -O1 -fwrapv -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$ROOT/clip_cpp_amd/csrc"
mkdir -p "$W/obj" "$W/seeds"
for f in gguf quant load forward tokenizer preprocess image_io image_formats jpeg_decode host_pipeline api; do
    $CL $FL -c "$ROOT/clip_cpp_amd/csrc/$f.cpp" -o "$W/obj/$f.o" &
done; wait
$CL $FL "$ROOT/scripts/fuzz/fuzz_images.cpp" "$W/obj/image_io.o" "$W/obj/image_formats.o" "$W/obj/jpeg_decode.o" -lz -o "$W/fuzz_images"
$CL $FL "$ROOT/scripts/fuzz/fuzz_model.cpp" "$W"/obj/*.o $(ls "$ROOT"/clip_cpp_amd/build/*.o | grep -E "/k_") -L/opt/rocm/lib -lamdhip64 -lz -lpthread -ldl -Wl,-rpath,/opt/rocm/lib -o "$W/fuzz_model"
python3 - "$W/seeds" <<'PY'
import sys, numpy as np, PIL.Image as I
d = sys.argv[1]; rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:37, 0:53]
im = np.clip(np.stack([(np.sin(xx / 9.0) * .5 + .5) * 255, (np.cos(yy / 7.0) * .5 + .5) * 255, (3 * xx + yy) % 256], -1) + rng.normal(0, 8, (37, 53, 3)), 0, 255).astype(np.uint8)
P = I.fromarray(im)
P.save(d + "/a.jpg", quality=90); P.save(d + "/b.jpg", quality=75, progressive=True); P.save(d + "/c.jpg", quality=85, subsampling=0)
P.save(d + "/d.jpg", quality=60, subsampling=2, restart_marker_blocks=3); P.save(d + "/e.jpg", quality=50, progressive=True, restart_marker_blocks=2)
P.convert("L").save(d + "/f.jpg"); P.convert("CMYK").save(d + "/g.jpg")
P.save(d + "/h.png"); P.convert("P").save(d + "/i.png"); P.convert("RGBA").save(d + "/j.png"); P.convert("LA").save(d + "/k.png"); P.convert("1").save(d + "/l.png")
I.fromarray((im.astype(np.uint16) * 257)[:, :, 0]).save(d + "/m.png")
P.save(d + "/n.bmp"); P.convert("RGBA").save(d + "/o.bmp"); P.save(d + "/p.ppm")
# image_formats.cpp: paletted / 1-bit BMP, TGA (raw, RLE, colour-mapped, grey + alpha), GIF (plain, interlaced + transparent), 16-bit PGM, PSD (PackBits, 4 channels)
P.convert("P").save(d + "/q.bmp"); P.convert("1").save(d + "/r.bmp")
P.save(d + "/s.tga"); P.convert("RGBA").save(d + "/t.tga", compression="tga_rle"); P.convert("P").save(d + "/u.tga"); P.convert("LA").save(d + "/v.tga", compression="tga_rle")
P.convert("P").save(d + "/w.gif"); P.convert("P", palette=1, colors=17).save(d + "/x.gif", interlace=True, transparency=3)
open(d + "/y.pgm", "wb").write(b"P5\n53 37\n65535\n" + (im[:, :, 0].astype(">u2") * 257).tobytes())
import struct
def packbits(row):
    out, i = bytearray(), 0
    while i < len(row):
        run = 1
        while i + run < len(row) and run < 128 and row[i + run] == row[i]: run += 1
        if run > 1: out += bytes((257 - run, row[i])); i += run
        else: out += bytes((0, row[i])); i += 1
    return bytes(out)
pl = np.concatenate([np.moveaxis(im, -1, 0), (im[None, :, :, 0] // 2 + 100)]).astype(np.uint8)
rows = [packbits(bytes(pl[c, y])) for c in range(4) for y in range(37)]
open(d + "/za.hdr", "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 37 +X 53\n" + b"".join(bytes((2, 2, 0, 53)) + b"".join(b"".join(bytes((1, int(v))) for v in im[y, :, k]) for k in range(3)) + bytes((128 + 53, 129)) for y in range(37)))
open(d + "/zb.hdr", "wb").write(b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 5 +X 7\n" + np.concatenate([im[:5, :7], np.full((5, 7, 1), 130, np.uint8)], -1).tobytes())
open(d + "/zc.pic", "wb").write(b"\x53\x80\xf6\x34" + bytes(84) + b"PICT" + struct.pack(">HHfHH", 53, 37, 1.0, 3, 0) + bytes((1, 8, 2, 0xE0, 0, 8, 1, 0x10)) + b"".join(b"".join(bytes((26,)) + im[y, x:x + 27].tobytes() for x in (0, 27))[:2 + 53 * 3] + bytes((53, 200)) for y in range(37)))
open(d + "/z.psd", "wb").write(b"8BPS" + struct.pack(">H6xHIIHH", 1, 4, 37, 53, 8, 3) + struct.pack(">III", 0, 0, 0) + struct.pack(">H", 1) + b"".join(struct.pack(">H", len(r)) for r in rows) + b"".join(rows))
PY
export ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096
"$W/fuzz_images" "$IT" "$W/t.bin" "$SEED" "$W"/seeds/* 2>&1 | grep -E "runtime error|ERROR|SUMMARY|decoded$|#[0-9]" | tail -20
CACHE=${CLIP_AMD_FIXTURE_CACHE:-/tmp/clip_amd_fixtures}
PYTHONPATH="$ROOT" python3 -c "from oracle import fixtures as f; [f.cached_model('$CACHE', 'tiny', t, text=tx, vision=True) for t, tx in (('q4_1', False), ('f32', False), ('q4_1', True))]"
"$W/fuzz_model" "$IT" "$W/m.gguf" "$SEED" "$CACHE"/tiny_q4_1_v_s1234.gguf "$CACHE"/tiny_f32_v_s1234.gguf "$CACHE"/tiny_q4_1_tv_s1234.gguf 2>&1 | grep -E "runtime error|ERROR|SUMMARY|loaded$|#[0-9]" | tail -20
