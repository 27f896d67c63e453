"""CPU tier: the C-ABI library loads, exports the whole clip.h / clip_amd.h surface, and its host-side
pieces (GGUF reader, tokenizer, preprocessing, quantizer, scoring) agree with the oracle.  No GPU compute."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import fixtures, ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return [n for n in names if n not in ("defined",)]


def test_library_exports_every_declared_symbol(clip_lib):
    L = clip_lib.lib()
    declared = set(_declared_functions("clip.h")) | set(_declared_functions("clip_amd.h")) | set(_declared_functions("ggml/ggml.h"))
    assert len(declared) >= 35
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, "libclip.so does not export: %s" % missing
    assert set(clip_lib.API_SYMBOLS) <= declared and set(clip_lib.AMD_SYMBOLS) <= declared
    # the stub libggml.so the reference's ctypes binding dlopens
    g = C.CDLL(os.path.join(os.path.dirname(clip_lib.LIB_PATH), "libggml.so"))
    g.ggml_time_init()
    assert g.ggml_time_us() >= 0


def test_struct_layouts_match_reference_header(clip_lib):
    assert C.sizeof(clip_lib.ClipTextHparams) == 32 and C.sizeof(clip_lib.ClipVisionHparams) == 32
    assert C.sizeof(clip_lib.ClipTokens) == 16
    assert C.sizeof(clip_lib.ClipImageU8) == 24 and C.sizeof(clip_lib.ClipImageF32) == 24


def test_reference_style_c_caller_compiles_and_links(clip_lib, tmp_path):
    """A plain-C translation unit written against include/clip.h (same calls as the reference's examples/simple.c)
    compiles and links against libclip.so unchanged."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "clip.h"
#include <stdio.h>
int main(int argc, char ** argv) {
    ggml_time_init();
    if (argc < 2) { printf("usage\n"); return 0; }
    struct clip_ctx * ctx = clip_model_load(argv[1], 0);
    if (!ctx) { printf("no ctx\n"); return 3; }
    struct clip_vision_hparams * hp = clip_get_vision_hparams(ctx);
    printf("proj=%d t=%lld\n", hp->projection_dim, (long long)ggml_time_us());
    struct clip_tokens tokens;
    if (!clip_tokenize(ctx, "a photo of a cat", &tokens)) return 4;
    printf("ntok=%zu\n", tokens.size);
    float a[4] = {1,2,3,4}, b[4] = {1,1,1,1};
    printf("sim=%g\n", clip_similarity_score(a, b, 4));
    clip_free(ctx);
    return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(clip_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir,
                           "-lclip", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], text=True)
    assert "usage" in out


@pytest.fixture(scope="module")
def tiny_ctx(clip_lib, fixture_cache, host_only_env):
    path = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    return clip_lib.Clip(path, verbosity=0), ref.OracleModel(path)


def test_model_load_reads_hparams(tiny_ctx):
    c, o = tiny_ctx
    v, t = c.vision_config, c.text_config
    assert (v["image_size"], v["patch_size"], v["hidden_size"], v["n_intermediate"], v["projection_dim"], v["n_head"], v["n_layer"]) == (32, 8, 64, 128, 32, 2, 2)
    assert (t["n_vocab"], t["num_positions"], t["hidden_size"], t["n_head"], t["n_layer"]) == (49408, 77, 64, 2, 2)
    assert abs(v["eps"] - 1e-5) < 1e-12


def test_model_load_failures_return_null_not_throw(clip_lib, tmp_path, host_only_env):
    L = clip_lib.lib()
    assert not L.clip_model_load(b"/nonexistent/file.gguf", 0)
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"GGUF" + b"\x02\x00\x00\x00" + b"\xff" * 40)
    assert not L.clip_model_load(os.fsencode(str(bad)), 0)
    junk = tmp_path / "junk.gguf"
    junk.write_bytes(b"not a gguf file at all, sorry........")
    assert not L.clip_model_load(os.fsencode(str(junk)), 0)
    # truncated real file
    src = fixtures.cached_model("/tmp/clip_amd_fixtures", "tiny", "q4_0")
    data = open(src, "rb").read()
    tr = tmp_path / "trunc.gguf"
    tr.write_bytes(data[: len(data) // 2])
    assert not L.clip_model_load(os.fsencode(str(tr)), 0)


def test_no_device_means_no_context_unless_opted_in(clip_lib, fixture_cache):
    """The product must fail loudly without a HIP device: no silent CPU fallback."""
    if clip_lib.device_count() > 0:
        pytest.skip("GPU present")
    path = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    env = dict(os.environ)
    env.pop("CLIP_AMD_ALLOW_NO_DEVICE", None)
    code = "import clip_cpp_amd as c; L=c.lib(); import sys; sys.exit(0 if not L.clip_model_load(%r.encode(),0) else 1)" % path
    r = subprocess.run(["python", "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "no HIP device" in r.stderr


def test_host_only_context_refuses_to_encode(tiny_ctx, clip_lib):
    if clip_lib.device_count() > 0:
        pytest.skip("GPU present")
    c, o = tiny_ctx
    assert c.device == -1
    with pytest.raises(RuntimeError):
        c.encode_images(np.zeros((1, 32, 32, 3), dtype=np.float32))
    with pytest.raises(RuntimeError):
        c.encode_text([49406, 5, 49407])
    with pytest.raises(RuntimeError):      # the GPU preprocessing path has no host fallback either
        c.encode_images_u8([np.zeros((40, 50, 3), dtype=np.uint8)])


TEXTS = ["a photo of a cat", "", " ", "  leading  spaces ", "dog's 42!!", "isn't it're've'm'll'd", "tab\there\nnew", "x  ", "  ",
         "1234567 89", "snowman ☃ café", "'", "''s", " 's", "a\t\tb", "hello   world  ", "!@#$%^&*()", "A B  C   D"]


def test_tokenizer_bit_exact_vs_oracle(tiny_ctx):
    c, o = tiny_ctx
    for t in TEXTS:
        assert c.tokenize(t) == list(o.tokenize(t)), repr(t)


def test_tokenizer_fuzz_vs_std_regex(tiny_ctx):
    """Hand-written scanner == std::regex restatement on random byte soup (bit-exact ids)."""
    c, o = tiny_ctx
    rng = np.random.default_rng(123)
    alphabet = list("abetz AB09 '!?.,\t\n-") + ["'s", "'re", "  ", " 'll", "é"]
    for _ in range(400):
        n = int(rng.integers(0, 24))
        s = "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n))
        assert c.tokenize(s) == list(o.tokenize(s)), repr(s)


def test_preprocess_bit_exact_vs_oracle(tiny_ctx):
    c, o = tiny_ctx
    rng = np.random.default_rng(9)
    for (ny, nx) in ((50, 60), (32, 32), (97, 41), (33, 200), (480, 640)):
        img = rng.integers(0, 256, size=(ny, nx, 3), dtype=np.uint8)
        a = c.preprocess(img)
        b = o.preprocess(img)
        assert np.array_equal(a, b), (ny, nx)


def test_batch_preprocess_threads(tiny_ctx, clip_lib):
    c, o = tiny_ctx
    L = clip_lib.lib()
    rng = np.random.default_rng(10)
    imgs = [rng.integers(0, 256, size=(40 + 3 * i, 50 + i, 3), dtype=np.uint8) for i in range(5)]
    arr = (clip_lib.ClipImageU8 * 5)()
    for i, im in enumerate(imgs):
        arr[i] = clip_lib.ClipImageU8(im.shape[1], im.shape[0], im.ctypes.data_as(C.POINTER(C.c_uint8)), im.size)
    inb = clip_lib.ClipImageU8Batch(C.cast(arr, C.POINTER(clip_lib.ClipImageU8)), 5)
    outarr = (clip_lib.ClipImageF32 * 5)()
    outb = clip_lib.ClipImageF32Batch(C.cast(outarr, C.POINTER(clip_lib.ClipImageF32)), 0)
    L.clip_image_batch_preprocess(c.ctx, 3, C.byref(inb), C.byref(outb))
    assert outb.size == 5
    for i, im in enumerate(imgs):
        got = np.ctypeslib.as_array(outarr[i].data, shape=(32, 32, 3))
        assert np.array_equal(got, o.preprocess(im))
        L.clip_image_f32_clean(C.byref(outarr[i]))


def test_scoring_helpers_match_reference_semantics(clip_lib):
    L = clip_lib.lib()
    rng = np.random.default_rng(2)
    a = rng.standard_normal(512).astype(np.float32)
    b = rng.standard_normal(512).astype(np.float32)
    got = L.clip_similarity_score(a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)), 512)
    assert got == ref.similarity(a, b)   # same sequential f32 accumulation -> bit-exact
    x = rng.standard_normal(37).astype(np.float32)
    x[5] = x[9]  # tie
    s0, i0 = ref.softmax_with_sorting(x)
    arr = x.copy()
    s1 = np.empty(37, dtype=np.float32)
    i1 = np.empty(37, dtype=np.int32)
    assert L.softmax_with_sorting(arr.ctypes.data_as(C.POINTER(C.c_float)), 37, s1.ctypes.data_as(C.POINTER(C.c_float)),
                                  i1.ctypes.data_as(C.POINTER(C.c_int)))
    assert np.array_equal(s0, s1) and np.array_equal(i0, i1)
    assert abs(s1.sum() - 1.0) < 1e-5 and np.all(np.diff(s1) <= 0)


@pytest.mark.parametrize("ftype", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_quantizer_bit_exact_vs_oracle_codecs(clip_lib, tmp_path, fixture_cache, ftype):
    """clip_model_quantize (product codecs) re-emits the f32 GGUF with bit-identical blocks to the oracle's restatement
    of quantize_row_q*_reference, and only 2-D '*weight' tensors are touched (clip.cpp:1711-1739)."""
    src = fixtures.cached_model(fixture_cache, "tiny", "f32", text=False, vision=True)
    dst = str(tmp_path / ("q_%s.gguf" % ftype))
    assert clip_lib.quantize(src, dst, ref.GGML_TYPES[ftype])
    expect = fixtures.cached_model(fixture_cache, "tiny", ftype, text=False, vision=True)
    a, b = ref.OracleModel(dst), ref.OracleModel(expect)
    assert a.info == b.info and a.info["ftype"] == ref.GGML_TYPES[ftype]
    imgs = fixtures.synthetic_images(2, 32)
    assert np.array_equal(a.image_batch_encode(imgs), b.image_batch_encode(imgs))
    assert not clip_lib.quantize(src, dst, 5)  # invalid itype


def test_reference_quantize_program_runs_unchanged(clip_lib, tmp_path, fixture_cache):
    """models/quantize.cpp compiled unchanged (oracle/_ref/ref_quantize, `make -C oracle ref`): the command-line tool a user of the
    reference runs to produce q4_0 ... q8_0 files, here over this library's clip_model_quantize; the file it writes is the file the
    binding's call writes, byte for byte, and bad arguments print the usage and fail."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_quantize")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_quantize not built (needs the reference tree)")
    src = fixtures.cached_model(fixture_cache, "tiny", "f32")
    for itype in (2, 7):
        out = str(tmp_path / ("prog_%d.gguf" % itype))
        r = subprocess.run([exe, src, out, str(itype)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "quantize time" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]
        same = str(tmp_path / ("call_%d.gguf" % itype))
        assert clip_lib.quantize(src, same, itype)
        assert open(out, "rb").read() == open(same, "rb").read()
    r = subprocess.run([exe, src, str(tmp_path / "x.gguf"), "5"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage" in r.stderr
    r = subprocess.run([exe, str(tmp_path / "missing.gguf"), str(tmp_path / "y.gguf"), "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "failed to quantize" in r.stderr


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree only exists in the dev container")
@pytest.mark.parametrize("prog", ["examples/main.cpp", "examples/zsl.cpp", "examples/extract.cpp", "examples/simple.c",
                                  "tests/benchmark.cpp", "models/quantize.cpp"])
def test_reference_programs_build_unchanged_against_this_library(clip_lib, tmp_path, prog):
    """Drop-in acceptance: the reference's own callers compile and link against include/ + libclip.so with no edits."""
    libdir = os.path.dirname(clip_lib.LIB_PATH)
    src = os.path.join(REFERENCE, prog)
    cmd = (["gcc", "-std=c11"] if prog.endswith(".c") else ["g++", "-std=c++17"]) + ["-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(REFERENCE, "examples"), src]
    if prog.endswith(".cpp") and "quantize" not in prog:
        cmd.append(os.path.join(REFERENCE, "examples", "common-clip.cpp"))
    cmd += ["-L", libdir, "-lclip", "-Wl,-rpath," + libdir, "-o", str(tmp_path / "prog")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree only exists in the dev container")
def test_reference_ctypes_binding_imports_and_drives_this_library(clip_lib, tmp_path, fixture_cache, monkeypatch):
    """The reference's own Python package (examples/python_bindings/clip_cpp, SURVEY 8b "Callers") with libclip.so + the stub
    libggml.so placed in its directory as INTEGRATION.md section 3 says: the module imports (every prototype it declares resolves),
    its Clip class loads a GGUF and its host-side methods agree with this repo's binding.  The package is linked, not copied; the
    encoders need a GPU and are covered by the compiled reference programs in the GPU tier."""
    pkg = tmp_path / "clip_cpp"
    pkg.mkdir()
    src = os.path.join(REFERENCE, "examples", "python_bindings", "clip_cpp")
    for f in os.listdir(src):
        if f.endswith(".py"):
            os.symlink(os.path.join(src, f), str(pkg / f))
    libdir = os.path.dirname(clip_lib.LIB_PATH)
    for so in ("libclip.so", "libggml.so"):
        os.symlink(os.path.join(libdir, so), str(pkg / so))
    model = fixtures.cached_model(fixture_cache, "tiny", "q4_1")
    jpg = os.path.join(REFERENCE, "tests", "red_apple.jpg")
    code = (
        "import json, ctypes, clip_cpp.clip as c\n"
        "m = c.Clip(%r, verbosity=0)\n"
        "u8 = c.make_clip_image_u8(); ok = bool(c.clip_image_load_from_file(%r.encode(), u8)) if %r else None\n"
        "f32 = c.make_clip_image_f32(); pre = bool(c.clip_image_preprocess(m.ctx, u8, f32)) if ok else None\n"
        "print(json.dumps({'text': m.text_config, 'vision': m.vision_config, 'tok': m.tokenize('a photo of a red apple'),\n"
        "                  'sim': m.calculate_similarity([1.0] + [0.0] * (m.vec_dim - 1), [0.6, 0.8] + [0.0] * (m.vec_dim - 2)),\n"
        "                  'loaded': ok, 'pre': pre, 'nx': f32.contents.nx if pre else None}))\n") % (model, jpg, os.path.exists(jpg))
    env = dict(os.environ, PYTHONPATH=str(tmp_path), CLIP_AMD_ALLOW_NO_DEVICE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    monkeypatch.setenv("CLIP_AMD_ALLOW_NO_DEVICE", "1")
    mine = clip_lib.Clip(model, verbosity=0)
    for k, v in got["text"].items():
        assert mine.text_config[k] == pytest.approx(v)
    for k, v in got["vision"].items():
        assert mine.vision_config[k] == pytest.approx(v)
    assert got["tok"] == list(mine.tokenize("a photo of a red apple"))
    assert got["sim"] == pytest.approx(0.6, abs=1e-6)
    if got["loaded"] is not None:
        assert got["loaded"] and got["pre"] and got["nx"] == mine.vision_config["image_size"]


def test_product_side_synthetic_models_load_everywhere(tmp_path, clip_lib):
    """clip_cpp_amd.synth (what bench.py uses: product GGUF writer + clip_model_quantize, no oracle/ involved) writes files
    with the reference's tensor inventory that both libclip.so and the oracle accept."""
    from clip_cpp_amd import synth
    env_had = os.environ.get("CLIP_AMD_ALLOW_NO_DEVICE")
    os.environ["CLIP_AMD_ALLOW_NO_DEVICE"] = "1"
    try:
        for ftype, text, vision, n in [("q4_0", True, True, 77), ("f16", False, True, 40), ("q8_0", True, False, 37)]:
            p = synth.cached_model(str(tmp_path), "tiny", ftype, text=text, vision=vision)
            kv, tensors = clip_lib.gguf_inspect(p)
            assert len(tensors) == n and kv["general.file_type"] == synth.FTYPES[ftype]
            o = ref.OracleModel(p)
            assert o.info["has_text"] == int(text) and o.info["has_vision"] == int(vision) and o.info["ftype"] == synth.FTYPES[ftype]
            c = clip_lib.Clip(p)
            if text:
                assert c.tokenize("a b")[0] == 49406 and len(kv["tokenizer.ggml.tokens"]) == 49408
        # the tensor inventory of the full-size architectures matches the counts the reference switches on (clip.cpp:267-289)
        assert len(synth.tensor_list(synth.ARCH["b32"])) == 397 and len(synth.tensor_list(synth.ARCH["l14"])) == 589
        assert len(synth.tensor_list(synth.ARCH["h14"])) == 909 and len(synth.tensor_list(synth.ARCH["b32"], text=False)) == 200
    finally:
        if env_had is None:
            os.environ.pop("CLIP_AMD_ALLOW_NO_DEVICE", None)


def test_repacked_weight_cache_keyed_by_content(clip_lib, tmp_path, fixture_cache, host_only_env, monkeypatch):
    """CLIP_AMD_WEIGHT_CACHE=<dir> keeps the HBM image of a model; a later load of the same content uses it, a truncated
    or foreign file is ignored and rewritten, another file content gets another key.  (Host-only contexts: layout and file handling;
    the GPU tier checks that a cached load encodes bit-identically.)"""
    import glob
    import shutil
    cache = tmp_path / "wcache"
    cache.mkdir()
    monkeypatch.setenv("CLIP_AMD_WEIGHT_CACHE", str(cache))
    src = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    a = tmp_path / "model_a.gguf"
    shutil.copy(src, a)
    c = clip_lib.Clip(str(a), verbosity=0)
    assert not c.weights_from_cache
    files = glob.glob(str(cache / "model_a.gguf.*.hbm"))
    assert len(files) == 1 and os.path.getsize(files[0]) > 32
    c.close()
    c = clip_lib.Clip(str(a), verbosity=0)
    assert c.weights_from_cache and c.tokenize("a b")[0] == 49406       # everything outside the weight image is still read from the GGUF
    c.close()
    # truncated cache file: ignored, rebuilt, usable again
    size = os.path.getsize(files[0])
    with open(files[0], "r+b") as f:
        f.truncate(size // 2)
    c = clip_lib.Clip(str(a), verbosity=0)
    assert not c.weights_from_cache
    c.close()
    assert os.path.getsize(files[0]) == size
    # foreign bytes under the right name (wrong magic): ignored
    with open(files[0], "r+b") as f:
        f.write(b"NOTACACHE")
    c = clip_lib.Clip(str(a), verbosity=0)
    assert not c.weights_from_cache
    c.close()
    # other content (another quantisation of the same architecture), same file name in another directory -> another key
    other = tmp_path / "sub"
    other.mkdir()
    shutil.copy(fixtures.cached_model(fixture_cache, "tiny", "q8_0"), other / "model_a.gguf")
    c = clip_lib.Clip(str(other / "model_a.gguf"), verbosity=0)
    assert not c.weights_from_cache
    c.close()
    assert len(glob.glob(str(cache / "model_a.gguf.*.hbm"))) == 2
    # same name, same size, same metadata, same first / last 256 KB — only a few bytes in the MIDDLE of an inner tensor differ
    # (a fine-tune or re-quantisation of inner layers): every tensor is sampled, so this is another key, never the stale image
    blob = bytearray(open(a, "rb").read())
    lo, hi = int(len(blob) * 0.55) & ~4095, int(len(blob) * 0.75) & ~4095     # deep inside the tensor data (the metadata ends in the first quarter)
    for off in range(lo, hi, 2048):                   # one byte per 2 KB over a fifth of the file: crosses sampled 4 KB blocks of the token table
        blob[off] ^= 0x5A
    third = tmp_path / "sub3"
    third.mkdir()
    (third / "model_a.gguf").write_bytes(bytes(blob))
    c = clip_lib.Clip(str(third / "model_a.gguf"), verbosity=0)
    assert not c.weights_from_cache
    c.close()
    assert len(glob.glob(str(cache / "model_a.gguf.*.hbm"))) == 3
    monkeypatch.setenv("CLIP_AMD_WEIGHT_CACHE_FULLHASH", "1")      # every byte hashed: a single flipped byte anywhere is another key
    blob2 = bytearray(open(a, "rb").read())
    blob2[int(len(blob2) * 0.6)] ^= 1
    fourth = tmp_path / "sub4"
    fourth.mkdir()
    (fourth / "model_a.gguf").write_bytes(bytes(blob2))
    for path in (a, fourth / "model_a.gguf"):
        c = clip_lib.Clip(str(path), verbosity=0)
        assert not c.weights_from_cache                # (the full-hash key of `a` differs from its sampled key too)
        c.close()
    assert len(glob.glob(str(cache / "model_a.gguf.*.hbm"))) == 5
    monkeypatch.delenv("CLIP_AMD_WEIGHT_CACHE_FULLHASH")
    # without the variable nothing is read or written
    monkeypatch.delenv("CLIP_AMD_WEIGHT_CACHE")
    c = clip_lib.Clip(str(a), verbosity=0)
    assert not c.weights_from_cache
    c.close()


def test_malformed_and_unsupported_files_are_rejected_at_load(clip_lib, tmp_path, host_only_env, capfd):
    """(a) tensor shapes whose element count wraps int64 must not pass the bounds check; (b) models outside the
    kernels' limits (hidden_size > 2048, projection_dim % 4, unsupported head size, too many tokens) fail at LOAD with a message,
    not at the first encode; (c) nothing throws across the C ABI (quantising such a file returns false)."""
    import struct
    L = clip_lib.lib()

    def s(b):
        return struct.pack("<Q", len(b)) + b

    # (a) one tensor with dims 2^32 x 2^32 x 2^32 x 2^32 (product wraps to 0), offset 0
    kv = s(b"general.alignment") + struct.pack("<II", 4, 32)
    ti = s(b"t.weight") + struct.pack("<I", 4) + struct.pack("<QQQQ", 1 << 32, 1 << 32, 1 << 32, 1 << 32) + struct.pack("<IQ", 0, 0)
    blob = b"GGUF" + struct.pack("<IQQ", 2, 1, 1) + kv + ti + b"\0" * 256
    f = tmp_path / "wrap.gguf"
    f.write_bytes(blob)
    assert not L.clip_model_load(os.fsencode(str(f)), 0)
    assert not L.clip_model_quantize(os.fsencode(str(f)), os.fsencode(str(tmp_path / "o.gguf")), 2)
    # offset + size wrap: huge offset
    ti2 = s(b"t.weight") + struct.pack("<I", 1) + struct.pack("<Q", 64) + struct.pack("<IQ", 0, (1 << 64) - 64)
    f2 = tmp_path / "wrap2.gguf"
    f2.write_bytes(b"GGUF" + struct.pack("<IQQ", 2, 1, 1) + kv + ti2 + b"\0" * 512)
    assert not L.clip_model_load(os.fsencode(str(f2)), 0)
    # string array whose count exceeds the remaining bytes
    kv3 = s(b"tokenizer.ggml.tokens") + struct.pack("<IIQ", 9, 8, 1 << 25)
    f3 = tmp_path / "arr.gguf"
    f3.write_bytes(b"GGUF" + struct.pack("<IQQ", 2, 0, 1) + kv3 + b"\0" * 64)
    assert not L.clip_model_load(os.fsencode(str(f3)), 0)
    capfd.readouterr()

    # (b) kernel limits
    base = fixtures.CONFIGS["tiny"]
    cases = {
        "hidden_size 4096 > 2048": dict(v=dict(base["v"], h=4096, nh=64, ff=64), t=base["t"]),
        "projection_dim 30": dict(v=dict(base["v"], proj=30), t=dict(base["t"], proj=30)),
        "head size 16": dict(v=dict(base["v"], nh=4), t=base["t"]),
        "tokens per sequence": dict(v=dict(base["v"], S=336, P=8, h=192, nh=2, ff=64), t=base["t"]),   # 42*42+1 = 1765 tokens, d_head 96
    }
    for needle, cfg in cases.items():
        path = str(tmp_path / ("lim_%d.gguf" % len(needle)))
        fixtures.make_model(path, cfg, "f32", text="projection" in needle, vision=True)
        assert not L.clip_model_load(os.fsencode(path), 0), needle
        err = capfd.readouterr().err
        assert needle in err, (needle, err)


def test_gemm_tile_heuristic_covers_the_model_shapes(clip_lib):
    """launch_gemm's tile choice is host arithmetic (clip_amd_test_gemm_tile): every shape of the target matrix maps to a tile code that exists,
    the regimes land where the measured sweeps put them (profiles/r02_ring_sweep_*.txt, r02_gemm8_experiments.txt), and the choice is monotone
    in the obvious sense (no ring tiles for one image's rows or for batch 256)."""
    L = clip_lib.lib()
    tile = lambda M, N, K, q=1: L.clip_amd_test_gemm_tile(M, N, K, q)
    known = {64064, 64128, 128064, 128128, 160128, 192128, 65064, 65128, 160256, 256260, 256261, 320261}
    models = {"b32": (50, 768, 3072), "b32t": (40, 512, 2048), "l14": (257, 1024, 4096), "l14t": (40, 768, 3072), "h14": (257, 1280, 5120)}
    for name, (T, h, ff) in models.items():
        for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024):
            for N, K in ((3 * h, h), (h, h), (ff, h), (h, ff)):
                for q in (0, 1):
                    t = tile(B * T, N, K, q)
                    assert t in known, (name, B, N, K, q, t)
                    if q == 1:
                        assert t % 1000 != 261, (name, B, N, K, t)       # the 32 x 32 x 16 kernel multiplies fp16 weights only (f16 file or resident panel)
                    assert L.clip_amd_test_gemm_tile_ex(B * T, N, K, q, 1) % 1000 != 261       # ... and never on a shared device
    # round 6: q/k/v and FFN-up of a ViT-B/32-class batch alone on the device -> k_gemm32.hip, on the tile that fills its last round; not the narrow-round shapes
    assert tile(12800, 2304, 768, 0) == 256261 and tile(12800, 3072, 768, 0) == 320261 and tile(10290, 1536, 512, 0) == 256261
    assert tile(10290, 2048, 512, 0) % 1000 != 261 and tile(12800, 768, 768, 0) % 1000 != 261 and tile(12800, 768, 3072, 0) % 1000 != 261
    assert tile(1600, 2304, 768, 0) % 1000 != 261 and tile(65792, 3072, 1024, 0) == 320261 and L.clip_amd_test_gemm_tile_ex(65792, 3072, 1024, 0, 1) == 256260
    for (M, N, K) in ((12800, 2304, 768), (12800, 3072, 768), (10290, 1536, 512)):
        assert L.clip_amd_test_gemm_tile_ex(M, N, K, 0, 1) == tile(M, N, K, 1) or L.clip_amd_test_gemm_tile_ex(M, N, K, 0, 1) // 1000 in (128, 160, 192)
    # <= 64 rows: the two-buffer 64 x 64 tile (the layers themselves run on k_skinny.hip there)
    assert tile(50, 768, 768) == 64064 and tile(13, 512, 2048) == 64064
    # mid-M: the ring kernel where the sweep has it ahead ...
    assert tile(1600, 2304, 768) == 65128 and tile(1600, 768, 768) == 65128 and tile(1600, 768, 3072) == 65128      # ViT-B/32 batch 32: q/k/v, out, FFN down
    assert tile(2560, 1536, 512) == 65128                                                                            # (K = 512: the ring stays ahead at 480 tiles)
    assert tile(1600, 3072, 768) == 160128                                                                           # ... its FFN up stays on the big tile
    assert tile(257, 3072, 1024, 0) == 65064 and tile(257, 4096, 1024, 0) == 65128 and tile(257, 1024, 4096, 0) == 65064   # one ViT-L/14 image, f16
    assert tile(1280, 512, 512) == 65064 and tile(1280, 1536, 512) == 65128                                          # a batch of 32 texts
    assert tile(514, 768, 3072, 1) == 64064 or tile(514, 768, 3072, 1) // 1000 == 64                                 # quantised long K below ~1000 rows: split-K form
    assert tile(1568, 768, 3072, 0) == 65064                                                                         # patch embedding of 32 images (f16 kernel)
    # 4500-8000 rows (r03 sweep): 128 x 128 unsplit, not 64 x 128 + split-K
    assert tile(5000, 512, 2048) == 128128 and tile(4500, 768, 3072) == 128128 and tile(6000, 768, 3072) == 160128 and tile(8000, 768, 3072) == 192128
    # batch 256 and beyond: never the ring
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        assert tile(12800, N, K) // 1000 in (128, 160, 192), (N, K, tile(12800, N, K))
    assert tile(65792, 4096, 1024, 0) == 320261 and L.clip_amd_test_gemm_tile_ex(65792, 4096, 1024, 0, 1) == 256260 and tile(65792, 1024, 4096, 0) == 256260                             # ViT-L/14 batch 256: four-wave 256 x 256
