"""CPU tier: the reference's OWN clip.cpp as a checker of the oracle.

oracle/_ref/libclip_ref.so is /root/reference/clip.cpp compiled unchanged (`make -C oracle ref`) on top of oracle/ggml_shim — an eager
stand-in for the slice of the un-vendored ggml submodule that clip.cpp calls, whose op arithmetic is the oracle's restatement.  What runs
here is therefore the reference's source for: the GGUF loader (tensor names, key names, the tensor-count switch), the tokenizer
(std::regex), the preprocessing, the scoring helpers, and the two graph builders op by op (reshape / permute / repeat / acc / get_rows
wiring, scale-after-bias, class + position embedding, causal mask, pooling).  Bit-equality with the oracle's faithful mode pins the oracle's
WIRING to the reference's source; the arithmetic inside the ops stays "unpinned against ggml" (oracle/ggml_shim/ggml/ggml.h).
The library is built in the dev container only (it needs the reference tree) and travels with the snapshot."""
import numpy as np
import pytest

from oracle import fixtures, ref, ref_graph

pytestmark = pytest.mark.skipif(not ref_graph.available(), reason="oracle/_ref/libclip_ref.so not built (needs the reference tree: make -C oracle ref)")

FTYPES = ["f32", "f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"]


@pytest.mark.parametrize("ftype", FTYPES)
def test_reference_graphs_give_the_oracles_bits(fixture_cache, ftype):
    """Two-tower file of the base-model tensor count (397): the reference's loader takes it, both of its graphs reproduce the oracle bit for bit."""
    p = fixtures.cached_model(fixture_cache, "base12", ftype)
    R, O = ref_graph.ReferenceModel(p), ref.OracleModel(p)
    assert R.text_hparams()["n_layer"] == 12 and R.vision_hparams()["patch_size"] == 8
    imgs = fixtures.synthetic_images(3, 32, seed=11)
    for normalize in (True, False):
        got = R.image_batch_encode(imgs, normalize=normalize)
        want = O.image_batch_encode(imgs, normalize=normalize, mode=ref.MODE_FAITHFUL)
        assert np.array_equal(got, want), (ftype, normalize, float(np.abs(got - want).max()))
    assert np.array_equal(R.image_batch_encode(imgs[1:2]), O.image_batch_encode(imgs[1:2], mode=ref.MODE_FAITHFUL))
    for ids in fixtures.synthetic_token_ids(6, seed=3, min_len=1, max_len=75) + [np.array([49406, 49407], np.int32)]:
        for normalize in (True, False):
            got = R.text_encode(ids, normalize=normalize)
            want = O.text_encode(ids, normalize=normalize, mode=ref.MODE_FAITHFUL)
            assert np.array_equal(got, want), (ftype, len(ids), normalize)
    R.close()


@pytest.mark.parametrize("kw,count", [(dict(text=False, vision=True), 200), (dict(text=True, vision=False), 197)])
def test_single_tower_files_and_the_missing_tower(fixture_cache, kw, count):
    p = fixtures.cached_model(fixture_cache, "base12", "q5_1", **kw)
    R, O = ref_graph.ReferenceModel(p), ref.OracleModel(p)
    assert O.info["n_tensors"] == count
    imgs = fixtures.synthetic_images(2, 32, seed=5)
    ids = np.array([49406, 7, 8, 9, 49407], np.int32)
    if kw["vision"]:
        assert np.array_equal(R.image_batch_encode(imgs), O.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL))
        assert R.text_encode(ids) is None                      # "This GGUF file seems to have no text encoder"
    else:
        assert np.array_equal(R.text_encode(ids), O.text_encode(ids, mode=ref.MODE_FAITHFUL))
        assert R.image_batch_encode(imgs) is None
    R.close()


def test_large_model_layer_counts_and_gelu(fixture_cache):
    """24 + 12 layers (589 tensors, the count of ViT-L/14), patch 14; and a use_gelu file (tanh-GELU table instead of quick-GELU)."""
    p = fixtures.cached_model(fixture_cache, "large24", "q8_0")
    R, O = ref_graph.ReferenceModel(p), ref.OracleModel(p)
    assert O.info["n_tensors"] == 589
    imgs = fixtures.synthetic_images(2, 28, seed=6)
    assert np.array_equal(R.image_batch_encode(imgs), O.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL))
    ids = np.array([49406, 17, 4, 900, 49407], np.int32)
    assert np.array_equal(R.text_encode(ids), O.text_encode(ids, mode=ref.MODE_FAITHFUL))
    R.close()
    # 32 + 24 layers (909 tensors: ViT-H/14's count, head size 80), and its text-only form (389)
    for kw, count in ((dict(), 909), (dict(text=True, vision=False), 389)):
        p = fixtures.cached_model(fixture_cache, "huge32", "q4_1", **kw)
        R, O = ref_graph.ReferenceModel(p), ref.OracleModel(p)
        assert O.info["n_tensors"] == count
        if count == 909:
            assert np.array_equal(R.image_batch_encode(imgs), O.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL))
        assert np.array_equal(R.text_encode(ids), O.text_encode(ids, mode=ref.MODE_FAITHFUL))
        R.close()
    p = fixtures.cached_model(fixture_cache, "base12", "f16", use_gelu=True)
    R, O = ref_graph.ReferenceModel(p), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(2, 32, seed=7)
    got, want = R.image_batch_encode(imgs), O.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    assert np.array_equal(got, want)
    quick = ref.OracleModel(fixtures.cached_model(fixture_cache, "base12", "f16")).image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    assert not np.array_equal(want, quick)                       # (the flag does change the network)
    assert np.array_equal(R.text_encode(ids), O.text_encode(ids, mode=ref.MODE_FAITHFUL))
    R.close()


def test_vit_b32_q4_0_at_model_size(fixture_cache):
    """BASELINE's model: the reference's graphs at ViT-B/32 q4_0 size, two images and a 42-token text, against the oracle's bits."""
    p = fixtures.cached_model(fixture_cache, "b32", "q4_0")
    R, O = ref_graph.ReferenceModel(p), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(2, 224, seed=1)
    assert np.array_equal(R.image_batch_encode(imgs), O.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL))
    ids = np.array([49406] + list(range(100, 140)) + [49407], np.int32)
    assert np.array_equal(R.text_encode(ids), O.text_encode(ids, mode=ref.MODE_FAITHFUL))
    R.close()


TEXTS = ["a photo of a cat", "", " ", "  leading  spaces ", "dog's 42!!", "isn't it're've'm'll'd", "tab\there\nnew", "x  ", "1234567 89",
         "snowman ☃ café", "'", "''s", " 's", "!@#$%^&*()", "A B  C   D", "<|startoftext|>a<|endoftext|>", "İstanbul ǅ ß", "don’t", "a\x7fb\x01c"]


def test_the_reference_tokenizer_itself(fixture_cache, clip_lib, monkeypatch):
    """clip_tokenize of the reference (its std::regex, its greedy longest match) against the oracle's restatement AND the product's hand-written
    scanner: same ids, same refusals, on the fixed cases and on 3 000 random strings."""
    monkeypatch.setenv("CLIP_AMD_ALLOW_NO_DEVICE", "1")
    p = fixtures.cached_model(fixture_cache, "base12", "q4_1")
    R, O, P = ref_graph.ReferenceModel(p), ref.OracleModel(p), clip_lib.Clip(p, verbosity=0)

    def product(s):
        try:
            return list(P.tokenize(s))
        except RuntimeError:
            return None

    def oracle(s):
        try:
            return list(O.tokenize(s))
        except Exception:      # noqa: BLE001 — the binding raises on refusal
            return None

    rng = np.random.default_rng(77)
    pieces = list("abcdefghijklmnopqrstuvwxyzABCXYZ0123456789 \t\n\r'!?.,-_/\\\"#$%&()*+:;<=>@[]^`{|}~") + \
        ["'s", "'t", "'re", "'ve", "'m", "'ll", "'d", "'S", "  ", " '", "é", "ß", "☃", "日本", "’", "<|startoftext|>", "<|endoftext|>", "<|", "|>", "İ"]
    cases = list(TEXTS) + ["".join(pieces[int(i)] for i in rng.integers(0, len(pieces), int(rng.integers(0, 40)))) for _ in range(3000)]
    for s in cases:
        want = R.tokenize(s)
        assert oracle(s) == want, repr(s)
        assert product(s) == want, repr(s)
    R.close()


def test_the_reference_preprocessing_and_scoring(fixture_cache, clip_lib, monkeypatch):
    """clip_image_preprocess / clip_similarity_score / softmax_with_sorting / clip_compare_text_and_image / clip_zero_shot_label_image of the
    reference against the oracle (bit for bit) and the product's host preprocessing (bit for bit)."""
    monkeypatch.setenv("CLIP_AMD_ALLOW_NO_DEVICE", "1")
    p = fixtures.cached_model(fixture_cache, "base12", "f16")
    R, O, P = ref_graph.ReferenceModel(p), ref.OracleModel(p), clip_lib.Clip(p, verbosity=0)
    rng = np.random.default_rng(8)
    for ny, nx in ((50, 60), (32, 32), (97, 41), (33, 200), (300, 33), (480, 640), (1, 1), (2, 500)):
        img = rng.integers(0, 256, size=(ny, nx, 3), dtype=np.uint8)
        want = R.preprocess(img)
        assert np.array_equal(O.preprocess(img), want), (ny, nx)
        assert np.array_equal(P.preprocess(img), want), (ny, nx)
    a, b = rng.normal(size=64).astype(np.float32), rng.normal(size=64).astype(np.float32)
    L = ref_graph.lib()
    fp = lambda v: v.ctypes.data_as(ref_graph.C.POINTER(ref_graph.C.c_float))      # noqa: E731
    assert L.clip_similarity_score(fp(a), fp(b), 64) == ref.similarity(a, b)
    arr = rng.normal(size=9).astype(np.float32)
    s_ref, i_ref = np.empty(9, np.float32), np.empty(9, np.int32)
    assert L.softmax_with_sorting(fp(arr.copy()), 9, fp(s_ref), i_ref.ctypes.data_as(ref_graph.C.POINTER(ref_graph.C.c_int)))
    s_orc, i_orc = ref.softmax_with_sorting(arr)
    assert np.array_equal(s_ref, s_orc) and np.array_equal(i_ref, i_orc)
    img = rng.integers(0, 256, size=(45, 70, 3), dtype=np.uint8)
    text = "a photo of a red apple"
    ie = O.image_batch_encode(O.preprocess(img)[None], normalize=True)[0]
    te = O.text_encode(O.tokenize(text), normalize=True)
    assert R.compare_text_and_image(text, img) == ref.similarity(ie, te)
    labels = ["cat", "dog", "red apple", "a photo of a car", "tree"]
    scores, idx = R.zero_shot(img, labels)
    ie_raw = O.image_batch_encode(O.preprocess(img)[None], normalize=False)[0]
    sims = np.array([ref.similarity(ie_raw, O.text_encode(O.tokenize(l), normalize=False)) for l in labels], dtype=np.float32)
    s0, i0 = ref.softmax_with_sorting(sims)
    assert np.array_equal(scores, s0) and np.array_equal(idx, i0)
    R.close()


@pytest.mark.parametrize("src_ftype", ["f32", "f16"])
def test_the_reference_quantize_loop_writes_the_products_file(fixture_cache, clip_lib, tmp_path, src_ftype):
    """clip_model_quantize of the reference (its choice of tensors — 2-D names matching ".*weight" —, its f16 -> f32 conversion, its key overrides,
    its padding; the codecs and the container writer behind it are the shim's) against the product's clip_model_quantize: the same file, byte
    for byte, for every target type; a quantised input is refused by both."""
    import ctypes as C
    L = ref_graph.lib()
    L.clip_model_quantize.restype = C.c_bool
    L.clip_model_quantize.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    src = fixtures.cached_model(fixture_cache, "base12", src_ftype)
    for itype in (2, 3, 6, 7, 8):
        a, b = str(tmp_path / ("ref_%d.gguf" % itype)), str(tmp_path / ("prod_%d.gguf" % itype))
        assert L.clip_model_quantize(src.encode(), a.encode(), itype)
        assert clip_lib.quantize(src, b, itype)
        assert open(a, "rb").read() == open(b, "rb").read(), (src_ftype, itype)
    assert not L.clip_model_quantize(src.encode(), str(tmp_path / "x.gguf").encode(), 5)
    q = fixtures.cached_model(fixture_cache, "base12", "q8_0")
    assert not L.clip_model_quantize(q.encode(), str(tmp_path / "y.gguf").encode(), 2)
    assert not clip_lib.quantize(q, str(tmp_path / "z.gguf"), 2)


def test_reference_programs_on_the_reference_library(fixture_cache, tmp_path):
    """examples/main.cpp, examples/zsl.cpp and tests/benchmark.cpp linked to libclip_ref.so (oracle/_ref/ref_*_cpu): the reference's programs on the
    reference's own clip.cpp, end to end on the CPU — BASELINE config 1 as the reference itself would run it (ggml replaced by the shim).  Their
    printed numbers are the oracle's composition.  (Run here only: the GPU tier compares the same programs linked to libclip.so with the oracle.)"""
    import os
    import re
    import subprocess
    PIL = pytest.importorskip("PIL.Image")
    ref_dir = os.path.dirname(ref_graph.LIB_PATH)
    exes = {n: os.path.join(ref_dir, "ref_%s_cpu" % n) for n in ("main", "zsl", "benchmark")}
    if not all(os.path.exists(e) for e in exes.values()):
        pytest.skip("oracle/_ref/ref_*_cpu not built")
    env = dict(os.environ, OMP_NUM_THREADS=str(ref.host_cores()))
    p = fixtures.cached_model(fixture_cache, "base12", "f16")
    O = ref.OracleModel(p)
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:120, 0:160]
    img = np.clip(np.stack([(np.sin(xx / 9.0) * 0.5 + 0.5) * 255, (np.cos(yy / 7.0) * 0.5 + 0.5) * 255, (xx + 2 * yy) % 256], -1) + rng.normal(0, 6, (120, 160, 3)), 0, 255).astype(np.uint8)
    png = str(tmp_path / "p.png")
    PIL.fromarray(img).save(png)
    text = "a photo of a red apple"
    out = subprocess.run([exes["main"], "-m", p, "--image", png, "--text", text, "-v", "0"], capture_output=True, text=True, timeout=300, env=env)
    m = re.search(r"Similarity score = ([-0-9.]+)", out.stdout)
    assert out.returncode == 0 and m, out.stdout[-800:] + out.stderr[-800:]
    ie = O.image_batch_encode(O.preprocess(img)[None], normalize=True)[0]
    assert abs(float(m.group(1)) - ref.similarity(ie, O.text_encode(O.tokenize(text), normalize=True))) <= 5.1e-4        # three decimals printed
    labels = ["cat", "dog", "a red apple", "tree"]
    cmd = [exes["zsl"], "-m", p, "--image", png, "-v", "0"]
    for l in labels:
        cmd += ["--text", l]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    got = [(g.group(1), float(g.group(2))) for g in re.finditer(r"^(.+) = ([0-9.]+)$", out.stdout, re.M)]
    assert out.returncode == 0 and len(got) == 4, out.stdout[-800:] + out.stderr[-800:]
    ie_raw = O.image_batch_encode(O.preprocess(img)[None], normalize=False)[0]
    sims = np.array([ref.similarity(ie_raw, O.text_encode(O.tokenize(l), normalize=False)) for l in labels], dtype=np.float32)
    s0, i0 = ref.softmax_with_sorting(sims)
    np.testing.assert_allclose([v for _, v in got], s0, atol=5.1e-5)                        # four decimals printed
    root = tmp_path / "tree"
    for ci, c in enumerate(("cat", "dog")):
        (root / c).mkdir(parents=True)
        for k in range(4):                                 # (the program encodes whole batches of 4 per class directory)
            PIL.fromarray(np.roll(img, 17 * (4 * ci + k), axis=1)).save(str(root / c / ("i%d.png" % k)))
    rep = str(tmp_path / "report.txt")
    out = subprocess.run([exes["benchmark"], p, str(root), "0", rep], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-800:] + out.stderr[-800:]
    report = open(rep).read()
    assert "8 images encoded" in report and "2 texts encoded" in report, report
