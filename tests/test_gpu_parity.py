"""GPU tier, end to end: embeddings from the HIP path vs the CPU oracle (ggml numerics) on the same
synthetic GGUF weights and the same seeded inputs, through the drop-in C ABI.

Stated tolerance (cosine-similarity delta, per embedding):
    1 - cos(gpu, oracle_faithful) <= 1e-3   for q4_0/q4_1/q5_0/q5_1/q8_0 files
    1 - cos(gpu, oracle_faithful) <= 1e-4   for f16 / f32 files
The dominant term is the ORACLE's own activation quantisation (ggml rounds activations to q8_0/q8_1
before every quantised mat-mul; the GPU keeps them in fp16), see DESIGN.md "Numerics".
Token ids and patch indexing must be bit-exact.
"""
import os

import numpy as np
import pytest

from oracle import fixtures, ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

# f32 files (round 5): the weights stay f32 in HBM and are multiplied on the exact-f32 MFMA (k_gemm_f32.hip) — 1e-6, the resolution of a
# float32 cosine; what is left is the fp16 rounding of the activations between the kernels (the reference keeps them f32 for f32 weights)
TOL = {"f32": 1e-6, "f16": 1e-4, "q4_0": 1e-3, "q4_1": 1e-3, "q5_0": 1e-3, "q5_1": 1e-3, "q8_0": 1e-3}
# What the tests at MODEL shape (ViT-B/32, L/14, H/14 widths) assert: the observed maxima are 1.0e-4 (images) / 1.9e-4 (texts) for q4_0
# and below 1e-5 for f16 files, so a regression that triples the error must fail; TOL stays the documented contract
# and is what the 32-128-wide test towers are held to (one rounding flip is a larger share of their embeddings).
TOL_MODEL = {"f32": 1e-6, "f16": 3e-5, "q4_0": 3e-4, "q4_1": 3e-4, "q5_0": 3e-4, "q5_1": 3e-4, "q8_0": 3e-4}
# text towers: the one-token text (a bare BOS, a single row through 12 layers) sits at 4.4e-4 against the oracle's 8-bit activations
# (ViT-B/32 q8_0, r03d), every longer text below 1.7e-4
TOL_MODEL_TEXT = {k: (6e-4 if k.startswith("q") else v) for k, v in TOL_MODEL.items()}


@pytest.fixture(scope="module")
def gpu(clip_lib):
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device: the product has no CPU fallback")
    return clip_lib


def one_minus_cos(a, b):
    a = a / np.linalg.norm(a, axis=-1, keepdims=True)
    b = b / np.linalg.norm(b, axis=-1, keepdims=True)
    return 1.0 - (a * b).sum(-1)


@pytest.mark.parametrize("ftype", ["f32", "f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
@pytest.mark.parametrize("config", ["tiny", "tiny14"])
def test_two_tower_parity_small_models(gpu, fixture_cache, config, ftype):
    path = fixtures.cached_model(fixture_cache, config, ftype)
    clip = gpu.Clip(path, device=0)
    orc = ref.OracleModel(path)
    S = clip.vision_config["image_size"]
    imgs = fixtures.synthetic_images(5, S, seed=21)
    for normalize in (True, False):
        got = clip.encode_images(imgs, normalize=normalize)
        want = orc.image_batch_encode(imgs, normalize=normalize, mode=ref.MODE_FAITHFUL)
        d = one_minus_cos(got, want)
        assert np.all(d <= TOL[ftype]), (config, ftype, d)
        if normalize:
            np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
        else:
            np.testing.assert_allclose(np.linalg.norm(got, axis=1), np.linalg.norm(want, axis=1), rtol=2e-2)
    texts = fixtures.synthetic_token_ids(6, seed=5, min_len=1, max_len=30)
    batch = clip.encode_texts(texts, normalize=True)
    for i, ids in enumerate(texts):
        single = np.asarray(clip.encode_text(list(ids), normalize=True), dtype=np.float32)
        want = orc.text_encode(ids, normalize=True, mode=ref.MODE_FAITHFUL)
        assert one_minus_cos(single, want) <= TOL[ftype], (config, ftype, i)
        # ragged batch == one-at-a-time, up to the fp32 re-association between the batch's tiled GEMMs and the single text's
        # small-M kernels (k_skinny.hip: intra-workgroup split-K, LayerNorm statistics from partial sums)
        # ... and, since round 3, the rounding placement of the LayerNorm fold (batches of > 64 rows multiply fp16(x gamma) and finish
        # the normalisation in the GEMM epilogue; the small-M kernels round the normalised value): a few 1e-4 on these 32-64-wide towers
        assert one_minus_cos(batch[i], single) <= 1e-6, (config, ftype, i)
        np.testing.assert_allclose(batch[i], single, atol=1e-3)
    clip.close()


def _ragged_text_batch(n, npos, seed):
    """n ragged texts (BOS + ids + EOS) whose token counts cover 1 ... npos: a bare BOS (1 token), BOS+EOS (2), the longest
    legal sequence (npos), duplicates of one length, the rest seeded uniform."""
    rng = np.random.default_rng(seed)
    lens = [1, 2, npos, npos, 9, 9, 9, npos - 1, 3] + [int(v) for v in rng.integers(2, npos + 1, size=n - 9)]
    out = []
    for ln in lens:
        if ln == 1:
            out.append(np.array([49406], np.int32))
            continue
        ids = rng.integers(0, fixtures.N_VOCAB - 2, size=ln - 2).astype(np.int32)
        out.append(np.concatenate([[49406], ids, [49407]]).astype(np.int32))
    return out


@pytest.mark.parametrize("config,ftype,n_texts", [("b32", "q4_0", 72), ("l14", "f16", 64), ("b32", "q8_0", 16), ("l14", "q5_1", 12)])
def test_text_tower_parity_at_model_shape(gpu, fixture_cache, config, ftype, n_texts):
    """Row a13: clip_text_encode at the REAL text-tower shapes (ViT-B/32: h=512, ff=2048, 8 heads; ViT-L/14: h=768, ff=3072,
    12 heads; 12 layers, 77 positions, q4_0 / f16 token_embd) against the oracle in ggml-faithful numerics, reference
    clip.cpp:1016-1233: token/position gather :1059-1061, causal mask :1101, last-row pooling :1154-1155.  One ragged batch
    of >= 64 texts with 1 ... 77 tokens (incl. a 77-token text and duplicate lengths); every text is also encoded alone."""
    p = fixtures.cached_model(fixture_cache, config, ftype, text=True, vision=False)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    npos = clip.text_config["num_positions"]
    assert npos == 77
    texts = _ragged_text_batch(n_texts, npos, seed=1000 + n_texts)
    assert max(len(t) for t in texts) == 77 and min(len(t) for t in texts) == 1
    for normalize in (True, False):
        batch = clip.encode_texts(texts, normalize=normalize)
        assert batch.shape == (n_texts, clip.text_config["projection_dim"]) and np.all(np.isfinite(batch))
        want = np.stack([orc.text_encode(ids, normalize=normalize, mode=ref.MODE_FAITHFUL) for ids in texts])
        d = one_minus_cos(batch, want)
        assert np.all(d <= TOL_MODEL_TEXT[ftype]), (config, ftype, normalize, float(d.max()), int(d.argmax()), len(texts[int(d.argmax())]))
        if not normalize:
            np.testing.assert_allclose(np.linalg.norm(batch, axis=1), np.linalg.norm(want, axis=1), rtol=2e-2)
    batch = clip.encode_texts(texts, normalize=True)
    assert np.array_equal(batch, clip.encode_texts(texts, normalize=True))                      # deterministic
    for i in list(range(9)) + [n_texts - 1]:
        single = np.asarray(clip.encode_text(list(texts[i]), normalize=True), dtype=np.float32)
        assert one_minus_cos(single, want_n(orc, texts[i])) <= TOL_MODEL_TEXT[ftype], i
        # batch row == the text alone, up to the fp32 re-association of a different GEMM schedule (split-K at small M)
        assert one_minus_cos(single, batch[i]) <= 1e-6, (i, len(texts[i]))
        np.testing.assert_allclose(batch[i], single, atol=3e-4)
    # a different batch composition around the same texts gives the same rows (no cross-text leakage through the ragged layout)
    perm = np.random.default_rng(3).permutation(n_texts)
    shuffled = clip.encode_texts([texts[j] for j in perm], normalize=True)
    assert np.all(one_minus_cos(shuffled, batch[perm]) <= 1e-6)
    np.testing.assert_allclose(shuffled, batch[perm], atol=3e-4)
    clip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_texts", [65, 300, 4200])
def test_text_batch_position_lookup_with_many_sequences(gpu, fixture_cache, n_texts):
    """The embedding kernel finds a token row's sequence with a 64-ary search over the sequence starts (1 step up to 64 texts, 2 up to
    4096, 3 beyond): a ragged batch of many short texts must give, row for row, what the same texts give in chunks of 50 (one-step
    searches) — a wrong position index changes an embedding at the 1e-2 level, the GEMM schedules differ at 1e-7."""
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    clip = gpu.Clip(p, device=0)
    npos = clip.text_config["num_positions"]
    rng = np.random.default_rng(7000 + n_texts)
    nv = clip.text_config["n_vocab"] if "n_vocab" in clip.text_config else 64
    texts = [np.concatenate([[nv - 2], rng.integers(0, nv - 2, size=int(rng.integers(0, min(8, npos - 2) + 1))), [nv - 1]]).astype(np.int32)
             for _ in range(n_texts)]
    batch = clip.encode_texts(texts, normalize=True)
    assert batch.shape[0] == n_texts and np.all(np.isfinite(batch))
    chunks = np.concatenate([clip.encode_texts(texts[i:i + 50], normalize=True) for i in range(0, n_texts, 50)])
    d = one_minus_cos(batch, chunks)
    assert np.all(d <= 1e-6), (float(d.max()), int(d.argmax()))
    np.testing.assert_allclose(batch, chunks, atol=1e-3)
    clip.close()


def want_n(orc, ids):
    return orc.text_encode(ids, normalize=True, mode=ref.MODE_FAITHFUL)


def test_gelu_models_and_vision_only_text_only(gpu, fixture_cache):
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", use_gelu=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(2, 32, seed=3)
    assert np.all(one_minus_cos(clip.encode_images(imgs), orc.image_batch_encode(imgs)) <= 1e-3)
    pv = fixtures.cached_model(fixture_cache, "tiny", "q8_0", text=False, vision=True)
    cv = gpu.Clip(pv, device=0)
    with pytest.raises(RuntimeError):
        cv.encode_text([49406, 1, 49407])      # tower absent -> false (reference clip.cpp:1018-1021)
    pt = fixtures.cached_model(fixture_cache, "tiny", "q8_0", text=True, vision=False)
    ct = gpu.Clip(pt, device=0)
    with pytest.raises(RuntimeError):
        ct.encode_images(imgs)
    # wrong image size is rejected, not asserted (reference GGML_ASSERT at clip.cpp:1293)
    with pytest.raises(RuntimeError):
        clip.encode_images(np.zeros((1, 16, 16, 3), dtype=np.float32))
    # too many tokens (reference would read out of bounds, clip.cpp:1054-1061)
    with pytest.raises(RuntimeError):
        clip.encode_text([49406] + [5] * 80 + [49407])


def test_compare_and_zero_shot_match_oracle_composition(gpu, fixture_cache):
    """clip_compare_text_and_image / clip_zero_shot_label_image = tokenize -> text -> preprocess -> image -> score."""
    import ctypes as C
    p = fixtures.cached_model(fixture_cache, "tiny", "q5_1")
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(45, 70, 3), dtype=np.uint8)
    u8 = gpu.ClipImageU8(70, 45, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.size)
    score = C.c_float()
    text = "a photo of a red apple"
    assert gpu.lib().clip_compare_text_and_image(clip.ctx, 4, text.encode(), C.byref(u8), C.byref(score))
    pre = orc.preprocess(img)
    ie = orc.image_batch_encode(pre[None], normalize=True)[0]
    te = orc.text_encode(orc.tokenize(text), normalize=True)
    assert abs(score.value - ref.similarity(ie, te)) < 3e-3
    labels = ["cat", "dog", "red apple", "a photo of a car", "tree"]
    scores, idx = clip.zero_shot_label_pixels(C.byref(u8), labels)
    ie_raw = orc.image_batch_encode(pre[None], normalize=False)[0]
    sims = np.array([ref.similarity(ie_raw, orc.text_encode(orc.tokenize(l), normalize=False)) for l in labels], dtype=np.float32)
    s0, i0 = ref.softmax_with_sorting(sims)
    np.testing.assert_allclose(sorted(scores, reverse=True), s0, atol=5e-3)
    assert abs(sum(scores) - 1.0) < 1e-4


def test_vit_b32_q4_0_batch_parity(gpu, fixture_cache):
    """BASELINE config 2 shapes (ViT-B/32 q4_0) on a batch the oracle finishes in seconds."""
    p = fixtures.cached_model(fixture_cache, "b32", "q4_0", text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(4, 224, seed=8)
    got = clip.encode_images(imgs)
    want = orc.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    d = one_minus_cos(got, want)
    assert np.all(d <= TOL_MODEL["q4_0"]), d
    ideal = orc.image_batch_encode(imgs, mode=ref.MODE_IDEAL)
    # the GPU (fp16 activations, exact weights) should sit closer to the ideal network than ggml's q8 activations do
    assert np.median(one_minus_cos(got, ideal)) <= np.median(one_minus_cos(want, ideal)) * 1.5 + 1e-6


def test_baseline_config2_32_images_vs_oracle(gpu, fixture_cache):
    """BASELINE config 2 itself: 32 ViT-B/32 q4_0 images = 1600 token rows, i.e. the mid-M ring kernel (q/k/v, out-projection, FFN-down)
    and the 160 x 128 tile (FFN-up) with the LayerNorm fold, every image against the oracle in ggml-faithful numerics
    (reference clip.cpp:1247-1523).  Observed maximum 1.0e-4 (bench sample): asserted at 3e-4, the documented contract stays 1e-3."""
    p = fixtures.cached_model(fixture_cache, "b32", "q4_0", text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(32, 224, seed=202)
    got = clip.encode_images(imgs)
    want = orc.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    d = one_minus_cos(got, want)
    assert np.all(d <= TOL_MODEL["q4_0"]), d.max()
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


def test_full_size_properties_b32_q4_0_batch256(gpu, fixture_cache):
    """BASELINE metric size (B=256): size-independent properties instead of a 256-image oracle run:
    batch invariance (row i of the batch == image i alone: bit for bit between batches that use the same GEMM
    schedule, to fp32 re-association (1 - cos <= 1e-6) against the small-batch split-K schedule), permutation
    equivariance (bit for bit), determinism (bit for bit), unit norms."""
    p = fixtures.cached_model(fixture_cache, "b32", "q4_0", text=False, vision=True)
    clip = gpu.Clip(p, device=0)
    imgs = fixtures.synthetic_images(256, 224, seed=99)
    full = clip.encode_images(imgs)
    assert full.shape == (256, 512) and np.all(np.isfinite(full))
    np.testing.assert_allclose(np.linalg.norm(full, axis=1), 1.0, atol=1e-5)
    assert np.array_equal(clip.encode_images(imgs), full)                     # run-to-run determinism
    for i in (0, 17, 255):
        one = clip.encode_images(imgs[i:i + 1])                              # batch 1: split-K GEMMs
        assert np.array_equal(one, clip.encode_images(imgs[i:i + 1])), i     # (deterministic: ordered fix-up)
        assert one_minus_cos(one, full[i:i + 1])[0] <= 1e-6, i
        np.testing.assert_allclose(one[0], full[i], atol=3e-4)               # fp16 activation roundings flip on re-association
    # permutation equivariance.  The host pipeline runs a 256-image call as two forwards — 64 + 192 images since round 6 (the first forward hides
    # the H2D of the rest; host_pipeline.cpp host_pipeline_groups) — and the two row counts use different GEMM schedules: bit for bit for a
    # permutation that keeps every image in its forward group, fp32 re-association (1 - cos <= 1e-6) for one that does not.
    rng = np.random.default_rng(0)
    perm_in = np.concatenate([rng.permutation(64), 64 + rng.permutation(192)])
    assert np.array_equal(clip.encode_images(imgs[perm_in]), full[perm_in])
    perm = rng.permutation(256)
    got = clip.encode_images(imgs[perm])
    assert np.all(one_minus_cos(got, full[perm]) <= 1e-6)
    np.testing.assert_allclose(got, full[perm], atol=3e-4)
    sub = clip.encode_images(imgs[:128])                                      # one forward of 128: another schedule than 64 + 192
    assert np.all(one_minus_cos(sub, full[:128]) <= 1e-6)
    sub32 = clip.encode_images(imgs[:32])
    assert np.all(one_minus_cos(sub32, full[:32]) <= 1e-6)
    # the device-pointer entry point runs ONE forward of 256: same embeddings to re-association, and bit-identical under any permutation
    torch = pytest.importorskip("torch")
    d_in, d_out = torch.from_numpy(imgs).cuda(), torch.empty((256, 512), dtype=torch.float32, device="cuda")
    clip.encode_images_device(d_in.data_ptr(), 256, d_out.data_ptr(), True); clip.synchronize()
    dev = d_out.cpu().numpy()
    assert np.all(one_minus_cos(dev, full) <= 1e-6)
    d_in2 = torch.from_numpy(imgs[perm]).cuda()
    clip.encode_images_device(d_in2.data_ptr(), 256, d_out.data_ptr(), True); clip.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), dev[perm])


@pytest.mark.parametrize("config,ftype,n_img", [("b32", "q4_0", 40), ("b32", "f16", 6), ("tiny14", "q5_1", 9)])
def test_layernorm_fold_matches_the_layernorm_kernel_form_end_to_end(gpu, fixture_cache, monkeypatch, config, ftype, n_img):
    """Above 64 rows the layers run with every LayerNorm folded into the GEMM epilogues (5 launches per layer; gemm_common.h).
    CLIP_AMD_LNFOLD=0 keeps the LayerNorm launches: both forms of BOTH towers must agree to fp16-activation rounding
    (1 - cos <= 1e-6, elements to 3e-4), and the folded form is the default."""
    p = fixtures.cached_model(fixture_cache, config, ftype)
    S = 224 if config == "b32" else 28
    imgs = fixtures.synthetic_images(n_img, S, seed=77)
    texts = _ragged_text_batch(24, fixtures.CONFIGS[config]["t"]["npos"], seed=3)
    monkeypatch.delenv("CLIP_AMD_LNFOLD", raising=False)     # (the tier may be run under the A/B switch: this test is about the default)
    clip = gpu.Clip(p, device=0)
    got_i, got_t = clip.encode_images(imgs), clip.encode_texts(texts)
    clip.profile(True)
    clip.encode_images(imgs)
    rep = clip.profile_report(reset=True)
    clip.close()
    ln_launches = sum(v["launches"] for k, v in rep.items() if k.startswith("layernorm") and not k.startswith("layernorm:%dx" % n_img))
    assert ln_launches <= 1, rep.keys()              # only the pre-LN (+ fold entry) launch is tagged (+ one on the n_img pooled rows of the last layer); the post-LN is part of the pooling tail
    monkeypatch.setenv("CLIP_AMD_LNFOLD", "0")
    clip0 = gpu.Clip(p, device=0)
    ref_i, ref_t = clip0.encode_images(imgs), clip0.encode_texts(texts)
    clip0.close()
    monkeypatch.delenv("CLIP_AMD_LNFOLD", raising=False)
    assert np.all(one_minus_cos(got_i, ref_i) <= 1e-6), one_minus_cos(got_i, ref_i).max()
    assert np.all(one_minus_cos(got_t, ref_t) <= 1e-6), one_minus_cos(got_t, ref_t).max()
    atol = 3e-4 if config == "b32" else 1e-3          # (the 64-128-wide test towers: one fp16 rounding flip is a larger share of an element)
    np.testing.assert_allclose(got_i, ref_i, atol=atol)
    np.testing.assert_allclose(got_t, ref_t, atol=atol)


@pytest.mark.parametrize("config,ftype,n_img", [("b32", "q4_0", 40), ("b32", "f16", 5), ("tiny14", "q5_1", 9), ("tiny", "q8_0", 70)])
def test_last_layer_on_pooled_rows_only_matches_the_full_row_form(gpu, fixture_cache, monkeypatch, config, ftype, n_img):
    """Round 4: behind the last layer's attention only the pooled row of each sequence is computed (class token, reference clip.cpp:1426-1431;
    last token of a text, :1154-1155) — out-projection + LayerNorm + FFN on B rows instead of B x T.  Same arithmetic per row, other tiles:
    against CLIP_AMD_PRUNE_LAST=0 (every row through the last layer) both towers agree to fp32 re-association / one fp16 rounding
    (1 - cos <= 1e-6), and the profile shows the pooled GEMMs at M = batch."""
    p = fixtures.cached_model(fixture_cache, config, ftype)
    S = fixtures.CONFIGS[config]["v"]["S"]
    imgs = fixtures.synthetic_images(n_img, S, seed=78)
    texts = _ragged_text_batch(24, fixtures.CONFIGS[config]["t"]["npos"], seed=4)
    monkeypatch.delenv("CLIP_AMD_PRUNE_LAST", raising=False)
    clip = gpu.Clip(p, device=0)
    got_i, got_t = clip.encode_images(imgs), clip.encode_texts(texts)
    clip.profile(True)
    clip.encode_images(imgs)
    clip.encode_texts(texts)
    rep = clip.profile_report(reset=True)
    clip.close()
    pooled = [k for k in rep if "_pooled" in k and k.startswith("gemm")]
    T = (S // fixtures.CONFIGS[config]["v"]["P"]) ** 2 + 1
    if n_img * T > 64:           # (up to 64 token rows the small-M kernels carry every layer: nothing to prune)
        assert sum((":%dx" % n_img) in k for k in pooled) == 3, sorted(rep)
    assert sum((":%dx" % len(texts)) in k for k in pooled) == 3, sorted(rep)
    monkeypatch.setenv("CLIP_AMD_PRUNE_LAST", "0")
    clip0 = gpu.Clip(p, device=0)
    ref_i, ref_t = clip0.encode_images(imgs), clip0.encode_texts(texts)
    clip0.close()
    monkeypatch.delenv("CLIP_AMD_PRUNE_LAST", raising=False)
    assert np.all(one_minus_cos(got_i, ref_i) <= 1e-6), one_minus_cos(got_i, ref_i).max()
    assert np.all(one_minus_cos(got_t, ref_t) <= 1e-6), one_minus_cos(got_t, ref_t).max()
    atol = 3e-4 if config == "b32" else 1e-3
    np.testing.assert_allclose(got_i, ref_i, atol=atol)
    np.testing.assert_allclose(got_t, ref_t, atol=atol)


@pytest.mark.parametrize("ftype", ["q4_0", "f16"])
def test_repeated_encodes_are_bit_identical(gpu, fixture_cache, ftype):
    """Round 6 regression (gemm_common.h GELU_SCALAR_FENCE): the same batch through the same context, eager and graph replays, must return the
    same bits — a mis-compiled activation epilogue made the ViT-B/32 text tower differ by up to 1 - cos = 8e-3 from run to run while staying
    inside every oracle tolerance."""
    p = fixtures.cached_model(fixture_cache, "b32", ftype)
    imgs = fixtures.synthetic_images(21, 224, seed=5)
    texts = _ragged_text_batch(24, fixtures.CONFIGS["b32"]["t"]["npos"], seed=4)
    clip = gpu.Clip(p, device=0)
    i0, t0 = clip.encode_images(imgs), clip.encode_texts(texts)
    for _ in range(5):
        assert np.array_equal(i0, clip.encode_images(imgs))
        assert np.array_equal(t0, clip.encode_texts(texts))
    clip.close()


@pytest.mark.parametrize("config,ftype", [("g88", "f16"), ("g88", "q4_0"), ("g104", "q5_1"), ("g104", "f16")])
def test_head_sizes_88_and_104_end_to_end(gpu, fixture_cache, config, ftype):
    """ViT-g/14 (d_head 88) and ViT-bigG/14 (d_head 104) head sizes, which the reference's generic graph accepts (clip.cpp:463-583, 1366-1388)
    and this library rejected at load until round 4: the attention kernel pads the last 16-wide output tile.  Vision tower against the oracle,
    5 images (T = 5: the small-M path) and 40 (T = 5 x 40 rows / T = 17: the tiled path)."""
    p = fixtures.cached_model(fixture_cache, config, ftype, text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    S = fixtures.CONFIGS[config]["v"]["S"]
    for n in (5, 40):
        imgs = fixtures.synthetic_images(n, S, seed=40 + n)
        got = clip.encode_images(imgs)
        want = orc.image_batch_encode(imgs, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=ref.host_cores())
        d = one_minus_cos(got, want)
        assert np.all(np.isfinite(got)) and np.all(d <= TOL[ftype]), (config, ftype, n, float(d.max()))
    clip.close()


def test_resident_ffn_down_panels_do_not_change_a_bit(gpu, fixture_cache, monkeypatch):
    """Round 4: at ViT-B/32 batch 256 the FFN-down GEMM (12800 x 768 x 3072) runs on the 8-wave kernel over a RESIDENT fp16 panel of the
    q4_0 weight (forward.cpp resident_panels) instead of the fused-dequant 4-wave kernel.  Same dequantised values, same MFMA, same k
    order: the embeddings must be bit-identical to CLIP_AMD_RESIDENT_PANELS=0, on the first call (panels built) and the second."""
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, "b32", "q4_0", text=False, vision=True)
    imgs = torch.from_numpy(fixtures.synthetic_images(256, 224, seed=17)).cuda()

    def run(clip):
        out = torch.empty((256, 512), dtype=torch.float32, device="cuda")
        clip.encode_images_device(imgs.data_ptr(), 256, out.data_ptr(), True)
        clip.synchronize()
        return out.cpu().numpy()

    monkeypatch.setenv("CLIP_AMD_RESIDENT_PANELS", "0")
    c0 = gpu.Clip(p, device=0)
    want = run(c0)
    c0.close()
    monkeypatch.delenv("CLIP_AMD_RESIDENT_PANELS", raising=False)
    c1 = gpu.Clip(p, device=0)
    c1.set_device_shared(True)       # the kernels of the two-tower step: only FFN-down runs on a panel (the panel-less run above is the same either way)
    c1.profile(True)
    first = run(c1)
    rep = c1.profile_report(reset=True)
    c1.profile(False)
    assert any(k.startswith("gemm8_kernel") and "ffn_down" in k for k in rep), sorted(rep)      # the panel kernel carries FFN-down
    assert not any(k.startswith("gemm32_kernel") for k in rep), sorted(rep)
    assert np.array_equal(first, want) and np.array_equal(run(c1), want)
    # round 6: alone on the device a block-quantised ViT-B/32-class batch stays on the same kernels (the 32 x 32 x 16 kernel on resident q/k/v and FFN-up panels
    # was measured behind the trimmed fused-dequant kernels: 79.0-79.3 k against 80.5-80.6 k img/s) -> the same bits
    c1.set_device_shared(False)
    c1.profile(True)
    alone = run(c1)
    rep = c1.profile_report(reset=True)
    c1.profile(False)
    assert not any(k.startswith("gemm32_kernel") for k in rep), sorted(rep)
    assert np.array_equal(alone, want)
    c1.close()
    # ... an f16 file (no panel to build) does take that kernel for q/k/v and FFN-up when it has the device to itself: another k order inside the MFMA ->
    # the embeddings agree to f32 rounding of the sums, not bit for bit; deterministic; back on a shared device the bits return
    p16 = fixtures.cached_model(fixture_cache, "b32", "f16", text=False, vision=True)
    c2 = gpu.Clip(p16, device=0)
    c2.set_device_shared(True)
    want16 = run(c2)
    c2.set_device_shared(False)
    c2.profile(True)
    alone16 = run(c2)
    rep = c2.profile_report(reset=True)
    c2.profile(False)
    assert any(k.startswith("gemm32_kernel<4,4,1>") and "qkv" in k for k in rep) and any(k.startswith("gemm32_kernel<5,4,") and "ffn_up" in k for k in rep), sorted(rep)
    assert float(one_minus_cos(alone16, want16).max()) <= 1e-6, float(one_minus_cos(alone16, want16).max())
    assert np.array_equal(run(c2), alone16)                                    # deterministic
    c2.set_device_shared(True)
    assert np.array_equal(run(c2), want16)
    c2.close()


@pytest.mark.parametrize("config,ftype,B,rule", [("b32", "q4_0", 32, "2,64"), ("b32", "q4_0", 33, "2,64"), ("tiny14", "f16", 12, "2,64"), ("b32", "f16", 8, "2,64"),
                                                 ("b32", "q4_0", 48, None)])        # None: the default rule (48 ViT-B/32 images = 2400 token rows: split)
def test_mid_size_batch_split_over_two_streams_is_the_two_half_batches(gpu, fixture_cache, monkeypatch, config, ftype, B, rule):
    """Round 4: a device-resident call of 8-64 images runs as two half-batches on two streams (the context and its
    weight-sharing sibling), forked / joined with events, captured into ONE hipGraph from the second sighting on.  Bit for bit the
    embeddings of the two halves encoded separately without the split (rows are independent; determinism), on the eager first call, the
    capturing second call and the replays; against the unsplit call of the whole batch: fp32 re-association only."""
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, config, ftype, text=False, vision=True)
    S, proj = fixtures.CONFIGS[config]["v"]["S"], fixtures.CONFIGS[config]["v"]["proj"]
    imgs = torch.from_numpy(fixtures.synthetic_images(B, S, seed=91)).cuda()
    n1 = (B + 1) // 2

    def run(clip, x, reps=1):
        outs = []
        for _ in range(reps):
            out = torch.full((x.shape[0], proj), float("nan"), dtype=torch.float32, device="cuda")
            clip.encode_images_device(x.data_ptr(), x.shape[0], out.data_ptr(), True)
            clip.synchronize()
            outs.append(out.cpu().numpy())
        return outs

    monkeypatch.setenv("CLIP_AMD_SPLIT", "0,0")
    c0 = gpu.Clip(p, device=0)
    halves = np.concatenate([run(c0, imgs[:n1].contiguous())[0], run(c0, imgs[n1:].contiguous())[0]])
    whole = run(c0, imgs)[0]
    c0.close()
    if rule:
        monkeypatch.setenv("CLIP_AMD_SPLIT", rule)   # (forced: the default rule splits by token rows, forward.cpp vision_forward_launch)
    else:
        monkeypatch.delenv("CLIP_AMD_SPLIT", raising=False)
    c1 = gpu.Clip(p, device=0)
    same_ptr_out = torch.empty((B, proj), dtype=torch.float32, device="cuda")
    for i in range(5):                        # same pointers every time: eager, capture, replay x 3
        same_ptr_out.fill_(float("nan"))
        c1.encode_images_device(imgs.data_ptr(), B, same_ptr_out.data_ptr(), True)
        c1.synchronize()
        assert np.array_equal(same_ptr_out.cpu().numpy(), halves), i
    big = torch.from_numpy(fixtures.synthetic_images(2 * B, S, seed=92)).cuda()       # grows the sibling's workspace: earlier graphs must not survive it
    run(c1, big, reps=2)
    for i in range(3):
        same_ptr_out.fill_(float("nan"))
        c1.encode_images_device(imgs.data_ptr(), B, same_ptr_out.data_ptr(), True)
        c1.synchronize()
        assert np.array_equal(same_ptr_out.cpu().numpy(), halves), ("after growth", i)
    c1.close()
    monkeypatch.delenv("CLIP_AMD_SPLIT", raising=False)
    assert np.all(one_minus_cos(halves, whole) <= 1e-6), one_minus_cos(halves, whole).max()


@pytest.mark.parametrize("ftype,dc,spike", [("f16", 6.0, 0.0), ("q4_0", 6.0, 0.0), ("f16", 20.0, 0.0), ("f16", 4.0, 60.0), ("q8_0", 0.0, 80.0)])
def test_layernorm_fold_under_a_large_common_mode_end_to_end(gpu, fixture_cache, monkeypatch, ftype, dc, spike):
    """a ViT-B/32-shaped model whose residual rows carry a common mode of |mean| / std >= ~5 that drifts from layer
    to layer (fixtures dc: pre-LN bias, position embedding, out-projection / FFN-down biases), both towers, against the oracle in the
    reference's numerics (normalise first: clip.cpp:1350-1355).  The default (fold with the operand centred on the previous LayerNorm's
    row mean) is held to TOL_MODEL like every other model-shape test, and to the error of the LayerNorm-launch form; the uncentred r03
    fold is measured beside it for the record (gpurun_out/)."""
    # (spike: three "massive activation" channels tens of sigma away from the rest in every row, as real ViT-L / H checkpoints have them)
    p = fixtures.cached_model(fixture_cache, "b32", ftype, dc=dc, spike=spike)
    orc = ref.OracleModel(p)
    imgs = fixtures.synthetic_images(6, 224, seed=31)
    texts = _ragged_text_batch(16, 77, seed=5)
    want_i = orc.image_batch_encode(imgs, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=ref.host_cores())
    want_t = np.stack([orc.text_encode(t, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=ref.host_cores()) for t in texts])
    res = {}
    for name, env in (("centred", {}), ("uncentred", {"CLIP_AMD_LNFOLD_CENTRE": "0"}), ("launches", {"CLIP_AMD_LNFOLD": "0"})):
        monkeypatch.delenv("CLIP_AMD_LNFOLD", raising=False)
        monkeypatch.delenv("CLIP_AMD_LNFOLD_CENTRE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        clip = gpu.Clip(p, device=0)
        gi, gt = clip.encode_images(imgs), clip.encode_texts(texts)
        clip.close()
        res[name] = (float(one_minus_cos(gi, want_i).max()), float(one_minus_cos(gt, want_t).max()))
    monkeypatch.delenv("CLIP_AMD_LNFOLD", raising=False)
    monkeypatch.delenv("CLIP_AMD_LNFOLD_CENTRE", raising=False)
    line = "b32 %s dc=%g spike=%g: max 1 - cos vs the oracle (images, texts): fold centred %.2e %.2e | fold r03 (uncentred) %.2e %.2e | LayerNorm launches %.2e %.2e" % (
        (ftype, dc, spike) + res["centred"] + res["uncentred"] + res["launches"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_lnfold_centre_e2e.txt"), "a") as f:
        f.write(line + "\n")
    assert res["centred"][0] <= 2.0 * res["launches"][0] + 2e-6 and res["centred"][1] <= 2.0 * res["launches"][1] + 2e-6, line
    if spike and ftype.startswith("q"):
        # Massive-activation channels and a block-quantised file: ggml quantises the ACTIVATIONS to 8 bits in blocks of 32, so a
        # spike costs the other 31 values of its block their resolution — the oracle's faithful mode moves ~1e-3 away from the ideal network on
        # the text tower, in every GPU form alike (fold or not).  The GPU keeps fp16 activations: it must sit at the oracle's IDEAL mode, and
        # the faithful oracle must be the outlier (distance to ideal >= 5 x the GPU's).
        ideal_i = orc.image_batch_encode(imgs, normalize=True, mode=ref.MODE_IDEAL, n_threads=ref.host_cores())
        ideal_t = np.stack([orc.text_encode(t, normalize=True, mode=ref.MODE_IDEAL, n_threads=ref.host_cores()) for t in texts])
        clip = gpu.Clip(p, device=0)
        gi, gt = clip.encode_images(imgs), clip.encode_texts(texts)
        clip.close()
        g_i, g_t = float(one_minus_cos(gi, ideal_i).max()), float(one_minus_cos(gt, ideal_t).max())
        f_t = float(one_minus_cos(want_t, ideal_t).max())
        with open(os.path.join(ROOT, "gpurun_out", "r04_lnfold_centre_e2e.txt"), "a") as f:
            f.write("    ... against the oracle's IDEAL mode: GPU %.2e %.2e; the faithful oracle itself is %.2e away from ideal on the texts\n" % (g_i, g_t, f_t))
        assert g_i <= 1e-4 and g_t <= 1e-4 and f_t >= 5.0 * g_t, (g_i, g_t, f_t)
        assert res["centred"][0] <= TOL[ftype] and res["centred"][1] <= 2.0 * TOL[ftype], line       # documented contract, with the oracle's own noise doubled
    else:
        assert res["centred"][0] <= TOL_MODEL[ftype] and res["centred"][1] <= TOL_MODEL_TEXT[ftype], line


@pytest.mark.parametrize("ftype", ["f16", "q4_0"])
def test_336px_geometry_t577(gpu, fixture_cache, ftype):
    """ViT-L/14@336 geometry (T = 577 tokens, d_head 64): the long-sequence attention instantiation (swizzled K, 155 KB LDS)."""
    p = fixtures.cached_model(fixture_cache, "tiny336", ftype, text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(3, 336, seed=15)
    got = clip.encode_images(imgs)
    want = orc.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    assert np.all(one_minus_cos(got, want) <= TOL[ftype]), one_minus_cos(got, want)


def test_f32_file_at_model_shape_keeps_f32_weights(gpu, fixture_cache):
    """an f32 GGUF is not run at narrower WEIGHT precision than the reference runs it (ggml's f32 vec_dot behind
    clip.cpp:1360) — ViT-B/32, both towers, every row count class (one image, a mid batch, texts of 1-60 tokens): 1 - cos <= 1e-6 against
    the oracle's f32 path, and element-wise a tenth of what fp16-rounded weights would cost."""
    p = fixtures.cached_model(fixture_cache, "b32", "f32")
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(5, 224, seed=77)
    got = clip.encode_images(imgs)
    want = orc.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    d = one_minus_cos(got, want)
    assert np.all(d <= TOL_MODEL["f32"]), d
    assert np.abs(got - want).max() <= 2e-4, np.abs(got - want).max()
    one = clip.encode_images(imgs[:1])
    assert one_minus_cos(one, want[:1])[0] <= TOL_MODEL["f32"]
    texts = fixtures.synthetic_token_ids(6, seed=78, min_len=1, max_len=60)
    got_t = clip.encode_texts(texts)
    want_t = np.stack([orc.text_encode(t, mode=ref.MODE_FAITHFUL) for t in texts])
    dt = one_minus_cos(got_t, want_t)
    assert np.all(dt <= TOL_MODEL["f32"]), dt
    print("f32 file: images 1-cos max %.3g max abs %.3g; texts 1-cos max %.3g" % (d.max(), np.abs(got - want).max(), dt.max()))
    clip.close()


def test_f32_file_keeps_f32_activations_between_the_kernels(gpu, fixture_cache, monkeypatch):
    """Round 6 (VERDICT r5 missing #3 / item 6): for an f32 GGUF the activations between the kernels — LayerNorm output, q/k/v, the attention
    (k_attn_f32.hip) and its output, the GELU output — are f32 as in the reference (ggml f32 x f32, clip.cpp:1360-1422), not fp16: against the
    oracle's f32 path in IDEAL numerics (f32 weights, libm exp / GELU: what f32 arithmetic converges to) both towers hold 1 - cos <= 5e-7 at
    model shape — ten times closer than the fp16-activation form of round 5 (CLIP_AMD_F32_ACTS=0), which is measured beside it — and the
    ggml-faithful path (fp16 exp / GELU tables, as ggml) stays inside TOL_MODEL."""
    p = fixtures.cached_model(fixture_cache, "b32", "f32")
    orc = ref.OracleModel(p)
    imgs = fixtures.synthetic_images(4, 224, seed=79)
    texts = fixtures.synthetic_token_ids(5, seed=80, min_len=1, max_len=70)
    ideal_i = orc.image_batch_encode(imgs, mode=ref.MODE_IDEAL)
    ideal_t = np.stack([orc.text_encode(t, mode=ref.MODE_IDEAL) for t in texts])
    faith_i = orc.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    res = {}
    for name, env in (("f32", None), ("fp16", "0")):
        if env is None:
            monkeypatch.delenv("CLIP_AMD_F32_ACTS", raising=False)
        else:
            monkeypatch.setenv("CLIP_AMD_F32_ACTS", env)
        clip = gpu.Clip(p, device=0)
        gi, gt = clip.encode_images(imgs), clip.encode_texts(texts)
        big = clip.encode_images(np.concatenate([imgs] * 12)[:45])            # 2250 token rows: the tiled path at another row count
        clip.close()
        assert np.all(one_minus_cos(big[:4], gi) <= 5e-7)                        # (same rows, other GEMM tiles: f32 re-association)
        res[name] = (float(one_minus_cos(gi, ideal_i).max()), float(one_minus_cos(gt, ideal_t).max()), float(np.abs(gi - ideal_i).max()), float(one_minus_cos(gi, faith_i).max()))
    monkeypatch.delenv("CLIP_AMD_F32_ACTS", raising=False)
    print("f32 file vs ideal f32: f32 activations images %.3g texts %.3g max abs %.3g (vs faithful %.3g) | fp16 activations images %.3g texts %.3g max abs %.3g" %
          (res["f32"][0], res["f32"][1], res["f32"][2], res["f32"][3], res["fp16"][0], res["fp16"][1], res["fp16"][2]))
    assert res["f32"][0] <= 5e-7 and res["f32"][1] <= 5e-7, res
    assert res["f32"][3] <= TOL_MODEL["f32"], res
    assert res["f32"][2] <= res["fp16"][2], res                                 # element-wise no worse than the fp16-activation form


def test_vit_l14_f16_shapes(gpu, fixture_cache):
    """BASELINE config 3 shapes (ViT-L/14 f16, T=257, d_head 64): 2 images vs the oracle."""
    p = fixtures.cached_model(fixture_cache, "l14", "f16", text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(2, 224, seed=12)
    got = clip.encode_images(imgs)
    want = orc.image_batch_encode(imgs, mode=ref.MODE_FAITHFUL)
    d = one_minus_cos(got, want)
    assert np.all(d <= TOL_MODEL["f16"]), d


@pytest.mark.parametrize("config,ftype", [("l14", "q5_1"), ("h14", "q8_0")])
def test_large_model_shapes_quantised(gpu, fixture_cache, config, ftype):
    """BASELINE config 4 (ViT-L/14 q5_1) and config 5 (ViT-H/14 q8_0: d_head 80, 32 layers) shapes: one image vs the oracle,
    plus batch consistency (the same image inside a batch of 3)."""
    p = fixtures.cached_model(fixture_cache, config, ftype, text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(3, 224, seed=31)
    got = clip.encode_images(imgs)
    want = orc.image_batch_encode(imgs[:1], mode=ref.MODE_FAITHFUL)
    assert one_minus_cos(got[:1], want)[0] <= TOL_MODEL[ftype], one_minus_cos(got[:1], want)
    assert one_minus_cos(clip.encode_images(imgs[:1]), got[:1])[0] <= 1e-6


@pytest.mark.parametrize("ftype", ["f16", "q5_1"])
def test_vit_l14_batch130_largest_m_kernels_match_single_images(gpu, fixture_cache, ftype):
    """130 ViT-L/14 images = 33410 token rows: the regime of the 256 x 256 four-wave GEMM (k_gemm4.hip) with its whole-rounds split
    and, for block-quantised files, of the per-layer fp16 weight panels.  No oracle run at this size: batch invariance against
    single-image forwards (other kernels, other tiles: equal up to fp32 re-association of split-K and fp16 activation roundings)
    and run-to-run determinism."""
    p = fixtures.cached_model(fixture_cache, "l14", ftype, text=False, vision=True)
    clip = gpu.Clip(p, device=0)
    imgs = fixtures.synthetic_images(130, 224, seed=57)
    full = clip.encode_images(imgs)
    assert full.shape[0] == 130 and np.all(np.isfinite(full))
    np.testing.assert_allclose(np.linalg.norm(full, axis=1), 1.0, atol=1e-5)
    assert np.array_equal(clip.encode_images(imgs), full)
    for i in (0, 64, 129):
        one = clip.encode_images(imgs[i:i + 1])
        assert one_minus_cos(one, full[i:i + 1])[0] <= 1e-6, i
        np.testing.assert_allclose(one[0], full[i], atol=3e-4)


@pytest.mark.parametrize("config,ftype,B,rows,expect", [
    # BASELINE config 3: ViT-L/14 f16, 256 images = 65792 token rows in ONE call: 256 x 256 four-wave GEMMs (k_gemm4.hip) on the whole rounds + a
    # second launch for the rows past them (image 255 straddles that seam), LayerNorm launches (fold_pays() declines), attn_kernel<17, ...>
    ("l14", "f16", 256, [0, 1, 63, 127, 128, 200, 254, 255], ["gemm4_kernel<", "layernorm", "attention"]),
    # config 4, one GPU's shard: ViT-L/14 q5_1, 128 images = 32896 rows: the same kernels on per-layer fp16 panels (dequant_layer)
    ("l14", "q5_1", 128, [0, 1, 31, 63, 64, 100, 126, 127], ["gemm4_kernel<", "dequant_layer", "layernorm"]),
    # config 5, one batch of the zero-shot harness: ViT-H/14 q8_0 (d_head 80, 32 layers), 64 images = 16448 rows: fused-dequant q8_0 tiles
    ("h14", "q8_0", 64, [0, 1, 15, 31, 32, 47, 62, 63], ["gemm_dma_kernel<5,", "attention"]),
])
def test_baseline_configs_3_4_5_at_their_benchmarked_batch_against_the_oracle(gpu, fixture_cache, config, ftype, B, rows, expect):
    """the kernels BASELINE configs 3-5 are TIMED on, checked against the oracle (reference path clip.cpp:1247-1523) and
    not only by batch invariance: the whole benchmarked batch goes through the device-pointer entry bench.py times in ONE call, and eight
    embeddings spread over it — first, last, the images either side of the middle, the one that straddles the whole-rounds seam of the
    256 x 256 kernel — are compared with the oracle in ggml-faithful numerics at TOL_MODEL.  A profiled repeat of the same call names the
    kernels that ran, so the test fails if the dispatch ever moves these shapes off the kernels it is meant to cover."""
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, config, ftype, text=False, vision=True)
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    proj = clip.vision_config["projection_dim"]
    imgs = fixtures.synthetic_images(B, 224, seed=300 + B)
    d_in = torch.from_numpy(imgs).cuda()
    d_out = torch.full((B, proj), float("nan"), dtype=torch.float32, device="cuda")
    clip.encode_images_device(d_in.data_ptr(), B, d_out.data_ptr(), True)
    clip.synchronize()
    got = d_out.cpu().numpy()
    assert np.all(np.isfinite(got))
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    want = orc.image_batch_encode(imgs[rows], mode=ref.MODE_FAITHFUL)
    d = one_minus_cos(got[rows], want)
    assert np.all(d <= TOL_MODEL[ftype]), (config, ftype, B, d.tolist())
    # the kernels behind that call (HIP-event profiling of a repeat; same dispatch — profiling only disables graphs and the mid-batch split)
    clip.profile(True)
    d_out.fill_(float("nan"))
    clip.encode_images_device(d_in.data_ptr(), B, d_out.data_ptr(), True)
    clip.synchronize()
    tags = list(clip.profile_report().keys())
    clip.profile(False)
    assert np.array_equal(d_out.cpu().numpy(), got)
    for e in expect:
        assert any(e in t for t in tags), (e, tags)
    print("%s %s batch %d: 1-cos vs oracle over rows %s: max %.3g mean %.3g" % (config, ftype, B, rows, d.max(), d.mean()))
    clip.close()


def test_batches_beyond_one_workspace_chunk_and_stream_restore(gpu, fixture_cache):
    """More images than one forward chunk (1024) and than one host-API staging chunk (256): rows must equal the small-batch
    results bit for bit (tiny model: no split-K at any size; batches of > 64 token rows: the LayerNorm statistics do not depend on
    the tile shapes either); clip_amd_set_stream(NULL) restores the context's own stream."""
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=False, vision=True)
    clip = gpu.Clip(p, device=0)
    imgs = fixtures.synthetic_images(1030, 32, seed=77)
    full = clip.encode_images(imgs)                                   # host API: 5 staging chunks
    assert full.shape == (1030, 32) and np.all(np.isfinite(full))
    for i in (0, 255, 256, 1023, 1024, 1026):
        # 4 images = 68 token rows: other tiles, other kernels (ring instead of the large tiles), the same LayerNorm-folded chain
        assert np.array_equal(clip.encode_images(imgs[i:i + 4])[0], full[i]), i
        # one image = 17 rows: the small-M kernels (LayerNorm fused on the operand instead of folded into the epilogue)
        one = clip.encode_images(imgs[i:i + 1])
        assert one_minus_cos(one, full[i:i + 1])[0] <= 1e-6, i
        np.testing.assert_allclose(one[0], full[i], atol=1e-3)
    d_in = torch.from_numpy(imgs).cuda()
    d_out = torch.empty((1030, 32), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        clip.set_stream(st.cuda_stream)
        clip.encode_images_device(d_in.data_ptr(), 1030, d_out.data_ptr())   # device API: two forward chunks (1024 + 6)
        st.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), full)
    clip.set_stream(0)                                                 # NULL -> own stream again
    assert np.array_equal(clip.encode_images(imgs[:5]), full[:5])


@pytest.mark.parametrize("n_threads", [1, 2, 16])
def test_host_api_with_one_thread_and_more_than_two_staging_chunks(gpu, fixture_cache, n_threads):
    """with n_threads = 1 the host pipeline used to pack EVERY 256-image chunk before driving any, and with only two
    pinned buffers chunk c >= 2 overwrote chunk c - 2 -> wrong images in the early rows of calls with more than 512 images.  Every row
    of a 1030-image call must equal the same image in a small call, for one, two and many packing threads."""
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=False, vision=True)
    clip = gpu.Clip(p, device=0)
    imgs = fixtures.synthetic_images(1030, 32, seed=123)
    full = clip.encode_images(imgs, n_threads=n_threads)
    assert full.shape == (1030, 32) and np.all(np.isfinite(full))
    for i0 in (0, 100, 252, 256, 508, 512, 768, 1020, 1026):
        assert np.array_equal(clip.encode_images(imgs[i0:i0 + 4], n_threads=n_threads), full[i0:i0 + 4]), i0
    assert np.array_equal(clip.encode_images(imgs, n_threads=16), full)


def test_device_entry_points_with_torch_memory(gpu, fixture_cache):
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    clip = gpu.Clip(p, device=0)
    imgs = fixtures.synthetic_images(7, 32, seed=1)
    host = clip.encode_images(imgs)
    d_in = torch.from_numpy(imgs).cuda()
    d_out = torch.empty((7, 32), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    clip.set_stream(st.cuda_stream)
    clip.encode_images_device(d_in.data_ptr(), 7, d_out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), host)
    texts = fixtures.synthetic_token_ids(4, seed=2, max_len=12)
    flat = np.concatenate(texts).astype(np.int32)
    off = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    d_ids = torch.from_numpy(flat).cuda()
    d_t = torch.empty((4, 32), dtype=torch.float32, device="cuda")
    clip.encode_texts_device(d_ids.data_ptr(), off, d_t.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(d_t.cpu().numpy(), clip.encode_texts(texts), atol=1e-6)


@pytest.mark.parametrize("config,S", [("tiny", 32), ("b32", 224)])
def test_gpu_preprocessing_is_bit_identical_to_host(gpu, fixture_cache, config, S):
    """resize / centre-crop / normalise on the GPU (clip_amd_image_batch_preprocess_device) against the host
    implementation of clip_image_preprocess (itself bit-exact vs the oracle, tests/test_host_api.py): identical bits for
    down-scaling, up-scaling, extreme aspect ratios and already-square inputs, in one mixed-size batch."""
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, config, "q4_0", text=False, vision=True)
    clip = gpu.Clip(p, device=0)
    rng = np.random.default_rng(77)
    sizes = [(45, 70), (S, S), (375, 500), (500, 375), (S, 3 * S + 1), (1000, 37), (17, 23), (1, 1), (S + 1, S - 1), (768, 1024)]
    images = [rng.integers(0, 256, size=(ny, nx, 3), dtype=np.uint8) for ny, nx in sizes]
    images[2][:] = 255            # saturated image: exercises the clamp after each pass
    images[3][::2] = 0
    d_out = torch.full((len(images), S, S, 3), float("nan"), dtype=torch.float32, device="cuda")
    clip.preprocess_device(images, d_out.data_ptr())
    got = d_out.cpu().numpy()
    for i, im in enumerate(images):
        want = clip.preprocess(im)
        assert np.array_equal(got[i], want), (i, sizes[i], np.abs(got[i] - want).max())
    # end to end: raw u8 -> embeddings == host preprocess + encode, bit for bit
    emb = clip.encode_images_u8(images)
    want = clip.encode_images(np.stack([clip.preprocess(im) for im in images]))
    assert np.array_equal(emb, want)
    # a second, differently shaped batch through the same context (buffers are re-used / re-grown)
    more = [rng.integers(0, 256, size=(ny, nx, 3), dtype=np.uint8) for ny, nx in [(300, 200), (64, 64), (90, 400)]]
    assert np.array_equal(clip.encode_images_u8(more), clip.encode_images(np.stack([clip.preprocess(im) for im in more])))


def test_u8_encode_of_several_chunks_is_the_chunkwise_result(gpu, fixture_cache):
    """clip_amd_image_batch_encode_u8 beyond 256 images per call: double-buffered staging (the host fills the next chunk's pinned blob and
    a copy stream ships it while the GPU preprocesses and encodes the current chunk; round 4).  The embeddings must be the bits of the
    same images encoded chunk by chunk through the single-slot path, for mixed image sizes, twice through the same context (slot re-use),
    and both the contiguous-ndarray and the list form of the wrapper."""
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=False, vision=True)
    clip = gpu.Clip(p, device=0)
    rng = np.random.default_rng(12)
    sizes = [(32, 32), (40, 64), (33, 90), (64, 48), (100, 37)]
    images = [rng.integers(0, 256, size=sizes[i % len(sizes)] + (3,), dtype=np.uint8) for i in range(700)]      # 3 chunks: 256 + 256 + 188
    want = np.concatenate([clip.encode_images_u8(images[i:i + 256]) for i in range(0, 700, 256)])
    for _ in range(2):
        assert np.array_equal(clip.encode_images_u8(images), want)
    block = rng.integers(0, 256, size=(520, 32, 32, 3), dtype=np.uint8)                                         # one contiguous [B, ny, nx, 3] block
    want_b = np.concatenate([clip.encode_images_u8(list(block[i:i + 256])) for i in range(0, 520, 256)])
    assert np.array_equal(clip.encode_images_u8(block), want_b)
    assert np.array_equal(clip.encode_images_u8(images[:300]), want[:300])       # a shorter multi-chunk call afterwards
    clip.close()


def test_batched_zero_shot_on_gpu_matches_per_image_reference_composition(gpu, fixture_cache):
    """clip_amd_zero_shot_label_images (labels encoded once, images preprocessed + encoded + scored on the
    GPU) == clip_zero_shot_label_image called per image (host scoring with the reference arithmetic)."""
    import ctypes as C
    p = fixtures.cached_model(fixture_cache, "tiny", "q8_0")
    clip = gpu.Clip(p, device=0)
    rng = np.random.default_rng(5)
    images = [rng.integers(0, 256, size=(ny, nx, 3), dtype=np.uint8) for ny, nx in [(45, 70), (64, 64), (100, 33), (32, 32), (80, 120)]]
    labels = ["cat", "dog", "red apple", "a photo of a car", "tree", "cat", "the quick brown fox", "x", "sky", "a", "b c d"]   # incl. a duplicate (tie)
    scores, idx = clip.zero_shot_label_images(images, labels)
    assert scores.shape == (5, len(labels)) and idx.shape == (5, len(labels))
    for i, im in enumerate(images):
        u8 = gpu.ClipImageU8(im.shape[1], im.shape[0], im.ctypes.data_as(C.POINTER(C.c_uint8)), im.size)
        s1, i1 = clip.zero_shot_label_pixels(C.byref(u8), labels)
        np.testing.assert_allclose(scores[i], s1, rtol=1e-4, atol=1e-6)     # (labels as ONE folded batch vs one label at a time on the small-M kernels)
        assert list(idx[i]) == i1, (i, list(idx[i]), i1)
        assert abs(scores[i].sum() - 1.0) < 1e-5 and np.all(np.diff(scores[i]) <= 0)
        assert sorted(idx[i]) == list(range(len(labels)))


@pytest.mark.parametrize("B,n,dim", [(3, 1000, 512), (1, 1, 64), (2, 8192, 128), (4, 37, 1024)])
def test_zero_shot_scoring_kernel_vs_reference_arithmetic(gpu, fixture_cache, B, n, dim):
    torch = pytest.importorskip("torch")
    clip = gpu.Clip(fixtures.cached_model(fixture_cache, "tiny", "q8_0"), device=0)
    rng = np.random.default_rng(B * 1000 + n)
    img = (rng.standard_normal((B, dim)) * 0.3).astype(np.float32)
    txt = (rng.standard_normal((n, dim)) * 0.3).astype(np.float32)
    if n > 4:
        txt[3] = txt[1]                                   # exact tie -> lower index first (stable sort)
    d_img, d_txt = torch.from_numpy(img).cuda(), torch.from_numpy(txt).cuda()
    d_sc = torch.empty((B, n), dtype=torch.float32, device="cuda")
    d_ix = torch.empty((B, n), dtype=torch.int32, device="cuda")
    assert gpu.lib().clip_amd_zero_shot_score_device(clip.ctx, d_img.data_ptr(), B, d_txt.data_ptr(), n, dim, d_sc.data_ptr(), d_ix.data_ptr())
    clip.synchronize()
    sc, ix = d_sc.cpu().numpy(), d_ix.cpu().numpy()
    for b in range(B):
        sims = np.array([ref.similarity(img[b], txt[j]) for j in range(n)], dtype=np.float32)   # sequential fp32 dot
        s0, i0 = ref.softmax_with_sorting(sims)
        np.testing.assert_allclose(sc[b], s0, rtol=2e-6, atol=1e-12)
        same = ix[b] == i0
        # indices may differ only where neighbouring scores are equal to the last bit (exp ulp differences)
        assert np.all(same | np.isclose(sc[b], s0[np.argsort(np.argsort(-sc[b], kind="stable"))], rtol=2e-6)), b
        assert sorted(ix[b]) == list(range(n))
        if n > 4:
            p1, p3 = int(np.where(ix[b] == 1)[0][0]), int(np.where(ix[b] == 3)[0][0])
            assert p3 == p1 + 1


def test_load_from_repacked_weight_cache_encodes_bit_identically(gpu, fixture_cache, tmp_path, monkeypatch):
    """the cached HBM image IS the repacked image — same embeddings to the last bit, both towers, a quantised file
    (planes + dequantised tables + raw token table) and an f16 one."""
    for ftype in ("q4_0", "f16"):
        p = fixtures.cached_model(fixture_cache, "tiny14", ftype)
        imgs = fixtures.synthetic_images(3, 28, seed=3)
        toks = [[49406, 320, 1125, 539, 320, 2368, 49407], [49406, 49407]]
        monkeypatch.delenv("CLIP_AMD_WEIGHT_CACHE", raising=False)
        plain = gpu.Clip(p, device=0)
        want_i, want_t = plain.encode_images(imgs), plain.encode_texts(toks)
        plain.close()
        monkeypatch.setenv("CLIP_AMD_WEIGHT_CACHE", str(tmp_path))
        first = gpu.Clip(p, device=0)
        assert not first.weights_from_cache
        first.close()
        # the published image carries the mode fopen would have given it (a shared cache directory stays shared; mkstemp's 0600
        # would not), and a temporary left behind by a writer that died before its rename is swept when the cache is opened again
        hbm = [f for f in os.listdir(tmp_path) if f.endswith(".hbm") and os.path.basename(p) in f]
        assert len(hbm) == 1, os.listdir(tmp_path)
        um = os.umask(0); os.umask(um)
        assert (os.stat(tmp_path / hbm[0]).st_mode & 0o777) == (0o666 & ~um)
        stale, fresh = tmp_path / (hbm[0] + ".tmp.dEaDbF"), tmp_path / (hbm[0] + ".tmp.aLiVe0")
        stale.write_bytes(b"x" * 64); fresh.write_bytes(b"y" * 64)
        os.utime(stale, (1.0e9, 1.0e9))
        cached = gpu.Clip(p, device=0)
        assert cached.weights_from_cache
        assert not stale.exists() and fresh.exists()
        fresh.unlink()
        assert np.array_equal(cached.encode_images(imgs), want_i) and np.array_equal(cached.encode_texts(toks), want_t)
        cached.close()


def test_graph_replay_is_bitwise_identical_to_eager(gpu, fixture_cache):
    """Small batches are captured into a hipGraph on the 2nd call with the same signature and replayed afterwards:
    eager (1st) == capture (2nd) == replay (3rd...), and a replay sees new pixel data written into the same buffers."""
    p = fixtures.cached_model(fixture_cache, "tiny14", "q4_0")
    clip = gpu.Clip(p, device=0)
    a = fixtures.synthetic_images(3, 28, seed=41)
    b = fixtures.synthetic_images(3, 28, seed=42)
    ea = clip.encode_images(a)              # eager
    for _ in range(4):                      # capture, then replays
        assert np.array_equal(clip.encode_images(a), ea)
    eb = clip.encode_images(b)              # replay on new data
    assert not np.array_equal(ea, eb)
    clip2 = gpu.Clip(p, device=0)           # fresh context: eager reference for b
    assert np.array_equal(clip2.encode_images(b), eb)
    one = clip.encode_images(a[:1])         # different batch -> different signature
    assert np.array_equal(one[0], ea[0])


def test_text_graph_replay_is_bitwise_identical_to_eager(gpu, fixture_cache):
    """Single texts / short lists replay a captured hipGraph from the third call with the same (texts, tokens) signature on:
    eager == capture == replay, a replay sees NEW ids and NEW per-text lengths (same total), and a fresh context agrees."""
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    clip, fresh = gpu.Clip(p, device=0), gpu.Clip(p, device=0)
    a = [49406, 11, 12, 13, 14, 49407]
    b = [49406, 21, 22, 23, 24, 49407]
    ea = np.asarray(clip.encode_text(a), dtype=np.float32)
    for _ in range(4):
        assert np.array_equal(np.asarray(clip.encode_text(a), dtype=np.float32), ea)
    eb = np.asarray(clip.encode_text(b), dtype=np.float32)                       # replay with new ids
    assert not np.array_equal(ea, eb)
    assert np.array_equal(eb, np.asarray(fresh.encode_text(b), dtype=np.float32))  # fresh context = eager
    # ragged lists with the same totals but different splits share one graph (offsets are device data)
    l1 = [np.array([49406, 5, 6, 49407], np.int32), np.array([49406, 7, 8, 9, 10, 11, 49407], np.int32)]
    l2 = [np.array([49406, 5, 6, 9, 10, 11, 49407], np.int32), np.array([49406, 7, 8, 49407], np.int32)]
    for _ in range(3):
        r1 = clip.encode_texts(l1)
    r2 = clip.encode_texts(l2)                                                    # replay: other lengths, same rows / bucket
    np.testing.assert_array_equal(r1, fresh.encode_texts(l1))
    np.testing.assert_array_equal(r2, fresh.encode_texts(l2))


def test_reference_example_main_runs_unchanged_on_the_gpu_library(gpu, fixture_cache, tmp_path):
    """BASELINE config 1 ("plumbing"): the reference's examples/main.cpp, compiled unchanged against include/ and linked to
    libclip.so (oracle/_ref/ref_main, built in the dev container by `make -C oracle ref`), loads a ViT-B/32 f16 two-tower GGUF,
    decodes a JPEG, tokenizes, encodes both towers on the GPU and prints the similarity score."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_main")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_main not built (needs the reference tree)")
    PIL = pytest.importorskip("PIL.Image")
    p = fixtures.cached_model(fixture_cache, "b32", "f16")
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:300, 0:400]
    img = np.clip(np.stack([(np.sin(xx / 29.0) * 0.5 + 0.5) * 255, (np.cos(yy / 19.0) * 0.5 + 0.5) * 255, (xx + 2 * yy) % 256], -1)
                  + rng.normal(0, 6, (300, 400, 3)), 0, 255).astype(np.uint8)
    jpg = str(tmp_path / "photo.jpg")
    PIL.fromarray(img).save(jpg, "JPEG", quality=90, progressive=True)
    text = "a photo of a red apple"
    out = subprocess.run([exe, "-m", p, "--image", jpg, "--text", text, "-v", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"Similarity score = ([-0-9.]+)", out.stdout)
    assert m, out.stdout[-2000:]
    # oracle composition on the same decoded pixels
    L = gpu.lib()
    u8 = L.clip_image_u8_make()
    assert L.clip_image_load_from_file(os.fsencode(jpg), u8)
    pix = np.ctypeslib.as_array(u8.contents.data, shape=(u8.contents.ny, u8.contents.nx, 3)).copy()
    L.clip_image_u8_free(u8)
    orc = ref.OracleModel(p)
    ie = orc.image_batch_encode(orc.preprocess(pix)[None], normalize=True)[0]
    te = orc.text_encode(orc.tokenize(text), normalize=True)
    assert abs(float(m.group(1)) - ref.similarity(ie, te)) < 2e-3, (m.group(1), ref.similarity(ie, te))


def test_reference_benchmark_program_runs_unchanged_on_the_gpu_library(gpu, fixture_cache, tmp_path):
    """BASELINE config 5 flow: the reference's tests/benchmark.cpp compiled unchanged (oracle/_ref/ref_benchmark) walks a
    class-per-directory image tree, encodes the class names and the images (batches of 4) through libclip.so on the GPU
    and prints zero-shot acc@1 / acc@5 per class.  Expected numbers: the same composition through the Python binding."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_benchmark")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_benchmark not built (needs the reference tree)")
    PIL = pytest.importorskip("PIL.Image")
    p = fixtures.cached_model(fixture_cache, "b32", "f16")
    rng = np.random.default_rng(11)
    classes = ["cat", "dog", "red apple", "car", "tree", "house", "boat"]
    root = tmp_path / "tree"
    files = {}
    for ci, c in enumerate(classes):
        d = root / c
        d.mkdir(parents=True)
        files[c] = []
        for k in range(4):
            ny, nx = int(rng.integers(60, 300)), int(rng.integers(60, 300))
            yy, xx = np.mgrid[0:ny, 0:nx]
            img = np.clip(np.stack([(np.sin(xx / (5.0 + ci)) * 0.5 + 0.5) * 255, (np.cos(yy / (3.0 + k)) * 0.5 + 0.5) * 255,
                                    (xx * (ci + 1) + yy * (k + 1)) % 256], -1) + rng.normal(0, 10, (ny, nx, 3)), 0, 255).astype(np.uint8)
            f = str(d / ("img%d.%s" % (k, "png" if k % 2 else "jpg")))
            PIL.fromarray(img).save(f)
            files[c].append(f)
    outf = str(tmp_path / "bench.txt")
    out = subprocess.run([exe, p, str(root), "0", outf], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    report = open(outf).read()
    got = {m.group(1).strip(): (float(m.group(2)), float(m.group(3))) for m in re.finditer(r"^\| (.{20}) \| ([0-9.]+) \| ([0-9.]+) \|$", report, re.M)}
    assert set(classes) <= set(got), report
    # expected: same composition via the binding (labels sorted like the std::map keys of benchmark.cpp)
    clip = gpu.Clip(p, device=0)
    L = gpu.lib()
    order = sorted(classes)
    txt = np.stack([np.asarray(clip.encode_text(clip.tokenize(c), normalize=True), dtype=np.float32) for c in order])
    close_calls = 0
    for li, c in enumerate(order):
        a1 = a5 = 0
        for f in files[c]:
            u8 = L.clip_image_u8_make()
            assert L.clip_image_load_from_file(os.fsencode(f), u8)
            pix = np.ctypeslib.as_array(u8.contents.data, shape=(u8.contents.ny, u8.contents.nx, 3)).copy()
            L.clip_image_u8_free(u8)
            e = clip.encode_images(clip.preprocess(pix)[None], normalize=True)[0]
            sims = np.array([ref.similarity(e, t) for t in txt], dtype=np.float32)
            rank = list(np.argsort(-sims, kind="stable"))
            srt = np.sort(sims)[::-1]
            close_calls += int(np.min(np.abs(np.diff(srt))) < 1e-5)
            a1 += rank[0] == li
            a5 += li in rank[:5]
        if close_calls == 0:
            assert abs(got[c][0] - a1 / 4) < 1e-4 and abs(got[c][1] - a5 / 4) < 1e-4, (c, got[c], a1, a5, report)
    assert "28 images encoded" in report and "7 texts encoded" in report, report


def _ref_program(name):
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s not built (needs the reference tree)" % name)
    return exe


def _photo(path, seed, ny=260, nx=340, **save):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:ny, 0:nx]
    img = np.clip(np.stack([(np.sin(xx / 23.0) * 0.5 + 0.5) * 255, (np.cos(yy / 17.0) * 0.5 + 0.5) * 255, (3 * xx + yy) % 256], -1)
                  + rng.normal(0, 8, (ny, nx, 3)), 0, 255).astype(np.uint8)
    PIL.fromarray(img).save(path, **save)


def _decoded(gpu, path):
    import os
    L = gpu.lib()
    u8 = L.clip_image_u8_make()
    assert L.clip_image_load_from_file(os.fsencode(path), u8)
    pix = np.ctypeslib.as_array(u8.contents.data, shape=(u8.contents.ny, u8.contents.nx, 3)).copy()
    L.clip_image_u8_free(u8)
    return pix


def test_reference_zero_shot_program_runs_unchanged_on_the_gpu_library(gpu, fixture_cache, tmp_path):
    """examples/zsl.cpp compiled unchanged (oracle/_ref/ref_zsl): load, decode, clip_zero_shot_label_image, print "label = score"
    lines in descending order.  Expected: the oracle's composition (un-normalised embeddings, cosine, softmax_with_sorting)."""
    import re
    import subprocess
    exe = _ref_program("ref_zsl")
    p = fixtures.cached_model(fixture_cache, "b32", "f16")      # (f16 as for examples/main above: the oracle's q8 activation rounding moves a cosine by ~1e-2)
    jpg = str(tmp_path / "z.jpg")
    _photo(jpg, 21, quality=92)
    labels = ["cat", "dog", "a red apple", "a photo of a car", "tree"]
    cmd = [exe, "-m", p, "--image", jpg, "-v", "0"]
    for l in labels:
        cmd += ["--text", l]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = [(m.group(1), float(m.group(2))) for m in re.finditer(r"^(.+) = ([0-9.]+)$", out.stdout, re.M)]
    assert sorted(l for l, _ in got) == sorted(labels), out.stdout
    assert all(a[1] >= b[1] for a, b in zip(got, got[1:])), got                  # printed best first
    orc = ref.OracleModel(p)
    ie = orc.image_batch_encode(orc.preprocess(_decoded(gpu, jpg))[None], normalize=False)[0]
    sims = np.array([ref.similarity(ie, orc.text_encode(orc.tokenize(l), normalize=False)) for l in labels], dtype=np.float32)
    s0, i0 = ref.softmax_with_sorting(sims)
    np.testing.assert_allclose([s for _, s in got], s0, atol=5e-3, err_msg=out.stdout)
    if np.min(np.abs(np.diff(s0))) > 1e-2:
        assert [l for l, _ in got] == [labels[i] for i in i0]


def test_reference_extract_program_runs_unchanged_on_the_gpu_library(gpu, fixture_cache, tmp_path):
    """examples/extract.cpp compiled unchanged (oracle/_ref/ref_extract): one .npy per image (clip_image_encode, not normalised)
    and per text (clip_text_encode, not normalised) in the working directory; vectors against the oracle."""
    import subprocess
    exe = _ref_program("ref_extract")
    ftype = "q8_0"
    p = fixtures.cached_model(fixture_cache, "b32", ftype)
    files = [str(tmp_path / "a.png"), str(tmp_path / "b.jpg")]
    _photo(files[0], 31, ny=224, nx=224)
    _photo(files[1], 32, ny=500, nx=281, quality=85)
    texts = ["a photo of a cat", "two dogs playing in the snow, seen from far away"]
    cmd = [exe, "-m", p, "-v", "0"]
    for f in files:
        cmd += ["--image", f]
    for t in texts:
        cmd += ["--text", t]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    orc = ref.OracleModel(p)
    for f in files:
        v = np.load(str(tmp_path / ("img_vec_%s.npy" % f.rsplit("/", 1)[1])))
        want = orc.image_batch_encode(orc.preprocess(_decoded(gpu, f))[None], normalize=False)[0]
        assert v.shape == (1, want.size) and v.dtype == np.float32
        assert one_minus_cos(v[0], want) < TOL_MODEL[ftype]
        assert abs(np.linalg.norm(v[0]) / np.linalg.norm(want) - 1) < 2e-2
    for i, t in enumerate(texts):
        v = np.load(str(tmp_path / ("text_vec_%d.npy" % i)))
        want = orc.text_encode(orc.tokenize(t), normalize=False)
        assert v.shape == (1, want.size)
        assert one_minus_cos(v[0], want) < TOL_MODEL_TEXT[ftype]
        assert abs(np.linalg.norm(v[0]) / np.linalg.norm(want) - 1) < 2e-2


@pytest.mark.parametrize("ftype", ["f16", "q4_0"])
@pytest.mark.parametrize("config", ["tiny", "tiny14"])
def test_every_batch_size_across_the_dispatch_boundaries_matches_the_oracle(gpu, fixture_cache, config, ftype):
    """One small model, batches of 1 ... 300 images (17 / 5 token rows each) and 1 ... 260 ragged texts: the row counts cross every
    boundary the dispatcher has — the <= 64-row kernels, the ring tiles, the 2000-3300-row two-stream split, the 4096-row end of the
    mid-M rule, the LayerNorm-fold and pooled-last-layer thresholds, odd tails behind full tiles — and every embedding of every batch
    is compared with the oracle's (computed once per image / text: the rows of a batch are independent)."""
    path = fixtures.cached_model(fixture_cache, config, ftype)
    clip, orc = gpu.Clip(path, device=0), ref.OracleModel(path)
    S = clip.vision_config["image_size"]
    T = (S // clip.vision_config["patch_size"]) ** 2 + 1
    imgs = fixtures.synthetic_images(300, S, seed=77)
    want = orc.image_batch_encode(imgs, normalize=True, mode=ref.MODE_FAITHFUL)
    sizes = [1, 2, 3, 4, 5, 7, 8, 12, 13, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 117, 118, 127, 128, 129, 194, 195, 200, 240, 241, 242, 255, 256, 257, 300]
    rows = set()
    for i, B in enumerate(sizes):
        lo = (i * 37) % (300 - B + 1)                                  # a different window of the 300 images each time
        got = clip.encode_images(imgs[lo:lo + B], normalize=True)
        d = one_minus_cos(got, want[lo:lo + B])
        assert got.shape == (B, want.shape[1]) and np.all(d <= TOL[ftype]), (config, ftype, B, float(d.max()), int(d.argmax()))
        rows.add(B * T)
    assert min(rows) <= 64 < max(rows) and (T < 17 or (any(2000 <= r <= 3300 for r in rows) and max(rows) > 4096))
    texts = fixtures.synthetic_token_ids(260, seed=9, min_len=1, max_len=75)
    want_t = np.stack([orc.text_encode(ids, normalize=True, mode=ref.MODE_FAITHFUL) for ids in texts])
    for i, n in enumerate([1, 2, 3, 5, 8, 16, 31, 50, 64, 65, 100, 128, 129, 200, 260]):
        lo = (i * 29) % (260 - n + 1)
        got = clip.encode_texts(texts[lo:lo + n], normalize=True)
        d = one_minus_cos(got, want_t[lo:lo + n])
        assert np.all(d <= TOL[ftype]), (config, ftype, n, float(d.max()), int(d.argmax()))
    clip.close()


@pytest.mark.parametrize("ftype", ["q4_0", "f16"])
def test_vit_b32_every_dispatch_boundary_gives_the_batch256_rows(gpu, fixture_cache, ftype):
    """The same sweep at the BASELINE model's widths (tile choice depends on N and K, not only on the row count): batches of 2 ... 255
    ViT-B/32 images (100 ... 12 750 token rows: ring tiles, the two-stream split at 2000-3300 rows, the 4096-row rule, 160 x 128 /
    192 x 128 / panel kernels) and 1 ... 200 ragged texts, each row against the same item inside the 256-item batch (1 - cos <= 1e-6:
    fp32 re-association between schedules only).  The anchors to the oracle are the 32-image test above and the bench's sample."""
    p = fixtures.cached_model(fixture_cache, "b32", ftype)
    clip = gpu.Clip(p, device=0)
    imgs = fixtures.synthetic_images(256, 224, seed=123)
    full = clip.encode_images(imgs)
    for i, B in enumerate([2, 3, 5, 8, 13, 20, 21, 39, 40, 41, 47, 48, 64, 66, 67, 81, 82, 83, 100, 128, 129, 200, 255]):
        lo = (i * 31) % (256 - B + 1)
        d = one_minus_cos(clip.encode_images(imgs[lo:lo + B]), full[lo:lo + B])
        assert np.all(d <= 1e-6), (ftype, B, float(d.max()), int(d.argmax()))
    texts = fixtures.synthetic_token_ids(256, seed=41, min_len=1, max_len=75)
    full_t = clip.encode_texts(texts)
    for i, n in enumerate([1, 2, 3, 4, 7, 20, 50, 64, 65, 100, 128, 200]):
        lo = (i * 23) % (256 - n + 1)
        d = one_minus_cos(clip.encode_texts(texts[lo:lo + n]), full_t[lo:lo + n])
        assert np.all(d <= 1e-6), (ftype, n, float(d.max()), int(d.argmax()))
    clip.close()


def test_graph_captures_survive_allocations_on_other_threads(gpu, fixture_cache):
    """HIP checks hipMalloc / hipFree against stream captures in flight — also those of other threads — and invalidates the capture
    (found with scripts/fuzz/run_gpu.sh; on ROCm 7.0 the stream then refuses every later launch).  Two cases: (a) a foreign thread of the
    application hammers hipMalloc / hipFree while this thread's text calls capture their launch chains: every call must succeed with the
    bits of the undisturbed call (eager relaunch on a replaced stream when a capture was lost); (b) two contexts on two threads, one
    capturing, one growing its workspace through this library (whose entry points run in the relaxed capture mode): same requirement."""
    import ctypes as C
    import threading
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0")
    clip = gpu.Clip(p, device=0)
    texts = [[49406] + [int(v) for v in np.random.default_rng(n).integers(1, 400, n)] + [49407] for n in range(1, 41)]
    want = [np.asarray(clip.encode_text(t), dtype=np.float32) for t in texts]           # first sighting of every shape: eager
    hip = C.CDLL("libamdhip64.so")
    stop = threading.Event()
    count = [0]

    def hammer():
        hip.hipSetDevice(0)
        ptr = C.c_void_p()
        while not stop.is_set():
            if hip.hipMalloc(C.byref(ptr), C.c_size_t(1 << 20)) == 0:
                hip.hipFree(ptr)
            count[0] += 1

    th = threading.Thread(target=hammer)
    th.start()
    try:
        for rep in range(3):                       # second sighting: capture (under the hammer); third: replay, or eager after a lost capture
            for t, w in zip(texts, want):
                assert np.array_equal(np.asarray(clip.encode_text(t), dtype=np.float32), w), (rep, len(t))
    finally:
        stop.set()
        th.join()
    assert count[0] > 0                            # (the hammer did run beside the calls; how often depends on the box)
    # (b) this library's own threads
    other = gpu.Clip(p, device=0)
    S = other.vision_config["image_size"]
    imgs = fixtures.synthetic_images(260, S, seed=5)
    base = other.encode_images(imgs[:4])
    texts2 = [[49406] + [int(v) for v in np.random.default_rng(100 + n).integers(1, 400, n)] + [49407] for n in range(1, 61)]
    want2 = [np.asarray(clip.encode_text(t), dtype=np.float32) for t in texts2]
    errs = []

    def grow():
        try:
            for B in (8, 20, 40, 64, 100, 130, 200, 260, 4):            # growing workspaces: hipMalloc / hipFree inside the library
                got = other.encode_images(imgs[:B])
                if not np.all(one_minus_cos(got[:4], base) <= 1e-6):
                    errs.append(("images", B))
        except Exception as e:                                           # noqa: BLE001 — reported below
            errs.append(repr(e))

    th = threading.Thread(target=grow)
    th.start()
    try:
        for rep in range(2):
            for t, w in zip(texts2, want2):
                assert np.array_equal(np.asarray(clip.encode_text(t), dtype=np.float32), w), (rep, len(t))
    finally:
        th.join()
    assert not errs, errs
    clip.close()
    other.close()


def test_an_invalidated_capture_falls_back_to_eager_launches_on_a_usable_stream(gpu, fixture_cache):
    """The recovery path itself, forced: CLIP_AMD_TEST_BREAK_CAPTURE=1 makes every capture contain a forbidden allocation (the state a
    colliding thread would leave).  The calls must still return the bits of the undisturbed context — eager relaunch, on a replaced
    stream if HIP left the old one unusable — and keep working afterwards, for both towers."""
    import subprocess
    import sys
    import os
    p = fixtures.cached_model(fixture_cache, "tiny", "q5_1")
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import clip_cpp_amd\n"
        "from oracle import fixtures\n"
        "c = clip_cpp_amd.Clip(%r, device=0, verbosity=1)\n"
        "imgs = fixtures.synthetic_images(3, c.vision_config['image_size'], seed=4)\n"
        "ids = [49406, 5, 6, 7, 8, 49407]\n"
        "out = []\n"
        "for rep in range(4):\n"
        "    out.append((c.encode_images(imgs), np.asarray(c.encode_text(ids), dtype=np.float32), c.encode_images(imgs[:1])))\n"
        "np.savez(sys.argv[1], **{'%%s%%d' %% (k, i): o[j] for i, o in enumerate(out) for j, k in enumerate('itx')})\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), p)
    res = {}
    for tag, env in (("plain", {}), ("broken", {"CLIP_AMD_TEST_BREAK_CAPTURE": "1"})):
        f = os.path.join(fixture_cache, "_break_capture_%s.npz" % tag)
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, errors="replace", timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        res[tag] = (np.load(f), r.stderr)
    plain, broken = res["plain"][0], res["broken"][0]
    for k in plain.files:
        assert np.array_equal(plain[k], broken[k]), k
        assert np.array_equal(plain[k], plain[k[0] + "0"])                 # eager (first), captured and replayed calls agree bit for bit
    assert "stream replaced" not in res["plain"][1]


def test_no_kernel_stores_past_the_end_of_its_workspace_buffer(gpu, fixture_cache):
    """CLIP_AMD_GUARD=1: canary blocks behind every buffer of the activation workspaces, verified after every forward.  Batch sizes and text
    lists across the dispatch boundaries on the small models, at ViT-B/32 width (q4_0, f16; 700 images: the 256 x 256 kernel) and on 1 ... 256 ViT-L/14
    images (65 792 rows = 257 tile rows of that kernel) — masked tile tails, split-K seams, the pooled last layer, the two-stream split with its sibling
    workspace.  A violation fails the call."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import clip_cpp_amd\n"
        "from oracle import fixtures\n"
        "cache = %r\n"
        "jobs = [('tiny', 'q4_0', [1, 3, 4, 17, 64, 65, 118, 194, 241, 300], [1, 2, 7, 64, 65, 260]),\n"
        "        ('tiny14', 'f16', [1, 13, 64, 129, 300], [1, 5, 100]),\n"
        "        ('b32', 'q4_0', [1, 2, 5, 21, 40, 48, 66, 67, 82, 129, 256, 700], [1, 3, 20, 64, 128, 256]),\n"
        "        ('b32', 'f16', [1, 8, 41, 83, 200], [2, 50, 200]),\n"
        "        ('l14', 'f16', [1, 2, 9, 33, 130, 256], [1, 40])]\n"
        "for cfg, ft, Bs, Ns in jobs:\n"
        "    c = clip_cpp_amd.Clip(fixtures.cached_model(cache, cfg, ft), device=0)\n"
        "    S = c.vision_config['image_size']\n"
        "    imgs = fixtures.synthetic_images(max(Bs), S, seed=3)\n"
        "    texts = fixtures.synthetic_token_ids(max(Ns), seed=4, min_len=1, max_len=75)\n"
        "    import zlib\n"
        "    crc = 0\n"
        "    for B in Bs:\n"
        "        e = c.encode_images(imgs[:B]); assert np.all(np.isfinite(e)), (cfg, ft, B); crc = zlib.crc32(e.tobytes(), crc)\n"
        "    for n in Ns:\n"
        "        e = c.encode_texts(texts[:n]); assert np.all(np.isfinite(e)), (cfg, ft, n); crc = zlib.crc32(e.tobytes(), crc)\n"
        "    print('CRC', cfg, ft, crc)\n"
        "    c.close()\n"
        "print('GUARD-OK')\n"
    ) % (root, fixture_cache)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, errors="replace", timeout=900, env=dict(os.environ, CLIP_AMD_GUARD="1"))
    assert r.returncode == 0 and "GUARD-OK" in r.stdout and "CLIP_AMD_GUARD" not in r.stderr, r.stdout[-1500:] + r.stderr[-3000:]
    # the workspace starts as NaN patterns in that mode: bytes nobody wrote must not reach an embedding — same bits as the normal run
    plain = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, errors="replace", timeout=900, env=dict(os.environ, CLIP_AMD_GUARD="0"))
    crcs = [l for l in r.stdout.splitlines() if l.startswith("CRC")]
    assert plain.returncode == 0 and len(crcs) == 5 and crcs == [l for l in plain.stdout.splitlines() if l.startswith("CRC")], (crcs, plain.stdout[-800:])
    # the checker's own test: a stray byte planted behind the last buffer must fail the call and name the offset
    probe = ("import sys\nsys.path.insert(0, %r)\nimport clip_cpp_amd\nfrom oracle import fixtures\n"
             "c = clip_cpp_amd.Clip(fixtures.cached_model(%r, 'tiny', 'q4_0'), device=0)\n"
             "try:\n    c.encode_images(fixtures.synthetic_images(2, c.vision_config['image_size'], seed=1))\n    print('NOT-CAUGHT')\n"
             "except RuntimeError:\n    print('CAUGHT')\n") % (root, fixture_cache)
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, errors="replace", timeout=300,
                       env=dict(os.environ, CLIP_AMD_GUARD="1", CLIP_AMD_GUARD_SELFTEST="1"))
    assert "CAUGHT" in r.stdout and "NOT-CAUGHT" not in r.stdout and "written 100 bytes past its end" in r.stderr, r.stdout[-500:] + r.stderr[-1500:]


@pytest.mark.parametrize("ftype", ["f32", "f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_gpu_against_the_reference_source_itself(gpu, fixture_cache, ftype):
    """The HIP path against /root/reference/clip.cpp ITSELF (compiled unchanged over oracle/ggml_shim: oracle/_ref/libclip_ref.so, built in the
    dev container, travels with the snapshot; tests/test_reference_graph.py shows it equals the oracle bit for bit): a two-tower file with the
    base model's tensor count, which the reference's loader accepts; ViT-B/32 q4_0 at full size for the q4_0 case."""
    from oracle import ref_graph
    if not ref_graph.available():
        pytest.skip("oracle/_ref/libclip_ref.so not built (needs the reference tree)")
    jobs = [("base12", 32, 5, TOL)] + ([("b32", 224, 2, TOL_MODEL)] if ftype == "q4_0" else [])
    for cfg, S, B, tol in jobs:
        p = fixtures.cached_model(fixture_cache, cfg, ftype)
        R, clip = ref_graph.ReferenceModel(p), gpu.Clip(p, device=0)
        imgs = fixtures.synthetic_images(B, S, seed=31)
        d = one_minus_cos(clip.encode_images(imgs), R.image_batch_encode(imgs))
        assert np.all(d <= tol[ftype]), (cfg, ftype, float(d.max()))
        for text in ("a photo of a cat", "two dogs, a red apple & the sea!"):
            ids = R.tokenize(text)
            assert list(clip.tokenize(text)) == ids
            d = one_minus_cos(np.asarray(clip.encode_text(ids), dtype=np.float32), R.text_encode(ids))
            assert d <= (TOL_MODEL_TEXT if cfg == "b32" else tol)[ftype], (cfg, ftype, text, float(d))
        R.close()
        clip.close()


def test_baseline_batch_every_row_against_the_oracle(gpu, fixture_cache):
    """The BASELINE batch itself — 256 ViT-B/32 q4_0 images and 256 ragged texts in one call each, the shapes bench.py times — with EVERY
    embedding compared with the oracle in ggml-faithful numerics (not a sample, not self-consistency): the oracle runs the 256 images in chunks
    on all host cores (rows are independent)."""
    p = fixtures.cached_model(fixture_cache, "b32", "q4_0")
    clip, orc = gpu.Clip(p, device=0), ref.OracleModel(p)
    imgs = fixtures.synthetic_images(256, 224, seed=256)
    got = clip.encode_images(imgs)
    want = np.concatenate([orc.image_batch_encode(imgs[i:i + 32], mode=ref.MODE_FAITHFUL) for i in range(0, 256, 32)])
    d = one_minus_cos(got, want)
    assert d.shape == (256,) and np.all(d <= TOL_MODEL["q4_0"]), (float(d.max()), int(d.argmax()))
    texts = fixtures.synthetic_token_ids(256, seed=257, min_len=1, max_len=75)
    got_t = clip.encode_texts(texts)
    want_t = np.stack([orc.text_encode(t, mode=ref.MODE_FAITHFUL) for t in texts])
    dt = one_minus_cos(got_t, want_t)
    assert np.all(dt <= TOL_MODEL_TEXT["q4_0"]), (float(dt.max()), int(dt.argmax()))
    print("batch 256: images 1-cos max %.3g mean %.3g; texts max %.3g mean %.3g" % (d.max(), d.mean(), dt.max(), dt.mean()))
