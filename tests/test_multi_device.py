"""Multi-GPU form of clip_image_batch_encode behind the C ABI (clip_amd_model_load_multi): shard arithmetic on CPU;
on the GPU tier the sharded path itself — replicas, per-replica host threads, padding of the last shard — runs on ONE device with
over-subscribed replicas (the collective is replaced by per-replica copies there: RCCL refuses two ranks on one device), and with
the real ncclAllGather when >= 2 devices are visible."""
import os

import numpy as np
import pytest

from oracle import fixtures


def test_shard_bounds_are_contiguous_ceil_shards(clip_lib):
    for total in (0, 1, 5, 8, 13, 256, 1000, 1024, 1023):
        for G in (1, 2, 3, 4, 8):
            spans = [clip_lib.shard_bounds(total, G, g) for g in range(G)]
            per = -(-total // G) if total else 0
            assert all(s[2] == per for s in spans)
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            assert all(0 <= hi - lo <= per for lo, hi, _ in spans)
            # only trailing shards are short: every shard before the first short one is full
            sizes = [hi - lo for lo, hi, _ in spans]
            first_short = next((i for i, s in enumerate(sizes) if s < per), G)
            assert all(s == per for s in sizes[:first_short]) and all(s == 0 for s in sizes[first_short + 1:])
    assert clip_lib.shard_bounds(10, 0, 0) == (0, 0, 0) and clip_lib.shard_bounds(10, 4, 7) == (0, 0, 0)   # invalid arguments


@pytest.mark.gpu
@pytest.mark.parametrize("G,B", [(1, 7), (2, 7), (3, 8), (4, 5), (2, 300)])
def test_sharded_batch_encode_matches_single_context(clip_lib, fixture_cache, G, B, monkeypatch):
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device")
    if clip_lib.device_count() < G:
        monkeypatch.setenv("CLIP_AMD_MULTI_OVERSUBSCRIBE", "1")
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=True, vision=True)
    single = clip_lib.Clip(p, device=0)
    multi = clip_lib.Clip(p, n_devices=G)
    assert multi.n_devices == G and single.n_devices == 1
    imgs = fixtures.synthetic_images(B, 32, seed=5)
    want = single.encode_images(imgs)
    got = multi.encode_images(imgs)                         # B >= 2 G: sharded, ceil(B / G) per replica, last shard padded
    # bit for bit where every shard runs the kernels of the unsharded batch (> 64 token rows = 4+ images of 17 tokens: the
    # LayerNorm-folded chain, whose results do not depend on tile shapes); smaller shards take the small-M kernels, which round the
    # normalised activations instead of x * gamma: equal to fp16 rounding
    per = -(-B // G)
    if per >= 4 and B - (G - 1) * per >= 4 and B >= 4:
        assert np.array_equal(got, want)
    else:
        a, b = got / np.linalg.norm(got, axis=1, keepdims=True), want / np.linalg.norm(want, axis=1, keepdims=True)
        assert np.all(1.0 - (a * b).sum(1) <= 1e-6)
        np.testing.assert_allclose(got, want, atol=1e-3)
    one = multi.encode_images(imgs[:1])                      # tiny batch: device 0 only
    np.testing.assert_allclose(one, single.encode_images(imgs[:1]), atol=0, rtol=0)
    # everything else of the API keeps working on the primary context
    ids = [49406, 5, 6, 7, 49407]
    assert np.array_equal(np.asarray(multi.encode_text(ids), np.float32), np.asarray(single.encode_text(ids), np.float32))
    multi.close()
    single.close()


@pytest.mark.gpu
def test_rccl_binding_runs_on_one_device(clip_lib, fixture_cache, monkeypatch):
    """The RCCL calls themselves — dlopen of librccl, ncclCommInitAll, grouped ncclAllGather with ncclFloat32, ncclCommDestroy — on ONE
    device (CLIP_AMD_MULTI_FORCE_RCCL=1: a single replica still goes through the collective): the binding, the enum value and the call
    sequence of the multi-GPU path execute on hardware even where a second GPU is not available."""
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device")
    import ctypes as C
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("CLIP_AMD_MULTI_FORCE_RCCL", "1")
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=False, vision=True)
    single = clip_lib.Clip(p, device=0)
    multi = clip_lib.Clip(p, n_devices=1)
    for B in (2, 37, 300):
        imgs = fixtures.synthetic_images(B, 32, seed=B)
        want = single.encode_images(imgs)
        assert np.array_equal(multi.encode_images(imgs), want)
        ptr = clip_lib.lib().clip_amd_gathered_embeddings(multi.ctx, 0)        # the all-gather's receive buffer
        assert ptr
        t = torch.empty((B, 32), dtype=torch.float32, device="cuda:0")
        assert C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), B * 32 * 4, 3) == 0
        assert np.array_equal(t.cpu().numpy(), want)
    multi.close()
    single.close()


@pytest.mark.gpu
def test_pair_call_through_rccl_on_one_device(clip_lib, fixture_cache, monkeypatch):
    """clip_amd_encode_pair_device_multi with the REAL collective on one device (CLIP_AMD_MULTI_FORCE_RCCL=1: one replica, ncclCommInitAll,
    a one-rank grouped ncclAllGather of the [images | texts] block): both towers on two streams (replica + sibling context), the gathered
    buffer holds the image rows then the text rows, the host copies are the single-context results bit for bit."""
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device")
    import ctypes as C
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("CLIP_AMD_MULTI_FORCE_RCCL", "1")
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=True, vision=True)
    single = clip_lib.Clip(p, device=0)
    multi = clip_lib.Clip(p, n_devices=1)
    B, NT = 23, 31
    imgs = fixtures.synthetic_images(B, 32, seed=8)
    texts = fixtures.synthetic_token_ids(NT, seed=4, min_len=3, max_len=40)
    want_i, want_t = single.encode_images(imgs), single.encode_texts(texts)
    d_img = torch.from_numpy(imgs).cuda()
    d_ids = torch.from_numpy(np.concatenate(texts).astype(np.int32)).cuda()
    offs = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    torch.cuda.synchronize()
    for _ in range(3):
        out_i, out_t = np.zeros((B, 32), np.float32), np.zeros((NT, 32), np.float32)
        multi.encode_pair_device_multi([d_img.data_ptr()], B, [d_ids.data_ptr()], offs, True, out_i, out_t)
        assert np.array_equal(out_i, want_i) and np.array_equal(out_t, want_t)
    ptr = clip_lib.lib().clip_amd_gathered_embeddings(multi.ctx, 0)
    t = torch.empty((B + NT, 32), dtype=torch.float32, device="cuda:0")
    assert C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), t.numel() * 4, 3) == 0
    assert np.array_equal(t.cpu().numpy(), np.concatenate([want_i, want_t]))
    multi.close()
    single.close()


@pytest.mark.gpu
def test_split_graph_is_not_replayed_while_the_sibling_carries_the_text_tower(clip_lib, fixture_cache, monkeypatch):
    """a vision hipGraph captured with the batch split over the sibling context holds kernel nodes that write the
    sibling's workspace.  clip_amd_encode_pair_device_multi runs the TEXT tower on that sibling; when it hits the same (batch, in, out)
    key — the replica's send buffer is stable — the graph must not be replayed (the eager path runs unsplit while sibling_busy).
    Sequence pair, images, images (capture, split), images (replay), then pairs: every result bit-identical to single-context calls."""
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device")
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("CLIP_AMD_MULTI_FORCE_RCCL", "1")
    p = fixtures.cached_model(fixture_cache, "tiny14", "f16", text=True, vision=True)
    S, proj = fixtures.CONFIGS["tiny14"]["v"]["S"], fixtures.CONFIGS["tiny14"]["v"]["proj"]
    B, NT = 12, 200
    imgs = fixtures.synthetic_images(B, S, seed=77)
    texts = fixtures.synthetic_token_ids(NT, seed=78, min_len=20, max_len=60)
    monkeypatch.setenv("CLIP_AMD_SPLIT", "0,0")
    single = clip_lib.Clip(p, device=0)
    n1 = (B + 1) // 2
    halves = np.concatenate([single.encode_images(imgs[:n1]), single.encode_images(imgs[n1:])])     # what a split call gives, bit for bit
    whole = single.encode_images(imgs)
    want_t = single.encode_texts(texts)
    single.close()
    monkeypatch.setenv("CLIP_AMD_SPLIT", "2,64")
    multi = clip_lib.Clip(p, n_devices=1)
    d_img = torch.from_numpy(imgs).cuda()
    d_ids = torch.from_numpy(np.concatenate(texts).astype(np.int32)).cuda()
    offs = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    torch.cuda.synchronize()

    def pair():
        oi, ot = np.zeros((B, proj), np.float32), np.zeros((NT, proj), np.float32)
        multi.encode_pair_device_multi([d_img.data_ptr()], B, [d_ids.data_ptr()], offs, True, oi, ot)
        return oi, ot

    def images():
        return multi.encode_images_device_multi([d_img.data_ptr()], B, True, np.zeros((B, proj), np.float32))

    oi, ot = pair()                                        # first sighting of the key, sibling busy: eager, unsplit
    assert np.array_equal(oi, whole) and np.array_equal(ot, want_t)
    for i in range(4):                                     # eager split, capture (split), replays
        assert np.array_equal(images(), halves), i
    for i in range(6):                                     # the captured split graph must stay out of the sibling's way
        oi, ot = pair()
        assert np.array_equal(oi, whole), ("pair images", i)
        assert np.array_equal(ot, want_t), ("pair texts", i)
        assert np.array_equal(images(), halves), ("images after pair", i)
    multi.close()
    monkeypatch.delenv("CLIP_AMD_SPLIT", raising=False)


@pytest.mark.gpu
def test_rccl_all_gather_path_with_two_devices(clip_lib, fixture_cache):
    if clip_lib.device_count() < 2:
        pytest.skip("needs >= 2 visible HIP devices (the 1-GPU box runs the over-subscribed form above)")
    import ctypes as C
    torch = pytest.importorskip("torch")
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=False, vision=True)
    G = min(clip_lib.device_count(), 8)
    single = clip_lib.Clip(p, device=0)
    multi = clip_lib.Clip(p, n_devices=G)
    for B in (2 * G, 5 * G + 3, 1024):
        imgs = fixtures.synthetic_images(B, 32, seed=B)
        want = single.encode_images(imgs)
        assert np.array_equal(multi.encode_images(imgs), want)
        lo, hi, per = clip_lib.shard_bounds(B, G, 0)
        # the gathered [G * per][proj] buffer on EVERY device holds the whole result (one ncclAllGather)
        for g in range(G):
            ptr = clip_lib.lib().clip_amd_gathered_embeddings(multi.ctx, g)
            assert ptr
            with torch.cuda.device(g):
                t = torch.empty((B, 32), dtype=torch.float32, device="cuda:%d" % g)
                C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), B * 32 * 4, 3)
                assert np.array_equal(t.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("G", [1, 2, 3])
def test_sharded_u8_text_and_device_resident_entry_points(clip_lib, fixture_cache, G, monkeypatch):
    """besides the f32 host path, clip_amd_image_batch_encode_u8 and clip_text_batch_encode shard on a
    clip_amd_model_load_multi handle, and the device-resident forms (what bench.py --single-process measures) take per-device shard
    pointers.  Replicas share device 0 where fewer than G devices exist (per-replica copies replace the collective there); with G
    devices the grouped ncclAllGather runs and the gathered buffer is checked on every device."""
    torch = pytest.importorskip("torch")
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device")
    if clip_lib.device_count() < G:
        monkeypatch.setenv("CLIP_AMD_MULTI_OVERSUBSCRIBE", "1")
    ndev = clip_lib.device_count()
    p = fixtures.cached_model(fixture_cache, "tiny", "q4_0", text=True, vision=True)
    single = clip_lib.Clip(p, device=0)
    multi = clip_lib.Clip(p, n_devices=G)
    rng = np.random.default_rng(9)
    # raw u8 images of mixed sizes: 16 per call -> shards of ceil(16 / G); every shard has >= 4 images (68+ rows: the folded chain everywhere)
    raws = [rng.integers(0, 256, size=(int(rng.integers(30, 90)), int(rng.integers(30, 90)), 3), dtype=np.uint8) for _ in range(16)]
    assert np.array_equal(multi.encode_images_u8(raws), single.encode_images_u8(raws))
    # ragged texts: 40 texts of 20-30 tokens: > 64 rows per shard
    texts = fixtures.synthetic_token_ids(40, seed=3, min_len=20, max_len=30)
    assert np.array_equal(multi.encode_texts(texts), single.encode_texts(texts))
    # device-resident shards
    B, S = 21, 32
    imgs = fixtures.synthetic_images(B, S, seed=6)
    want = single.encode_images(imgs)
    keep, ptrs = [], []
    for g in range(G):
        lo, hi, per = clip_lib.shard_bounds(B, G, g)
        t = torch.from_numpy(imgs[lo:hi].copy()).to("cuda:%d" % (g % ndev))
        keep.append(t)
        ptrs.append(t.data_ptr())
    torch.cuda.synchronize()
    out = np.empty((B, 32), dtype=np.float32)
    multi.encode_images_device_multi(ptrs, B, True, out)
    assert np.array_equal(out, want)
    flat = [np.concatenate(texts[clip_lib.shard_bounds(40, G, g)[0]:clip_lib.shard_bounds(40, G, g)[1]]).astype(np.int32) for g in range(G)]
    offs = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    tk = [torch.from_numpy(f).to("cuda:%d" % (g % ndev)) for g, f in enumerate(flat)]
    torch.cuda.synchronize()
    out_t = np.empty((40, 32), dtype=np.float32)
    multi.encode_texts_device_multi([t.data_ptr() for t in tk], offs, True, out_t)
    assert np.array_equal(out_t, single.encode_texts(texts))
    # both towers of a step in one call: vision on the replica stream, text on the twin context's stream, one all-gather (bench.py --single-process)
    for rep in range(2):     # second round: the twin contexts exist, buffers are warm
        out_i2, out_t2 = np.zeros((B, 32), dtype=np.float32), np.zeros((40, 32), dtype=np.float32)
        multi.encode_pair_device_multi(ptrs, B, [t.data_ptr() for t in tk], offs, True, out_i2, out_t2)
        assert np.array_equal(out_i2, want) and np.array_equal(out_t2, out_t)
    if ndev >= G and G > 1 or os.environ.get("CLIP_AMD_MULTI_FORCE_RCCL") == "1":
        # gathered layout of the pair call on every device: G blocks of (rows_per_device(B) image rows, rows_per_device(40) text rows)
        import ctypes as C
        per_i, per_t = clip_lib.shard_bounds(B, G, 0)[2], clip_lib.shard_bounds(40, G, 0)[2]
        for g in range(G):
            ptr = clip_lib.lib().clip_amd_gathered_embeddings(multi.ctx, g)
            with torch.cuda.device(g):
                t = torch.empty((G * (per_i + per_t), 32), dtype=torch.float32, device="cuda:%d" % g)
                assert C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), t.numel() * 4, 3) == 0
                blk = t.cpu().numpy().reshape(G, per_i + per_t, 32)
            for r in range(G):
                lo, hi, _ = clip_lib.shard_bounds(B, G, r)
                lt, ht, _ = clip_lib.shard_bounds(40, G, r)
                assert np.array_equal(blk[r, :hi - lo], want[lo:hi]) and np.array_equal(blk[r, per_i:per_i + ht - lt], out_t[lt:ht])
    multi.close()
    single.close()
