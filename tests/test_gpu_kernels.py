"""GPU tier, kernel level: every HIP kernel of the hot path against the oracle / numpy on seeded inputs,
called through the C ABI test hooks (include/clip_amd.h).  All tests need a real MI355X."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref

pytestmark = pytest.mark.gpu

TYPES = ["f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "f32"]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _h(x):
    """round through fp16 (what the GPU activations are)."""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def run_gemm(L, tid, raw, N, K, X, bias=None, resid=None, epi=0, tile=0):
    M = X.shape[0]
    y = np.full((M, N), np.nan, dtype=np.float32)
    rc = L.clip_amd_test_gemm(tid, raw.ctypes.data_as(C.c_void_p), N, K, _fp(X), M,
                              _fp(bias) if bias is not None else None, _fp(resid) if resid is not None else None, _fp(y), epi, tile)
    assert rc == 0, "clip_amd_test_gemm rc=%d" % rc
    return y


def gelu_tanh(x):
    x = x.astype(np.float64)
    return 0.5 * x * (1 + np.tanh(0.7978845608028654 * x * (1 + 0.044715 * x * x)))


def gelu_quick(x):
    x = x.astype(np.float64)
    return x / (1 + np.exp(-1.702 * x))


@pytest.fixture(scope="module")
def L(clip_lib):
    if clip_lib.device_count() < 1:
        pytest.fail("GPU tier needs a HIP device: the product has no CPU fallback")
    return clip_lib.lib()


def _weights(rng, N, K):
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    W[:, rng.integers(0, K, size=max(1, K // 128))] *= 8.0   # outlier columns
    return W


@pytest.mark.parametrize("tname", TYPES)
@pytest.mark.parametrize("shape", [(50, 64, 64), (77, 96, 128), (130, 192, 256), (257, 768, 768), (16, 512, 3072)])
def test_gemm_all_weight_types_vs_dequant_reference(L, tname, shape):
    """Y = X.W^T + b for every weight format: exact up to the fp16 rounding of the dequantised weight
    (<= 2^-11 relative per product) -> rigorous elementwise bound 1e-3 * (|X|.|W|^T)."""
    M, N, K = shape
    rng = np.random.default_rng(hash((tname, shape)) % (2 ** 31))
    tid = ref.GGML_TYPES[tname]
    W = _weights(rng, N, K)
    raw = ref.quantize(tid, W)
    Wd = ref.dequantize(tid, raw, N, K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    y = run_gemm(L, tid, raw, N, K, X, bias=bias, epi=0)
    Xh = _h(X)
    want = Xh.astype(np.float64) @ Wd.astype(np.float64).T + bias
    bound = 1.0e-3 * (np.abs(Xh).astype(np.float64) @ np.abs(Wd).astype(np.float64).T) + 1e-5
    err = np.abs(y - want)
    assert np.all(np.isfinite(y))
    bad = np.argwhere(err > bound)
    assert bad.size == 0, "%d/%d bad, first %s got %g want %g; max err %g" % (
        len(bad), y.size, bad[0], y[tuple(bad[0])], want[tuple(bad[0])], err.max())
    # and close (statistically) to the ggml-numerics oracle, whose own noise is the q8 activation quantisation
    yo = ref.mul_mat(tid, raw, N, K, X, ref.MODE_FAITHFUL) + bias
    rel = np.linalg.norm(y - yo) / np.linalg.norm(yo)
    assert rel < (5e-4 if tname in ("f16", "f32") else 2e-2), rel


@pytest.mark.parametrize("tile", [64064, 64128, 128064, 128128, 160128, 192128])
@pytest.mark.parametrize("tname", ["f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_gemm_tiles_are_bitwise_identical(L, tile, tname):
    """Every tile shape accumulates each output in the same k order -> identical bits; also exercises M/N edges."""
    rng = np.random.default_rng(42)
    M, N, K = 203, 320, 192
    tid = ref.GGML_TYPES[tname]
    raw = ref.quantize(tid, _weights(rng, N, K))
    X = rng.standard_normal((M, K)).astype(np.float32)
    base = run_gemm(L, tid, raw, N, K, X, epi=0, tile=64064)
    y = run_gemm(L, tid, raw, N, K, X, epi=0, tile=tile)
    assert np.array_equal(base, y)


@pytest.mark.parametrize("ksplit,tile", [(2, 64064), (3, 64064), (5, 64128), (12, 64064)])
@pytest.mark.parametrize("tname,epi", [("q4_0", 4), ("f16", 1), ("q8_0", 3), ("q5_1", 0)])
def test_gemm_split_k_is_deterministic_and_matches_unsplit(L, ksplit, tile, tname, epi):
    """Small-M path: K split over `ksplit` workgroups per tile with the in-kernel ordered fix-up.  Two runs give the
    same bits (summation order does not depend on arrival order); the result differs from the unsplit one only by
    fp32 re-association."""
    rng = np.random.default_rng(100 + ksplit)
    M, N, K = 50, 320, 1536
    tid = ref.GGML_TYPES[tname]
    raw = ref.quantize(tid, _weights(rng, N, K))
    X = rng.standard_normal((M, K)).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.5).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    code = ksplit * 1000000 + tile
    a = run_gemm(L, tid, raw, N, K, X, bias=bias, resid=resid, epi=epi, tile=code)
    b = run_gemm(L, tid, raw, N, K, X, bias=bias, resid=resid, epi=epi, tile=code)
    assert np.array_equal(a, b)
    base = run_gemm(L, tid, raw, N, K, X, bias=bias, resid=resid, epi=epi, tile=1000000 + tile)
    assert np.abs(a - base).max() <= 2e-3 * max(1.0, np.abs(base).max())
    auto = run_gemm(L, tid, raw, N, K, X, bias=bias, resid=resid, epi=epi, tile=0)     # heuristic (splits here)
    assert np.abs(auto - base).max() <= 2e-3 * max(1.0, np.abs(base).max())


@pytest.mark.parametrize("epi", [1, 2, 3, 4])
def test_gemm_epilogues(L, epi):
    rng = np.random.default_rng(7 + epi)
    M, N, K = 100, 256, 128
    tid = ref.GGML_TYPES["q4_0"]
    raw = ref.quantize(tid, _weights(rng, N, K) * 4)
    Wd = ref.dequantize(tid, raw, N, K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.5).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    y = run_gemm(L, tid, raw, N, K, X, bias=bias, resid=resid, epi=epi)
    lin = _h(X).astype(np.float64) @ Wd.astype(np.float64).T + bias
    if epi == 1:
        want, tol = lin, 2e-3
    elif epi == 2:
        want, tol = gelu_tanh(lin), 2e-3
    elif epi == 3:
        want, tol = gelu_quick(lin), 2e-3
    else:
        want, tol = lin + resid, 1e-3
    np.testing.assert_allclose(y, want, atol=tol * max(1.0, np.abs(want).max()), rtol=2e-3)


def test_gemm_detects_transposes(L):
    """A = I style check with an ASYMMETRIC weight: catches row/col swaps of the MFMA C layout."""
    N = K = 64
    W = np.zeros((N, K), dtype=np.float32)
    for n in range(N):
        W[n, (n * 7 + 3) % K] = 1.0 + n / 64.0
    raw = ref.quantize(1, W)
    X = np.arange(64 * K, dtype=np.float32).reshape(64, K) % 13 - 6
    y = run_gemm(L, 1, raw, N, K, X, epi=0)
    want = X @ _h(W).T
    np.testing.assert_allclose(y, want, atol=1e-3)


@pytest.mark.parametrize("h", [64, 512, 768, 1024, 1280])
def test_layernorm_vs_oracle(L, h):
    rng = np.random.default_rng(h)
    rows = 37
    x = (rng.standard_normal((rows, h)) * 3 + 0.5).astype(np.float32)
    w = (1 + rng.standard_normal(h) * 0.05).astype(np.float32)
    b = (rng.standard_normal(h) * 0.05).astype(np.float32)
    y = np.empty_like(x)
    assert L.clip_amd_test_layernorm(_fp(x), _fp(w), _fp(b), 1e-5, rows, h, _fp(y), 0) == 0
    want = ref.layer_norm(x, w, b, 1e-5)
    np.testing.assert_allclose(y, want, atol=2e-5, rtol=1e-5)
    y16 = np.empty_like(x)
    assert L.clip_amd_test_layernorm(_fp(x), _fp(w), _fp(b), 1e-5, rows, h, _fp(y16), 1) == 0
    np.testing.assert_allclose(y16, _h(want), atol=4e-3, rtol=1e-3)


def attention_ref(qkv, nseq, T, h, nh, causal):
    qkv = _h(qkv).astype(np.float64).reshape(nseq, T, 3, nh, h // nh)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    s = np.einsum("bqhd,bkhd->bhqk", q, k)
    if causal:
        mask = np.triu(np.ones((T, T), dtype=bool), 1)
        s = np.where(mask, -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    o = np.einsum("bhqk,bkhd->bqhd", p, v)
    return o.reshape(nseq * T, h)


@pytest.mark.parametrize("cfg", [(3, 17, 64, 2, 0), (2, 50, 768, 12, 0), (2, 257, 1024, 16, 0), (1, 257, 1280, 16, 0),
                                 (5, 9, 512, 8, 1), (2, 77, 512, 8, 1), (1, 1, 64, 2, 1), (1, 64, 128, 2, 0), (1, 288, 128, 2, 1),
                                 (2, 577, 1024, 16, 0), (1, 300, 128, 2, 0), (1, 592, 64, 1, 1)])   # 336-px ViT-L/14: T = 577
def test_attention_vs_reference(L, cfg):
    nseq, T, h, nh, causal = cfg
    rng = np.random.default_rng(sum(cfg))
    qkv = (rng.standard_normal((nseq * T, 3 * h)) * 0.7).astype(np.float32)
    qkv[:, :h] *= 1.0 / np.sqrt(h // nh)   # q arrives pre-scaled
    out = np.full((nseq * T, h), np.nan, dtype=np.float32)
    rc = L.clip_amd_test_attention(_fp(qkv), nseq, T, h, nh, causal, _fp(out))
    assert rc == 0, rc
    want = attention_ref(qkv, nseq, T, h, nh, causal)
    assert np.all(np.isfinite(out))
    err = np.abs(out - want).max()
    assert err < 4e-3, (cfg, err)
