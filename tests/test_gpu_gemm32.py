"""GPU tier: the 32 x 32 x 16 GEMM kernel (clip_cpp_amd/csrc/k_gemm32.hip; tile codes 256261 / 320261) — the fp16-output weight GEMMs
(q/k/v, FFN-up; reference clip.cpp:1360-1380, 1407-1413) of a large batch.  Its MFMA instruction sums k in another order than the
16 x 16 x 32 kernels, so it is held to the float64 product of the operands it multiplies (rigorous elementwise bound) and to the other
kernels within fp16 output rounding — not bit for bit."""
import numpy as np
import pytest

from oracle import ref
from test_gpu_kernels import (L, _diff_report, _h, _lnfold_case, _lnfold_reference, _weights, gelu_quick, gelu_tanh, run_gemm_ex,  # noqa: F401
                              run_lnfold)

pytestmark = pytest.mark.gpu

TILES32 = [256261, 320261]


def _want(Xh, Wd, bias, epi, qcols, qscale):
    lin = Xh.astype(np.float64) @ Wd.astype(np.float64).T + bias
    want = {1: lin, 2: gelu_tanh(lin), 3: gelu_quick(lin)}[epi].copy()
    want[:, :qcols] *= qscale
    # fp16 rounding of the dequantised weight (2^-11 per product), activations 1.13-Lipschitz, + the output's own fp16 rounding
    bound = 1.2 * 1.0e-3 * (np.abs(Xh).astype(np.float64) @ np.abs(Wd).astype(np.float64).T) + np.abs(want) * 2.0 ** -10 + 1e-4
    return want, bound


@pytest.mark.parametrize("tile", TILES32)
@pytest.mark.parametrize("tname", ["f16", "q4_0", "q5_1", "q8_0"])
@pytest.mark.parametrize("M,N,K,epi", [(203, 320, 192, 1), (700, 448, 128, 3), (1000, 768, 448, 2), (333, 64, 256, 1), (1600, 2304, 768, 1),
                                       (641, 3072, 768, 3), (77, 1536, 512, 2)])
def test_gemm32_vs_float64_and_the_other_kernels(L, tile, tname, M, N, K, epi):
    """M / N edges (rows past M clamped and never stored, column slabs past N skipped), every ring length class (K-tiles = 4, 6, 8, 14,
    16, 24: T % 4 in {0, 2}), the three fp16 epilogues incl. the Q-scale columns."""
    rng = np.random.default_rng(hash((tile, tname, M, N, K)) % (2 ** 31))
    tid = ref.GGML_TYPES[tname]
    raw = ref.quantize(tid, _weights(rng, N, K))
    Wd = ref.dequantize(tid, raw, N, K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    qc, qs = ((N // 3) // 64 * 64, 0.125) if epi == 1 else (0, 1.0)
    y = run_gemm_ex(L, tid, raw, N, K, X, bias=bias, epi=epi, tile=tile, qcols=qc, qscale=qs)
    assert np.all(np.isfinite(y))
    want, bound = _want(_h(X), Wd, bias, epi, qc, qs)
    err = np.abs(y - want)
    bad = np.argwhere(err > bound)
    assert bad.size == 0, "%d/%d bad, first %s got %g want %g bound %g" % (len(bad), y.size, bad[0], y[tuple(bad[0])], want[tuple(bad[0])], bound[tuple(bad[0])])
    # against the 16 x 16 x 32 kernel: same operands, f32 sums in another order -> at most one fp16 ulp apart after the output rounding
    y0 = run_gemm_ex(L, tid, raw, N, K, X, bias=bias, epi=epi, tile=128128, qcols=qc, qscale=qs)
    ulp = np.abs(y - y0) / (np.maximum(np.abs(y0), 2.0 ** -14) * 2.0 ** -10)
    assert ulp.max() <= 1.01, "max %.2f ulp at %s" % (ulp.max(), np.unravel_index(ulp.argmax(), ulp.shape))
    assert (ulp > 0).mean() < 0.02                       # ... and almost everywhere identical


@pytest.mark.parametrize("tile", TILES32)
def test_gemm32_detects_transposes_and_fragment_layout(L, tile):
    """asymmetric operands: X = one-hot rows shifted by the row index, W = distinct integers -> every output element names its (m, n)."""
    M, N, K = 320, 256, 128
    X = np.zeros((M, K), dtype=np.float32)
    X[np.arange(M), np.arange(M) % K] = 1.0
    X[np.arange(M), (np.arange(M) * 7 + 3) % K] += 2.0
    W = ((np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 5) % 61 - 30).astype(np.float32) / 4.0
    y = run_gemm_ex(L, ref.GGML_TYPES["f16"], ref.quantize(ref.GGML_TYPES["f16"], W), N, K, X, epi=1, tile=tile)
    want = X.astype(np.float64) @ W.astype(np.float64).T
    assert np.array_equal(y, want.astype(np.float16).astype(np.float32)), _diff_report(want.astype(np.float32), y)


@pytest.mark.parametrize("tile", TILES32)
def test_gemm32_race_screen_many_tiles_is_deterministic(L, tile):
    """several rounds of tiles on 256 CUs, short and long K: run to run bit-identical (the ring hand-offs are counted, not lucky)."""
    rng = np.random.default_rng(5)
    for (M, N, K) in [(12800, 768, 256), (5000, 1536, 768)]:
        tid = ref.GGML_TYPES["q4_0"]
        raw = ref.quantize(tid, _weights(rng, N, K))
        Wd = ref.dequantize(tid, raw, N, K)
        X = rng.standard_normal((M, K)).astype(np.float32)
        bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
        base = run_gemm_ex(L, tid, raw, N, K, X, bias=bias, epi=3, tile=tile)
        want, bound = _want(_h(X), Wd, bias, 3, 0, 1.0)
        assert np.all(np.abs(base - want) <= bound)
        for _ in range(4):
            y = run_gemm_ex(L, tid, raw, N, K, X, bias=bias, epi=3, tile=tile)
            assert np.array_equal(base, y), _diff_report(base, y)


@pytest.mark.parametrize("tname", ["q4_0", "f16"])
@pytest.mark.parametrize("tile1,tile2,epi2", [(160128, 256261, 1), (128128, 320261, 3), (65064, 256261, 2), (160256, 320261, 1)])
@pytest.mark.parametrize("fold", [1, 2])
def test_gemm32_lnfold_consumer(L, tname, tile1, tile2, epi2, fold):
    """Consumer half of the LayerNorm fold in the 32 x 32 x 16 kernel (statistics of 64- and 32-column slots, plain and centred operand):
    inside the rigorous bound of the float64 LayerNorm + product and within 3 fp16 ulp of the three-launch form."""
    rng = np.random.default_rng(17 + epi2)
    M, h, K1, N2 = 403, 320, 192, 448
    tid, raw1, raw2, Wd2, A, resid, b1, g, beta, b2 = _lnfold_case(rng, tname, M, h, K1, N2)
    qc, qs = (128, 0.125) if epi2 == 1 else (0, 1.0)
    x1a, ya = run_lnfold(L, tid, raw1, h, K1, raw2, N2, A, b1, resid, g, beta, b2, epi2, tile1, 128128, 0, qc, qs)
    x1b, yb = run_lnfold(L, tid, raw1, h, K1, raw2, N2, A, b1, resid, g, beta, b2, epi2, tile1, tile2, fold, qc, qs)
    assert np.array_equal(x1a, x1b)
    want, bound, lin = _lnfold_reference(x1b, g, beta, Wd2, b2, epi2, qc, qs)
    err = np.abs(yb - want)
    assert np.all(np.isfinite(yb))
    bad = np.argwhere(err > bound)
    assert bad.size == 0, "%d/%d bad, first %s got %g want %g bound %g" % (len(bad), yb.size, bad[0], yb[tuple(bad[0])], want[tuple(bad[0])], bound[tuple(bad[0])])
    scale = np.maximum(np.abs(ya), np.sqrt((ya.astype(np.float64) ** 2).mean(1, keepdims=True)))
    ab = np.abs(yb - ya) / (scale * 2.0 ** -10)
    assert ab.max() <= 3.0 and np.quantile(ab, 0.999) <= 2.0, "A/B max %.2f ulp, p99.9 %.2f" % (ab.max(), np.quantile(ab, 0.999))


def test_gemm32_at_the_baseline_shapes(L):
    """q/k/v and FFN-up of the BASELINE batch (256 ViT-B/32 images = 12800 token rows; 10290 token rows of 256 texts) and q/k/v of 130 ViT-L/14
    images: sampled rows of the whole output against float64, and the heuristic sends these shapes here."""
    rng = np.random.default_rng(2026)
    assert L.clip_amd_test_gemm_tile(65792, 3072, 1024, 0) == 320261 and L.clip_amd_test_gemm_tile(65792, 4096, 1024, 0) == 320261     # ViT-L/14, batch 256, alone
    for (M, N, K, epi, tile) in [(12800, 2304, 768, 1, 256261), (12800, 3072, 768, 3, 320261), (10290, 1536, 512, 1, 256261), (10290, 2048, 512, 3, 320261),
                                 (33410, 3072, 1024, 1, 320261)]:
        tid = ref.GGML_TYPES["q4_0"]
        raw = ref.quantize(tid, _weights(rng, N, K))
        Wd = ref.dequantize(tid, raw, N, K)
        X = rng.standard_normal((M, K)).astype(np.float32)
        bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
        qc, qs = (N // 3, 0.125) if epi == 1 else (0, 1.0)
        y = run_gemm_ex(L, tid, raw, N, K, X, bias=bias, epi=epi, tile=tile, qcols=qc, qscale=qs)
        rows = np.unique(np.concatenate([np.arange(0, 330), np.arange(M - 330, M), rng.integers(0, M, 600)]))
        want, bound = _want(_h(X[rows]), Wd, bias, epi, qc, qs)
        assert np.all(np.isfinite(y))
        assert np.all(np.abs(y[rows] - want) <= bound), (M, N, K, float(np.abs(y[rows] - want).max()))
