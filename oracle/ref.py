"""ctypes front-end of the CPU oracle (oracle/clip_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (clip_cpp_amd / libclip.so) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libclip_oracle.so")

MODE_FAITHFUL = 0  # ggml CPU numerics (q8 activation quantisation, fp16 tables, ...)
MODE_IDEAL = 1     # dequantised weights, f32 activations, libm

GGML_TYPES = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}


def build(force=False):
    src = os.path.join(_HERE, "clip_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libclip_oracle.so"])
    return _LIB_PATH


_lib = None


def host_cores():
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota (a container on a
    256-thread host often owns only a few cores; spawning 256 OpenMP threads there is catastrophically slow)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, min(n, 32))


def _threads(n):
    return n if n and n > 0 else host_cores()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_float)
        L.orc_load.restype = vp
        L.orc_load.argtypes = [C.c_char_p]
        L.orc_free.argtypes = [vp]
        L.orc_info.argtypes = [vp, C.POINTER(C.c_int32)]
        L.orc_image_batch_encode.argtypes = [vp, i32, i32, f32p, i32, f32p, i32]
        L.orc_image_batch_encode_taps.argtypes = [vp, i32, i32, f32p, i32, f32p, i32, f32p, f32p]
        L.orc_text_encode.argtypes = [vp, i32, i32, C.POINTER(C.c_int32), i32, f32p, i32]
        L.orc_tokenize.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32), i32]
        L.orc_preprocess.argtypes = [vp, C.POINTER(C.c_uint8), i32, i32, f32p]
        L.orc_similarity.restype = C.c_float
        L.orc_similarity.argtypes = [f32p, f32p, i32]
        L.orc_softmax_with_sorting.argtypes = [f32p, i32, f32p, C.POINTER(C.c_int)]
        L.orc_row_bytes.restype = C.c_size_t
        L.orc_row_bytes.argtypes = [i32, i64]
        L.orc_quantize.restype = C.c_size_t
        L.orc_quantize.argtypes = [i32, f32p, vp, i64, i64]
        L.orc_dequantize.argtypes = [i32, vp, f32p, i64, i64]
        L.orc_f2h.restype = C.c_uint16
        L.orc_f2h.argtypes = [C.c_float]
        L.orc_h2f.restype = C.c_float
        L.orc_h2f.argtypes = [C.c_uint16]
        L.orc_mul_mat.argtypes = [i32, vp, i64, i64, f32p, i64, f32p, i32, i32]
        L.orc_set_dot_simd.restype = i32
        L.orc_set_dot_simd.argtypes = [i32]
        L.orc_layer_norm.argtypes = [f32p, f32p, i64, i64, f32p, f32p, C.c_float]
        L.orc_activation.argtypes = [f32p, i64, i32, i32]
        L.orc_softmax_rows.argtypes = [f32p, i64, i64, i32]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def row_bytes(type_id, k):
    return lib().orc_row_bytes(type_id, k)


def quantize(type_id, w):
    """w: float32 [nrows, k] -> uint8 bytes in the ggml block format."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    nrows, k = w.shape
    out = np.empty(row_bytes(type_id, k) * nrows, dtype=np.uint8)
    n = lib().orc_quantize(type_id, _fp(w), out.ctypes.data_as(C.c_void_p), nrows, k)
    assert n == out.size
    return out


def dequantize(type_id, raw, nrows, k):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    out = np.empty((nrows, k), dtype=np.float32)
    lib().orc_dequantize(type_id, raw.ctypes.data_as(C.c_void_p), _fp(out), nrows, k)
    return out


DOT_SCALAR, DOT_AVX2, DOT_VNNI, DOT_VNNI16, DOT_BEST = 0, 1, 2, 3, -1


def set_dot_simd(form=DOT_BEST):
    """Form of the integer dot products of the block-quantised mul_mat (clip_oracle.cpp mul_mat_quant_simd): 0 scalar loop, 1 AVX2 vpmaddubsw,
    2 AVX-512 VNNI vpdpbusd (one dot product at a time), 3 AVX-512 VNNI with 16 output columns per register (no horizontal sums), -1 the best the CPU has (default).  Every form gives the same bits (tests/test_oracle_golden.py).  Returns the form in use."""
    return lib().orc_set_dot_simd(form)


def dot_simd_name(form=None):
    form = set_dot_simd(DOT_BEST) if form is None else form
    return {0: "scalar", 1: "AVX2 vpmaddubsw", 2: "AVX-512 VNNI vpdpbusd", 3: "AVX-512 VNNI vpdpbusd, 16 output columns per register"}[form]


def mul_mat(type_id, raw, N, K, X, mode=MODE_FAITHFUL, n_threads=0):
    """Y[M,N] = X[M,K] . W[N,K]^T with ggml numerics (mode 0) or f32 (mode 1)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    X = np.ascontiguousarray(X, dtype=np.float32)
    M = X.shape[0]
    Y = np.empty((M, N), dtype=np.float32)
    lib().orc_mul_mat(type_id, raw.ctypes.data_as(C.c_void_p), N, K, _fp(X), M, _fp(Y), mode, _threads(n_threads))
    return Y


def layer_norm(x, w, b, eps):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    lib().orc_layer_norm(_fp(x), _fp(y), x.shape[0], x.shape[1], _fp(w), _fp(b), eps)
    return y


def activation(x, use_gelu, mode):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().orc_activation(_fp(y), y.size, int(use_gelu), mode)
    return y


def softmax_rows(s, mode):
    y = np.ascontiguousarray(s, dtype=np.float32).copy()
    lib().orc_softmax_rows(_fp(y), y.shape[0], y.shape[1], mode)
    return y


class OracleModel:
    INFO_KEYS = ["has_text", "has_vision", "use_gelu", "ftype", "t_n_vocab", "t_npos", "t_h", "t_ff", "t_proj",
                 "t_nh", "t_nl", "v_S", "v_P", "v_h", "v_ff", "v_proj", "v_nh", "v_nl", "n_tensors"]

    def __init__(self, path):
        self.h = lib().orc_load(os.fsencode(path))
        if not self.h:
            raise RuntimeError("oracle: cannot load %s" % path)
        buf = (C.c_int32 * 32)()
        n = lib().orc_info(self.h, buf)
        self.info = dict(zip(self.INFO_KEYS, list(buf)[:n]))

    def close(self):
        if self.h:
            lib().orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def image_batch_encode(self, imgs, normalize=True, mode=MODE_FAITHFUL, n_threads=0, taps=False):
        """imgs: float32 [B,S,S,3] (already preprocessed, interleaved RGB)."""
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        B = imgs.shape[0]
        out = np.empty((B, self.info["v_proj"]), dtype=np.float32)
        if taps:
            T = (self.info["v_S"] // self.info["v_P"]) ** 2 + 1
            t0 = np.empty((B * T, self.info["v_h"]), dtype=np.float32)
            t1 = np.empty_like(t0)
            ok = lib().orc_image_batch_encode_taps(self.h, mode, _threads(n_threads), _fp(imgs), B, _fp(out), int(normalize),
                                                   _fp(t0), _fp(t1))
            assert ok
            return out, t0, t1
        ok = lib().orc_image_batch_encode(self.h, mode, _threads(n_threads), _fp(imgs), B, _fp(out), int(normalize))
        assert ok
        return out

    def text_encode(self, ids, normalize=True, mode=MODE_FAITHFUL, n_threads=0):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.empty(self.info["t_proj"], dtype=np.float32)
        ok = lib().orc_text_encode(self.h, mode, _threads(n_threads), ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size,
                                   _fp(out), int(normalize))
        if not ok:
            raise RuntimeError("oracle text_encode failed")
        return out

    def tokenize(self, text):
        if isinstance(text, str):
            text = text.encode("utf-8")
        buf = (C.c_int32 * 4096)()
        n = lib().orc_tokenize(self.h, text, buf, 4096)
        if n < 0:
            raise RuntimeError("no text encoder")
        return np.array(list(buf)[:n], dtype=np.int32)

    def preprocess(self, rgb):
        """rgb: uint8 [ny,nx,3] -> float32 [S,S,3]."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        ny, nx, _ = rgb.shape
        S = self.info["v_S"]
        out = np.empty((S, S, 3), dtype=np.float32)
        ok = lib().orc_preprocess(self.h, rgb.ctypes.data_as(C.POINTER(C.c_uint8)), nx, ny, _fp(out))
        assert ok
        return out


def similarity(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_similarity(_fp(a), _fp(b), a.size))


def softmax_with_sorting(arr):
    arr = np.ascontiguousarray(arr, dtype=np.float32).copy()
    n = arr.size
    scores = np.empty(n, dtype=np.float32)
    idx = np.empty(n, dtype=np.int32)
    lib().orc_softmax_with_sorting(_fp(arr), n, _fp(scores), idx.ctypes.data_as(C.POINTER(C.c_int)))
    return scores, idx
