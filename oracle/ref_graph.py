"""ctypes binding of oracle/_ref/libclip_ref.so — TEST INFRASTRUCTURE (checker only).

libclip_ref.so is the reference's own clip.cpp, compiled unchanged from /root/reference by `make -C oracle ref`, on top of
oracle/ggml_shim (an eager stand-in for the slice of the absent ggml submodule that clip.cpp calls; the arithmetic behind every op is the
oracle's restatement).  It runs the reference's loader, tokenizer, preprocessing, scoring and its two graph builders op by op, and so
checks the oracle's WIRING against the reference's source; it is not ggml and does not pin the op arithmetic
(oracle/ggml_shim/ggml/ggml.h).  The reference's loader exits the process on tensor counts other than those of the base / large / huge
models: use the `base12` / `large24` / `b32` ... configurations of oracle/fixtures.py.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libclip_ref.so")


class Tokens(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_int32)), ("size", C.c_size_t)]


class ImageU8(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("data", C.POINTER(C.c_uint8)), ("size", C.c_size_t)]


class ImageF32(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("data", C.POINTER(C.c_float)), ("size", C.c_size_t)]


class ImageF32Batch(C.Structure):
    _fields_ = [("data", C.POINTER(ImageF32)), ("size", C.c_size_t)]


class Hparams(C.Structure):       # clip_text_hparams and clip_vision_hparams have the same shape: 7 x int32 + float
    _fields_ = [("f%d" % i, C.c_int32) for i in range(7)] + [("eps", C.c_float)]


_lib = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.clip_model_load.restype = C.c_void_p
        L.clip_model_load.argtypes = [C.c_char_p, C.c_int]
        L.clip_free.argtypes = [C.c_void_p]
        for f in (L.clip_get_text_hparams, L.clip_get_vision_hparams):
            f.restype = C.POINTER(Hparams)
            f.argtypes = [C.c_void_p]
        L.clip_tokenize.restype = C.c_bool
        L.clip_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Tokens)]
        L.clip_image_preprocess.restype = C.c_bool
        L.clip_image_preprocess.argtypes = [C.c_void_p, C.POINTER(ImageU8), C.POINTER(ImageF32)]
        L.clip_image_load_from_file.restype = C.c_bool
        L.clip_image_load_from_file.argtypes = [C.c_char_p, C.POINTER(ImageU8)]
        L.clip_text_encode.restype = C.c_bool
        L.clip_text_encode.argtypes = [C.c_void_p, C.c_int, C.POINTER(Tokens), C.POINTER(C.c_float), C.c_bool]
        L.clip_image_batch_encode.restype = C.c_bool
        L.clip_image_batch_encode.argtypes = [C.c_void_p, C.c_int, C.POINTER(ImageF32Batch), C.POINTER(C.c_float), C.c_bool]
        L.clip_similarity_score.restype = C.c_float
        L.clip_similarity_score.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.softmax_with_sorting.restype = C.c_bool
        L.softmax_with_sorting.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.clip_compare_text_and_image.restype = C.c_bool
        L.clip_compare_text_and_image.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(ImageU8), C.POINTER(C.c_float)]
        L.clip_zero_shot_label_image.restype = C.c_bool
        L.clip_zero_shot_label_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(ImageU8), C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_float),
                                                 C.POINTER(C.c_int)]
        # OpenMP team size of the op kernels = the cores this process may use (a container quota can be far below the machine's core count,
        # and a 256-thread team spinning on 16 CPUs is pathologically slow): the oracle's own rule (ref.host_cores), set through a 1 x 1 product
        from . import ref as _ref
        L.orc_mul_mat.restype = C.c_int
        L.orc_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float), C.c_int, C.c_int]
        one, out = np.ones(1, np.float32), np.zeros(1, np.float32)
        L.orc_mul_mat(0, one.ctypes.data_as(C.c_void_p), 1, 1, _fp(one), 1, _fp(out), 1, int(_ref.host_cores()))
        _lib = L
    return _lib


class ReferenceModel:
    """The reference's clip_ctx, loaded by the reference's clip_model_load."""

    def __init__(self, path, verbosity=0):
        self.ctx = lib().clip_model_load(os.fsencode(path), verbosity)
        if not self.ctx:
            raise RuntimeError("reference clip_model_load failed for %r" % path)

    def close(self):
        if self.ctx:
            lib().clip_free(self.ctx)
            self.ctx = None

    def text_hparams(self):
        h = lib().clip_get_text_hparams(self.ctx).contents
        return dict(n_vocab=h.f0, num_positions=h.f1, hidden_size=h.f2, n_intermediate=h.f3, projection_dim=h.f4, n_head=h.f5, n_layer=h.f6, eps=h.eps)

    def vision_hparams(self):
        h = lib().clip_get_vision_hparams(self.ctx).contents
        return dict(image_size=h.f0, patch_size=h.f1, hidden_size=h.f2, n_intermediate=h.f3, projection_dim=h.f4, n_head=h.f5, n_layer=h.f6, eps=h.eps)

    def tokenize(self, text):
        t = Tokens()
        if not lib().clip_tokenize(self.ctx, text.encode("utf-8"), C.byref(t)):
            return None
        return [t.data[i] for i in range(t.size)]      # (the reference has no call that frees tokens->data; a few bytes per call leak here as there)

    def preprocess(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        u8 = ImageU8(rgb.shape[1], rgb.shape[0], rgb.ctypes.data_as(C.POINTER(C.c_uint8)), rgb.size)
        f32 = ImageF32()
        if not lib().clip_image_preprocess(self.ctx, C.byref(u8), C.byref(f32)):
            return None
        return np.ctypeslib.as_array(f32.data, shape=(f32.ny, f32.nx, 3)).copy()          # (new[]-allocated by the callee; leaked like above)

    def image_batch_encode(self, imgs, normalize=True, n_threads=4):
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        B, S = imgs.shape[0], imgs.shape[1]
        arr = (ImageF32 * B)()
        for b in range(B):
            arr[b] = ImageF32(S, S, _fp(imgs[b]), imgs[b].size)
        batch = ImageF32Batch(arr, B)
        out = np.empty((B, self.vision_hparams()["projection_dim"]), dtype=np.float32)
        if not lib().clip_image_batch_encode(self.ctx, n_threads, C.byref(batch), _fp(out), normalize):
            return None
        return out

    def text_encode(self, ids, normalize=True, n_threads=4):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        t = Tokens(ids.ctypes.data_as(C.POINTER(C.c_int32)), ids.size)
        out = np.empty(self.text_hparams()["projection_dim"], dtype=np.float32)
        if not lib().clip_text_encode(self.ctx, n_threads, C.byref(t), _fp(out), normalize):
            return None
        return out

    def compare_text_and_image(self, text, rgb, n_threads=4):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        u8 = ImageU8(rgb.shape[1], rgb.shape[0], rgb.ctypes.data_as(C.POINTER(C.c_uint8)), rgb.size)
        score = C.c_float()
        if not lib().clip_compare_text_and_image(self.ctx, n_threads, text.encode("utf-8"), C.byref(u8), C.byref(score)):
            return None
        return score.value

    def zero_shot(self, rgb, labels, n_threads=4):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        u8 = ImageU8(rgb.shape[1], rgb.shape[0], rgb.ctypes.data_as(C.POINTER(C.c_uint8)), rgb.size)
        n = len(labels)
        arr = (C.c_char_p * n)(*[l.encode("utf-8") for l in labels])
        scores = np.empty(n, dtype=np.float32)
        idx = np.empty(n, dtype=np.int32)
        if not lib().clip_zero_shot_label_image(self.ctx, n_threads, C.byref(u8), arr, n, _fp(scores), idx.ctypes.data_as(C.POINTER(C.c_int))):
            return None
        return scores, idx
