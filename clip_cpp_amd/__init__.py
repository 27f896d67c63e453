"""clip_cpp_amd — Python host layer over libclip.so (the MI355X-native CLIP encoder).

Mirrors the reference's ctypes binding (`examples/python_bindings/clip_cpp/clip.py:103-424`):
class `Clip` with `tokenize`, `encode_text`, `load_preprocess_encode_image`, `calculate_similarity`,
`compare_text_and_image`, `zero_shot_label_image`, plus MI355X extensions: batched encoders on host
(numpy) or device (raw HBM pointers, e.g. torch tensors) and the data-parallel helpers in
`clip_cpp_amd.parallel`.  Everything heavy happens in the C-ABI library (`include/clip.h`,
`include/clip_amd.h`); this module holds no arithmetic and NO CPU fallback: if the HIP library or a
GPU is missing the encoders raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLIP_AMD_LIB") or os.path.join(_HERE, "libclip.so")   # CLIP_AMD_LIB: kernel A/B builds (scripts/build_variant.sh)


class ClipTextHparams(C.Structure):  # reference clip.h:14-23
    _fields_ = [("n_vocab", C.c_int32), ("num_positions", C.c_int32), ("hidden_size", C.c_int32),
                ("n_intermediate", C.c_int32), ("projection_dim", C.c_int32), ("n_head", C.c_int32),
                ("n_layer", C.c_int32), ("eps", C.c_float)]


class ClipVisionHparams(C.Structure):  # reference clip.h:25-34
    _fields_ = [("image_size", C.c_int32), ("patch_size", C.c_int32), ("hidden_size", C.c_int32),
                ("n_intermediate", C.c_int32), ("projection_dim", C.c_int32), ("n_head", C.c_int32),
                ("n_layer", C.c_int32), ("eps", C.c_float)]


class ClipTokens(C.Structure):  # reference clip.h:37-40
    _fields_ = [("data", C.POINTER(C.c_int32)), ("size", C.c_size_t)]


class ClipImageU8(C.Structure):  # reference clip.h:50-55
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("data", C.POINTER(C.c_uint8)), ("size", C.c_size_t)]


class ClipImageF32(C.Structure):  # reference clip.h:57-64
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("data", C.POINTER(C.c_float)), ("size", C.c_size_t)]


class ClipImageU8Batch(C.Structure):
    _fields_ = [("data", C.POINTER(ClipImageU8)), ("size", C.c_size_t)]


class ClipImageF32Batch(C.Structure):
    _fields_ = [("data", C.POINTER(ClipImageF32)), ("size", C.c_size_t)]


# every symbol include/clip.h and include/clip_amd.h declare (tests check they are all exported)
API_SYMBOLS = [
    "clip_model_load", "clip_free", "clip_get_text_hparams", "clip_get_vision_hparams", "clip_tokenize",
    "clip_image_u8_make", "clip_image_f32_make", "clip_image_u8_clean", "clip_image_f32_clean", "clip_image_u8_free",
    "clip_image_f32_free", "clip_image_load_from_file", "clip_image_preprocess", "clip_image_batch_preprocess",
    "clip_text_encode", "clip_image_encode", "clip_image_batch_encode", "clip_compare_text_and_image",
    "clip_similarity_score", "softmax_with_sorting", "clip_zero_shot_label_image", "clip_model_quantize",
    "ggml_time_init", "ggml_time_us", "ggml_time_ms",
]
AMD_SYMBOLS = [
    "clip_amd_device_count", "clip_amd_model_load", "clip_amd_model_load_multi", "clip_amd_ctx_device_count", "clip_amd_weights_from_cache", "clip_amd_shard_bounds",
    "clip_amd_gathered_embeddings", "clip_amd_ctx_device", "clip_amd_set_stream", "clip_amd_set_device_shared", "clip_amd_image_batch_encode_device_multi",
    "clip_amd_text_batch_encode_device_multi", "clip_amd_encode_pair_device_multi",
    "clip_amd_image_batch_encode_device", "clip_text_batch_encode", "clip_amd_text_batch_encode_device",
    "clip_amd_image_batch_preprocess_device", "clip_amd_image_batch_encode_u8",
    "clip_amd_zero_shot_score_device", "clip_amd_zero_shot_label_images",
    "clip_amd_synchronize", "clip_amd_profile_enable", "clip_amd_profile_read", "clip_amd_profile_report",
    "clip_amd_test_gemm", "clip_amd_test_gemm_ex", "clip_amd_test_gemm_tile", "clip_amd_test_gemm_tile_ex", "clip_amd_test_skinny", "clip_amd_test_layernorm", "clip_amd_test_attention", "clip_amd_bench_gemm",
]

_lib = None


def build_lib(force=False):
    import importlib
    return importlib.import_module(__name__ + ".build").build(force=force)


def lib():
    """Load libclip.so (building it first if the in-tree binary is missing). Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build_lib()
    L = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    L.clip_model_load.restype = vp
    L.clip_model_load.argtypes = [C.c_char_p, i32]
    L.clip_amd_model_load.restype = vp
    L.clip_amd_model_load.argtypes = [C.c_char_p, i32, i32]
    L.clip_amd_model_load_multi.restype = vp
    L.clip_amd_model_load_multi.argtypes = [C.c_char_p, i32, i32]
    L.clip_amd_ctx_device_count.restype = i32
    L.clip_amd_ctx_device_count.argtypes = [vp]
    L.clip_amd_weights_from_cache.restype = i32
    L.clip_amd_weights_from_cache.argtypes = [vp]
    L.clip_amd_shard_bounds.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.clip_amd_gathered_embeddings.restype = vp
    L.clip_amd_gathered_embeddings.argtypes = [vp, i32]
    L.clip_free.argtypes = [vp]
    L.clip_get_text_hparams.restype = C.POINTER(ClipTextHparams)
    L.clip_get_text_hparams.argtypes = [vp]
    L.clip_get_vision_hparams.restype = C.POINTER(ClipVisionHparams)
    L.clip_get_vision_hparams.argtypes = [vp]
    L.clip_tokenize.restype = C.c_bool
    L.clip_tokenize.argtypes = [vp, C.c_char_p, C.POINTER(ClipTokens)]
    L.clip_image_u8_make.restype = C.POINTER(ClipImageU8)
    L.clip_image_f32_make.restype = C.POINTER(ClipImageF32)
    L.clip_image_u8_clean.argtypes = [C.POINTER(ClipImageU8)]
    L.clip_image_f32_clean.argtypes = [C.POINTER(ClipImageF32)]
    L.clip_image_u8_free.argtypes = [C.POINTER(ClipImageU8)]
    L.clip_image_f32_free.argtypes = [C.POINTER(ClipImageF32)]
    L.clip_image_load_from_file.restype = C.c_bool
    L.clip_image_load_from_file.argtypes = [C.c_char_p, C.POINTER(ClipImageU8)]
    L.clip_image_preprocess.restype = C.c_bool
    L.clip_image_preprocess.argtypes = [vp, C.POINTER(ClipImageU8), C.POINTER(ClipImageF32)]
    L.clip_image_batch_preprocess.argtypes = [vp, i32, C.POINTER(ClipImageU8Batch), C.POINTER(ClipImageF32Batch)]
    L.clip_text_encode.restype = C.c_bool
    L.clip_text_encode.argtypes = [vp, i32, C.POINTER(ClipTokens), f32p, C.c_bool]
    L.clip_text_batch_encode.restype = C.c_bool
    L.clip_text_batch_encode.argtypes = [vp, i32, C.POINTER(ClipTokens), C.c_size_t, f32p, C.c_bool]
    L.clip_image_encode.restype = C.c_bool
    L.clip_image_encode.argtypes = [vp, i32, C.POINTER(ClipImageF32), f32p, C.c_bool]
    L.clip_image_batch_encode.restype = C.c_bool
    L.clip_image_batch_encode.argtypes = [vp, i32, C.POINTER(ClipImageF32Batch), f32p, C.c_bool]
    L.clip_compare_text_and_image.restype = C.c_bool
    L.clip_compare_text_and_image.argtypes = [vp, i32, C.c_char_p, C.POINTER(ClipImageU8), f32p]
    L.clip_similarity_score.restype = C.c_float
    L.clip_similarity_score.argtypes = [f32p, f32p, i32]
    L.softmax_with_sorting.restype = C.c_bool
    L.softmax_with_sorting.argtypes = [f32p, i32, f32p, C.POINTER(C.c_int)]
    L.clip_zero_shot_label_image.restype = C.c_bool
    L.clip_zero_shot_label_image.argtypes = [vp, i32, C.POINTER(ClipImageU8), C.POINTER(C.c_char_p), C.c_size_t, f32p,
                                             C.POINTER(C.c_int)]
    L.clip_model_quantize.restype = C.c_bool
    L.clip_model_quantize.argtypes = [C.c_char_p, C.c_char_p, i32]
    L.clip_amd_device_count.restype = i32
    L.clip_amd_ctx_device.restype = i32
    L.clip_amd_ctx_device.argtypes = [vp]
    L.clip_amd_set_stream.argtypes = [vp, vp]
    L.clip_amd_set_device_shared.restype = None
    L.clip_amd_set_device_shared.argtypes = [vp, i32]
    L.clip_amd_synchronize.argtypes = [vp]
    L.clip_amd_image_batch_encode_device.restype = C.c_bool
    L.clip_amd_image_batch_encode_device.argtypes = [vp, vp, i32, vp, C.c_bool]
    L.clip_amd_image_batch_preprocess_device.restype = C.c_bool
    L.clip_amd_image_batch_preprocess_device.argtypes = [vp, C.POINTER(ClipImageU8), i32, vp]
    L.clip_amd_zero_shot_score_device.restype = C.c_bool
    L.clip_amd_zero_shot_score_device.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
    L.clip_amd_zero_shot_label_images.restype = C.c_bool
    L.clip_amd_zero_shot_label_images.argtypes = [vp, C.POINTER(ClipImageU8), i32, C.POINTER(C.c_char_p), C.c_size_t, f32p, C.POINTER(C.c_int)]
    L.clip_amd_image_batch_encode_u8.restype = C.c_bool
    L.clip_amd_image_batch_encode_u8.argtypes = [vp, C.POINTER(ClipImageU8), i32, f32p, C.c_bool]
    L.clip_amd_text_batch_encode_device.restype = C.c_bool
    L.clip_amd_text_batch_encode_device.argtypes = [vp, vp, C.POINTER(C.c_int32), i32, vp, C.c_bool]
    L.clip_amd_image_batch_encode_device_multi.restype = C.c_bool
    L.clip_amd_image_batch_encode_device_multi.argtypes = [vp, C.POINTER(vp), i32, C.c_bool, f32p]
    L.clip_amd_text_batch_encode_device_multi.restype = C.c_bool
    L.clip_amd_text_batch_encode_device_multi.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), i32, C.c_bool, f32p]
    L.clip_amd_encode_pair_device_multi.restype = C.c_bool
    L.clip_amd_encode_pair_device_multi.argtypes = [vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(C.c_int32), i32, C.c_bool, f32p, f32p]
    L.clip_amd_profile_enable.argtypes = [vp, C.c_bool]
    L.clip_amd_profile_report.restype = i32
    L.clip_amd_profile_report.argtypes = [vp, C.c_char_p, i32, C.c_bool]
    L.clip_amd_test_gemm.restype = i32
    L.clip_amd_test_gemm.argtypes = [i32, vp, C.c_int64, C.c_int64, f32p, C.c_int64, f32p, f32p, f32p, i32, i32]
    L.clip_amd_test_gemm_ex.restype = i32
    L.clip_amd_test_gemm_ex.argtypes = [i32, vp, C.c_int64, C.c_int64, f32p, C.c_int64, f32p, f32p, f32p, i32, i32, i32, C.c_float, i32, i32, f32p]
    L.clip_amd_test_lnfold.restype = i32
    L.clip_amd_test_lnfold.argtypes = [i32, vp, C.c_int64, C.c_int64, vp, C.c_int64, f32p, C.c_int64, f32p, f32p, f32p, f32p, C.c_float, f32p,
                                       i32, i32, i32, i32, i32, C.c_float, f32p, f32p]
    L.clip_amd_test_gemm_tile.restype = i32
    L.clip_amd_test_gemm_tile.argtypes = [C.c_int64, C.c_int64, C.c_int64, i32]
    L.clip_amd_test_gemm_tile_ex.restype = i32
    L.clip_amd_test_gemm_tile_ex.argtypes = [C.c_int64, C.c_int64, C.c_int64, i32, i32]
    L.clip_amd_test_skinny.restype = i32
    L.clip_amd_test_skinny.argtypes = [i32, vp, C.c_int64, C.c_int64, f32p, C.c_int64, f32p, f32p, f32p, f32p, C.c_float, f32p, i32, i32, C.c_float, f32p]
    L.clip_amd_bench_gemm.restype = C.c_float
    L.clip_amd_bench_gemm.argtypes = [i32, C.c_int64, C.c_int64, C.c_int64, i32, i32, i32]
    L.clip_amd_test_layernorm.restype = i32
    L.clip_amd_test_layernorm.argtypes = [f32p, f32p, f32p, C.c_float, C.c_int64, C.c_int64, f32p, i32]
    L.clip_amd_test_attention.restype = i32
    L.clip_amd_test_attention.argtypes = [f32p, i32, i32, i32, i32, i32, f32p]
    _lib = L
    return L


def device_count():
    return lib().clip_amd_device_count()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _struct_to_dict(s):
    return {f: getattr(s, f) for f, _ in s._fields_}


class Clip:
    """Same surface as the reference binding's `Clip` (clip.py:215-424), minus the HF-hub downloader."""

    def __init__(self, model_path_or_repo_id, verbosity=0, device=None, n_devices=None):
        L = lib()
        path = os.fsencode(model_path_or_repo_id)
        if n_devices is not None:       # single process, replicas on n_devices GPUs, batch sharding + RCCL all-gather behind the C ABI
            self.ctx = L.clip_amd_model_load_multi(path, verbosity, int(n_devices))
        elif device is None:
            self.ctx = L.clip_model_load(path, verbosity)
        else:
            self.ctx = L.clip_amd_model_load(path, verbosity, int(device))
        if not self.ctx:
            raise RuntimeError("clip_model_load failed for %r (no file, malformed GGUF, or no HIP device)" % (model_path_or_repo_id,))
        self.vec_dim = self.vision_config["projection_dim"] or self.text_config["projection_dim"]

    # ---- reference surface ----
    @property
    def vision_config(self):
        return _struct_to_dict(lib().clip_get_vision_hparams(self.ctx).contents)

    @property
    def text_config(self):
        return _struct_to_dict(lib().clip_get_text_hparams(self.ctx).contents)

    @property
    def device(self):
        return lib().clip_amd_ctx_device(self.ctx)

    def tokenize(self, text):
        t = ClipTokens()
        if not lib().clip_tokenize(self.ctx, text.encode("utf-8"), C.byref(t)):
            raise RuntimeError("Could not tokenize text")
        return [t.data[i] for i in range(t.size)]   # (t.data is leaked exactly as in the reference; a few bytes)

    def encode_text(self, tokens, n_threads=os.cpu_count(), normalize=True):
        arr = (C.c_int32 * len(tokens))(*tokens)
        t = ClipTokens(C.cast(arr, C.POINTER(C.c_int32)), len(tokens))
        out = np.empty(self.text_config["projection_dim"], dtype=np.float32)
        if not lib().clip_text_encode(self.ctx, n_threads or 1, C.byref(t), _fp(out), normalize):
            raise RuntimeError("Could not encode text")
        return out.tolist()

    def load_preprocess_encode_image(self, image_path, n_threads=os.cpu_count(), normalize=True):
        L = lib()
        img = L.clip_image_u8_make()
        res = L.clip_image_f32_make()
        try:
            if not L.clip_image_load_from_file(os.fsencode(image_path), img):
                raise RuntimeError("Could not load image '%s'" % image_path)
            if not L.clip_image_preprocess(self.ctx, img, res):
                raise RuntimeError("Could not preprocess image")
            out = np.empty(self.vec_dim, dtype=np.float32)
            if not L.clip_image_encode(self.ctx, n_threads or 1, res, _fp(out), normalize):
                raise RuntimeError("Could not encode image")
            return out.tolist()
        finally:
            L.clip_image_u8_free(img)
            L.clip_image_f32_free(res)

    def calculate_similarity(self, text_embedding, image_embedding):
        a = np.asarray(text_embedding, dtype=np.float32)
        b = np.asarray(image_embedding, dtype=np.float32)
        return float(lib().clip_similarity_score(_fp(a), _fp(b), a.size))

    def compare_text_and_image(self, text, image_path, n_threads=os.cpu_count()):
        L = lib()
        img = L.clip_image_u8_make()
        try:
            if not L.clip_image_load_from_file(os.fsencode(image_path), img):
                raise RuntimeError("Could not load image '%s'" % image_path)
            score = C.c_float()
            if not L.clip_compare_text_and_image(self.ctx, n_threads or 1, text.encode("utf-8"), img, C.byref(score)):
                raise RuntimeError("Could not compare text and image")
            return score.value
        finally:
            L.clip_image_u8_free(img)

    def zero_shot_label_image(self, image_path, labels, n_threads=os.cpu_count()):
        L = lib()
        img = L.clip_image_u8_make()
        try:
            if not L.clip_image_load_from_file(os.fsencode(image_path), img):
                raise RuntimeError("Could not load image '%s'" % image_path)
            return self.zero_shot_label_pixels(img, labels, n_threads)
        finally:
            L.clip_image_u8_free(img)

    # ---- extensions ----
    def zero_shot_label_pixels(self, img_u8_ptr, labels, n_threads=1):
        n = len(labels)
        arr = (C.c_char_p * n)(*[s.encode("utf-8") for s in labels])
        scores = np.empty(n, dtype=np.float32)
        idx = np.empty(n, dtype=np.int32)
        if not lib().clip_zero_shot_label_image(self.ctx, n_threads or 1, img_u8_ptr, arr, n, _fp(scores),
                                                idx.ctypes.data_as(C.POINTER(C.c_int))):
            raise RuntimeError("Could not zero-shot label image")
        return scores.tolist(), idx.tolist()

    def preprocess(self, rgb):
        """uint8 [ny,nx,3] -> float32 [S,S,3] through clip_image_preprocess (host)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        ny, nx, _ = rgb.shape
        src = ClipImageU8(nx, ny, rgb.ctypes.data_as(C.POINTER(C.c_uint8)), rgb.size)
        res = ClipImageF32()
        if not lib().clip_image_preprocess(self.ctx, C.byref(src), C.byref(res)):
            raise RuntimeError("Could not preprocess image")
        S = self.vision_config["image_size"]
        out = np.ctypeslib.as_array(res.data, shape=(S, S, 3)).copy()
        lib().clip_image_f32_clean(C.byref(res))
        return out

    @staticmethod
    def _u8_array(images):
        if isinstance(images, np.ndarray) and images.ndim == 4 and images.shape[0] > 0:
            # one contiguous [B, ny, nx, 3] block: the clip_image_u8 array without a Python loop (as encode_images; ~7 us per ctypes object)
            blk = np.ascontiguousarray(images, dtype=np.uint8)
            B = blk.shape[0]
            rec = np.empty(B, dtype=np.dtype([("nx", np.int32), ("ny", np.int32), ("data", np.uint64), ("size", np.uint64)], align=True))
            assert rec.dtype.itemsize == C.sizeof(ClipImageU8)
            rec["nx"], rec["ny"], rec["size"] = blk.shape[2], blk.shape[1], blk[0].size
            rec["data"] = blk.ctypes.data + np.arange(B, dtype=np.uint64) * np.uint64(blk.strides[0])
            return (blk, rec), C.cast(rec.ctypes.data, C.POINTER(ClipImageU8)), B
        keep = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        arr = (ClipImageU8 * len(keep))()
        for i, im in enumerate(keep):
            arr[i] = ClipImageU8(im.shape[1], im.shape[0], im.ctypes.data_as(C.POINTER(C.c_uint8)), im.size)
        return keep, arr, len(keep)

    def encode_images_u8(self, images, normalize=True):
        """list of uint8 [ny,nx,3] raw images (any sizes) -> float32 [n,proj]; resize/crop/normalise run on the GPU
        (clip_amd_image_batch_encode_u8), bit-identical to preprocess() + encode_images()."""
        keep, arr, n = self._u8_array(images)
        out = np.empty((n, self.vision_config["projection_dim"]), dtype=np.float32)
        if not lib().clip_amd_image_batch_encode_u8(self.ctx, arr, n, _fp(out), normalize):
            raise RuntimeError("clip_amd_image_batch_encode_u8 failed (see stderr)")
        return out

    def zero_shot_label_images(self, images, labels):
        """Batched clip_zero_shot_label_image on the GPU: list of uint8 [ny,nx,3] images x list of label strings ->
        (scores [B,n] sorted descending, indices [B,n])."""
        keep, arr, B = self._u8_array(images)
        n = len(labels)
        lab = (C.c_char_p * n)(*[l.encode("utf-8") for l in labels])
        scores = np.empty((B, n), dtype=np.float32)
        idx = np.empty((B, n), dtype=np.int32)
        if not lib().clip_amd_zero_shot_label_images(self.ctx, arr, B, lab, n, _fp(scores), idx.ctypes.data_as(C.POINTER(C.c_int))):
            raise RuntimeError("clip_amd_zero_shot_label_images failed (see stderr)")
        return scores, idx

    def preprocess_device(self, images, d_out_ptr):
        """list of uint8 [ny,nx,3] raw images -> [n,S,S,3] float32 at device address d_out_ptr (asynchronous)."""
        keep, arr, B = self._u8_array(images)
        if not lib().clip_amd_image_batch_preprocess_device(self.ctx, arr, B, C.c_void_p(d_out_ptr)):
            raise RuntimeError("clip_amd_image_batch_preprocess_device failed (see stderr)")
        self.synchronize()   # `keep` (the host pixels) must outlive the copy into the pinned blob — it does: the copy is synchronous

    @property
    def weights_from_cache(self):
        """True when this load read the repacked HBM image from CLIP_AMD_WEIGHT_CACHE instead of repacking the GGUF."""
        return bool(lib().clip_amd_weights_from_cache(self.ctx))

    @property
    def n_devices(self):
        return lib().clip_amd_ctx_device_count(self.ctx)

    def encode_images(self, imgs, normalize=True, n_threads=None):
        """float32 [B,S,S,3] preprocessed images (host) -> float32 [B,proj] via clip_image_batch_encode."""
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        B, S = imgs.shape[0], imgs.shape[1]
        # the clip_image_f32 array, filled without a Python loop (one ctypes object per image costs ~7 us: 1.8 ms per 256 images,
        # a third of the call); layout = struct clip_image_f32 {int nx, ny; float * data; size_t size;} (clip.h:57-64)
        rec = np.empty(B, dtype=np.dtype([("nx", np.int32), ("ny", np.int32), ("data", np.uint64), ("size", np.uint64)], align=True))
        assert rec.dtype.itemsize == C.sizeof(ClipImageF32)
        rec["nx"], rec["ny"], rec["size"] = S, imgs.shape[2], imgs[0].size if B else 0
        rec["data"] = imgs.ctypes.data + np.arange(B, dtype=np.uint64) * np.uint64(imgs.strides[0] if B else 0)
        batch = ClipImageF32Batch(C.cast(rec.ctypes.data, C.POINTER(ClipImageF32)), B)
        out = np.empty((B, self.vision_config["projection_dim"]), dtype=np.float32)
        if not lib().clip_image_batch_encode(self.ctx, n_threads or min(16, os.cpu_count() or 1), C.byref(batch), _fp(out), normalize):
            raise RuntimeError("clip_image_batch_encode failed (see stderr)")
        return out

    def encode_texts(self, token_lists, normalize=True):
        """list of int token lists -> float32 [n,proj] via clip_text_batch_encode (one ragged batch)."""
        n = len(token_lists)
        keep = [np.ascontiguousarray(t, dtype=np.int32) for t in token_lists]
        arr = (ClipTokens * n)()
        for i, t in enumerate(keep):
            arr[i] = ClipTokens(t.ctypes.data_as(C.POINTER(C.c_int32)), t.size)
        out = np.empty((n, self.text_config["projection_dim"]), dtype=np.float32)
        if not lib().clip_text_batch_encode(self.ctx, 1, C.cast(arr, C.POINTER(ClipTokens)), n, _fp(out), normalize):
            raise RuntimeError("clip_text_batch_encode failed (see stderr)")
        return out

    def set_stream(self, stream_handle):
        lib().clip_amd_set_stream(self.ctx, C.c_void_p(stream_handle))

    def set_device_shared(self, shared=True):
        """the caller runs other work on this device concurrently (clip_amd_set_device_shared: the other tower of a two-tower step)."""
        lib().clip_amd_set_device_shared(self.ctx, 1 if shared else 0)

    def synchronize(self):
        lib().clip_amd_synchronize(self.ctx)

    def encode_images_device(self, d_imgs_ptr, batch, d_out_ptr, normalize=True):
        """Device-resident batch encode: raw HBM pointers (ints), asynchronous on the ctx stream."""
        if not lib().clip_amd_image_batch_encode_device(self.ctx, C.c_void_p(d_imgs_ptr), batch, C.c_void_p(d_out_ptr), normalize):
            raise RuntimeError("clip_amd_image_batch_encode_device failed (see stderr)")

    def encode_texts_device(self, d_ids_ptr, offsets, d_out_ptr, normalize=True):
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        if not lib().clip_amd_text_batch_encode_device(self.ctx, C.c_void_p(d_ids_ptr), off.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       off.size - 1, C.c_void_p(d_out_ptr), normalize):
            raise RuntimeError("clip_amd_text_batch_encode_device failed (see stderr)")

    def encode_images_device_multi(self, d_img_ptrs, total, normalize=True, out=None):
        """Multi-GPU handle (n_devices=...): shard g of `total` preprocessed images already on device g at d_img_ptrs[g] (ints); one
        RCCL all-gather of the embeddings; returns the host copy [total, proj] when out is given (np.float32 array), else None."""
        arr = (C.c_void_p * len(d_img_ptrs))(*[C.c_void_p(p) for p in d_img_ptrs])
        if not lib().clip_amd_image_batch_encode_device_multi(self.ctx, arr, total, normalize, _fp(out) if out is not None else None):
            raise RuntimeError("clip_amd_image_batch_encode_device_multi failed (see stderr)")
        return out

    def encode_texts_device_multi(self, d_ids_ptrs, offsets, normalize=True, out=None):
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        arr = (C.c_void_p * len(d_ids_ptrs))(*[C.c_void_p(p) for p in d_ids_ptrs])
        if not lib().clip_amd_text_batch_encode_device_multi(self.ctx, arr, off.ctypes.data_as(C.POINTER(C.c_int32)), off.size - 1, normalize,
                                                             _fp(out) if out is not None else None):
            raise RuntimeError("clip_amd_text_batch_encode_device_multi failed (see stderr)")
        return out

    def encode_pair_device_multi(self, d_img_ptrs, n_images, d_ids_ptrs, offsets, normalize=True, out_img=None, out_txt=None):
        """Multi-GPU handle: both towers of a step in one call (clip_amd_encode_pair_device_multi) — per device the vision tower on the
        replica's stream and the text tower on a twin context's stream, ONE all-gather of both towers' rows."""
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        ia = (C.c_void_p * len(d_img_ptrs))(*[C.c_void_p(p) for p in d_img_ptrs])
        ta = (C.c_void_p * len(d_ids_ptrs))(*[C.c_void_p(p) for p in d_ids_ptrs])
        if not lib().clip_amd_encode_pair_device_multi(self.ctx, ia, n_images, ta, off.ctypes.data_as(C.POINTER(C.c_int32)), off.size - 1, normalize,
                                                       _fp(out_img) if out_img is not None else None, _fp(out_txt) if out_txt is not None else None):
            raise RuntimeError("clip_amd_encode_pair_device_multi failed (see stderr)")
        return out_img, out_txt

    def gathered_embeddings_ptr(self, device_index=0):
        return lib().clip_amd_gathered_embeddings(self.ctx, device_index)

    def profile(self, on=True):
        lib().clip_amd_profile_enable(self.ctx, on)

    def profile_report(self, reset=True):
        """dict tag -> dict(launches, ms, flops, bytes) from HIP events recorded around each launch."""
        n = lib().clip_amd_profile_report(self.ctx, None, 0, False)
        buf = C.create_string_buffer(n + 16)
        lib().clip_amd_profile_report(self.ctx, buf, n + 16, reset)
        out = {}
        for line in buf.value.decode().splitlines():
            tag, launches, ms, fl, by = line.split()
            out[tag] = dict(launches=int(launches), ms=float(ms), flops=float(fl), bytes=float(by))
        return out

    def close(self):
        if getattr(self, "ctx", None):
            lib().clip_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gguf_inspect(path):
    """(kv dict, [(tensor name, dims, ggml type)]) of a GGUF file (metadata only; see convert_hf_to_gguf.py)."""
    from .convert_hf_to_gguf import gguf_inspect as _gi
    return _gi(path)


def quantize(fname_inp, fname_out, itype):
    return bool(lib().clip_model_quantize(os.fsencode(fname_inp), os.fsencode(fname_out), int(itype)))


def shard_bounds(total, n_devices, device_index):
    """(lo, hi, rows_per_device) of the multi-GPU clip_image_batch_encode (clip_amd_shard_bounds; pure arithmetic)."""
    lo, hi, per = C.c_int(), C.c_int(), C.c_int()
    lib().clip_amd_shard_bounds(total, n_devices, device_index, C.byref(lo), C.byref(hi), C.byref(per))
    return lo.value, hi.value, per.value
