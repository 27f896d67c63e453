"""Raw-image path: host preprocessing (clip_image_batch_preprocess, reference clip.cpp:728-1008) + host-pointer encode
versus GPU preprocessing + encode (clip_amd_image_batch_encode_u8), same images.  usage: python scripts/preproc_bench.py [n] [ny nx]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import clip_cpp_amd as cc  # noqa: E402
from oracle import fixtures  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ny, nx = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (375, 500)
path = fixtures.cached_model("/tmp/clip_amd_fixtures", "b32", "q4_0", text=False, vision=True)
clip = cc.Clip(path, device=0)
rng = np.random.default_rng(0)
images = [rng.integers(0, 256, size=(ny, nx, 3), dtype=np.uint8) for _ in range(n)]
L = cc.lib()
S = clip.vision_config["image_size"]


def host_path(threads):
    keep, arr, _n = clip._u8_array(images)
    src = cc.ClipImageU8Batch(C.cast(arr, C.POINTER(cc.ClipImageU8)), n)
    out_arr = (cc.ClipImageF32 * n)()
    dst = cc.ClipImageF32Batch(C.cast(out_arr, C.POINTER(cc.ClipImageF32)), n)
    t0 = time.perf_counter()
    L.clip_image_batch_preprocess(clip.ctx, threads, C.byref(src), C.byref(dst))
    t1 = time.perf_counter()
    vec = np.empty((n, clip.vision_config["projection_dim"]), dtype=np.float32)
    assert L.clip_image_batch_encode(clip.ctx, threads, C.byref(dst), vec.ctypes.data_as(C.POINTER(C.c_float)), True)
    t2 = time.perf_counter()
    for i in range(n):
        L.clip_image_f32_clean(C.byref(out_arr[i]))
    return vec, t1 - t0, t2 - t1


for _ in range(2):
    dev = clip.encode_images_u8(images)
t0 = time.perf_counter()
for _ in range(3):
    dev = clip.encode_images_u8(images)
t_dev = (time.perf_counter() - t0) / 3
threads = min(32, len(os.sched_getaffinity(0)))
host_path(threads)
vec, t_pre, t_enc = host_path(threads)
print("n=%d images %dx%d -> %d: bitwise equal %s" % (n, nx, ny, S, np.array_equal(vec, dev)))
print("host path  (%2d threads): preprocess %.1f ms + host-pointer encode %.1f ms = %.0f img/s" % (threads, t_pre * 1e3, t_enc * 1e3, n / (t_pre + t_enc)))
print("GPU  path (encode_u8)   : %.1f ms total = %.0f img/s   (H2D of %.1f MB raw u8 instead of %.1f MB f32)" % (
    t_dev * 1e3, n / t_dev, n * ny * nx * 3 / 1e6, n * S * S * 12 / 1e6))
d_out = torch.empty((n, S, S, 3), dtype=torch.float32, device="cuda")
clip.preprocess_device(images, d_out.data_ptr())
t0 = time.perf_counter()
for _ in range(3):
    clip.preprocess_device(images, d_out.data_ptr())
print("GPU preprocess only     : %.1f ms = %.0f img/s (incl. host packing + H2D)" % ((time.perf_counter() - t0) / 3 * 1e3, n * 3 / (time.perf_counter() - t0)))
