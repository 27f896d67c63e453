"""Host-API single-item latency (the reference's primary calls): clip_text_encode, clip_image_encode, clip_compare_text_and_image."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import clip_cpp_amd as cc  # noqa: E402
from oracle import fixtures  # noqa: E402

for cfg, ft in [("b32", "q4_0"), ("b32", "f16")]:
    clip = cc.Clip(fixtures.cached_model("/tmp/clip_amd_fixtures", cfg, ft), device=0)
    S = clip.vision_config["image_size"]
    img = fixtures.synthetic_images(1, S, seed=1)
    for n_tok in (3, 8, 77):
        ids = [49406] + [5] * (n_tok - 2) + [49407]
        for _ in range(5):
            clip.encode_text(ids)
        t0 = time.perf_counter()
        for _ in range(50):
            clip.encode_text(ids)
        print("%s %s clip_text_encode  N=%2d : %.3f ms" % (cfg, ft, n_tok, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
    for _ in range(5):
        clip.encode_images(img)
    t0 = time.perf_counter()
    for _ in range(50):
        clip.encode_images(img)
    print("%s %s clip_image_encode B=1   : %.3f ms" % (cfg, ft, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
    raw = np.random.default_rng(0).integers(0, 256, size=(375, 500, 3), dtype=np.uint8)
    for _ in range(3):
        clip.encode_images_u8([raw])
    t0 = time.perf_counter()
    for _ in range(30):
        clip.encode_images_u8([raw])
    print("%s %s encode_u8 500x375 B=1   : %.3f ms" % (cfg, ft, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
