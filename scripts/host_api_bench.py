"""Host-pointer API throughput: clip_image_batch_encode from pageable caller buffers (PCIe included).
usage: python scripts/host_api_bench.py [batch] [subchunk ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
try:
    import torch  # noqa: F401
except Exception:
    pass
import clip_cpp_amd
from clip_cpp_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
path = synth.cached_model(os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures"), "b32", "q4_0", text=False, vision=True)
clip = clip_cpp_amd.Clip(path, device=0)
imgs = np.random.default_rng(0).standard_normal((B, 224, 224, 3), dtype=np.float32)
for nt in (1, 4, 8, 16, 32):
    clip.encode_images(imgs, n_threads=nt); clip.encode_images(imgs, n_threads=nt)
    t = time.perf_counter(); reps = 5
    for _ in range(reps):
        clip.encode_images(imgs, n_threads=nt)
    dt = (time.perf_counter() - t) / reps
    print("batch %d n_threads %2d : %.2f ms per call  %.0f img/s" % (B, nt, dt * 1e3, B / dt), flush=True)
