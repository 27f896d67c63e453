"""A/B of per-context switches over batch sizes (device-resident inputs, vision tower): CLIP_AMD_PRUNE_LAST, CLIP_AMD_LNFOLD_CENTRE.
usage: python scripts/prune_bench.py [b32:q4_0 l14:f16]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import clip_cpp_amd  # noqa: E402
from clip_cpp_amd import synth  # noqa: E402

cache = os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures")
models = [a for a in sys.argv[1:] if ":" in a] or ["b32:q4_0"]
VARIANTS = [("default", {}), ("prune off", {"CLIP_AMD_PRUNE_LAST": "0"}), ("centre off", {"CLIP_AMD_LNFOLD_CENTRE": "0"}), ("both off", {"CLIP_AMD_PRUNE_LAST": "0", "CLIP_AMD_LNFOLD_CENTRE": "0"})]
for spec in models:
    arch, ftype = spec.split(":")
    path = synth.cached_model(cache, arch, ftype, text=False, vision=True, seed=1234)
    batches = (2, 4, 8, 16, 32, 64, 128) if arch == "b32" else (1, 2, 4, 8, 32)
    for B in batches:
        row = []
        for name, env in VARIANTS:
            for k in ("CLIP_AMD_PRUNE_LAST", "CLIP_AMD_LNFOLD_CENTRE"):
                os.environ.pop(k, None)
            os.environ.update(env)
            clip = clip_cpp_amd.Clip(path, verbosity=0, device=0)
            S, proj = clip.vision_config["image_size"], clip.vision_config["projection_dim"]
            stream = torch.cuda.Stream()
            clip.set_stream(stream.cuda_stream)
            imgs = torch.randn((B, S, S, 3), dtype=torch.float32, device="cuda")
            out = torch.empty((B, proj), dtype=torch.float32, device="cuda")
            with torch.cuda.stream(stream):
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.25:
                    clip.encode_images_device(imgs.data_ptr(), B, out.data_ptr(), True)
                    torch.cuda.synchronize()
                reps = 300 if arch == "b32" else 80
                t0 = time.perf_counter()
                for _ in range(reps):
                    clip.encode_images_device(imgs.data_ptr(), B, out.data_ptr(), True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
            row.append("%s %7.3f ms" % (name, dt * 1e3))
            clip.close()
        print("%s %s B=%-4d | %s" % (arch, ftype, B, " | ".join(row)), flush=True)
