"""Mid-size batches: one forward per call vs two half-batches on two streams (CLIP_AMD_SPLIT=min,max; forward.cpp vision_forward_launch).
usage: python scripts/split_bench.py [b32:q4_0 l14:f16]   -> images/s and ms per call, device-resident inputs, hipGraph replay where it applies"""
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
for spec in models:
    arch, ftype = spec.split(":")
    path = synth.cached_model(cache, arch, ftype, text=False, vision=True, seed=1234)
    batches = (2, 4, 8, 16, 24, 32, 48, 64, 96, 128) if arch == "b32" else (2, 4, 8, 16, 32)
    if os.environ.get("SPLIT_BENCH_BATCHES"):
        batches = tuple(int(v) for v in os.environ["SPLIT_BENCH_BATCHES"].split(","))
    for B in batches:
        row = []
        for split in ("0,0", "2,4096,2", "2,4096,3", "2,4096,4"):
            os.environ["CLIP_AMD_SPLIT"] = split
            clip = clip_cpp_amd.Clip(path, verbosity=0, device=0)
            S, proj = clip.vision_config["image_size"], clip.vision_config["projection_dim"]
            stream = torch.cuda.Stream()
            clip.set_stream(stream.cuda_stream)
            imgs = torch.randn((B, S, S, 3), dtype=torch.float32, device="cuda")
            out = torch.empty((B, proj), dtype=torch.float32, device="cuda")
            with torch.cuda.stream(stream):
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.3:
                    clip.encode_images_device(imgs.data_ptr(), B, out.data_ptr(), True)
                    torch.cuda.synchronize()
                reps = (300 if B <= 128 else 60) if arch == "b32" else 60
                t0 = time.perf_counter()
                for _ in range(reps):
                    clip.encode_images_device(imgs.data_ptr(), B, out.data_ptr(), True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
            row.append("%s: %7.3f ms %8.0f img/s" % ("one forward" if split == "0,0" else "%s parts" % split[-1], dt * 1e3, B / dt))
            clip.close()
        print("%s %s B=%-4d | %s" % (arch, ftype, B, " | ".join(row)), flush=True)
