#!/usr/bin/env python
"""bench.py — image+text embeddings/sec of the MI355X-native CLIP encoder (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic input that is ALREADY resident in HBM:
`clip_amd_image_batch_encode_device` on B preprocessed images + `clip_amd_text_batch_encode_device` on B
ragged token sequences, both through the C ABI of libclip.so, then (N > 1) ONE RCCL all-gather of the
final embeddings.  Per-GPU work is fixed as N grows ("weak" scaling).  Weights are seeded synthetic
(no real checkpoints exist offline) in the ViT-B/32 architecture, q4_0 file type.

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  "roofline"     — dominant kernel (the dequant-GEMM instantiation with the largest total time), timed
                   live with HIP events on the launch stream in a second pass of the same K steps
  "cpu_baseline" — the CPU oracle (restatement of the ggml path; ggml itself is absent) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="b32")
    ap.add_argument("--ftype", default="q4_0")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--texts", type=int, default=-1, help="texts per GPU per step (default = batch)")
    ap.add_argument("--vision-only", action="store_true", help="vision tower only (implies --texts 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32, help="images (and texts) of the workload timed on the CPU oracle: ~20-30 core-seconds")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--preheat", type=float, default=1.0,
                    help="seconds of untimed stepping BEFORE the W warm-up steps (a cold MI355X needs ~0.5 s of load to reach its "
                         "sustained clocks: measured 7.6 ms/step for the first process on a fresh box vs 5.9 ms once warm)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the text tower behind the vision tower on ONE stream (default: two contexts on two HIP streams, so "
                         "the small text kernels fill the tails of the vision kernels)")
    ap.add_argument("--json-out", default=None)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import clip_cpp_amd
    from clip_cpp_amd import synth   # synthetic GGUF through the product's own writer + clip_model_quantize (no oracle/ in the measured path)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N:
        if N > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (N, world))
    if not torch.cuda.is_available() or clip_cpp_amd.device_count() < 1:
        raise SystemExit("bench.py: no HIP device — the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n_texts = 0 if args.vision_only else (args.batch if args.texts < 0 else args.texts)
    cache = os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures")
    path = synth.cached_model(cache, args.model, args.ftype, text=not args.vision_only, vision=True, seed=1234)
    clip = clip_cpp_amd.Clip(path, verbosity=0, device=local_rank)
    vc, tc = clip.vision_config, clip.text_config
    if args.vision_only:
        tc = dict(tc, num_positions=77)
    S, proj = vc["image_size"], vc["projection_dim"]
    # a dedicated (non-null) torch stream carries the HIP kernels AND the RCCL all-gather, so they are ordered
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    clip.set_stream(stream.cuda_stream)
    # the two towers of a step are independent: the text tower gets its own context (own workspace; 2nd copy of the 85 MB of
    # weights) and its own stream, joined back into the main stream before the step ends
    overlap = n_texts > 0 and not args.no_overlap
    if overlap:
        clip_t = clip_cpp_amd.Clip(path, verbosity=0, device=local_rank)
        tstream = torch.cuda.Stream()
        clip_t.set_stream(tstream.cuda_stream)
    else:
        clip_t, tstream = clip, None

    # synthetic inputs, resident in HBM before the timed region
    g = torch.Generator(device="cuda")
    g.manual_seed(1000 + rank)
    imgs = torch.randn((args.batch, S, S, 3), dtype=torch.float32, device="cuda", generator=g)
    texts = synth.token_ids(n_texts, seed=11 + rank, min_len=1, max_len=min(75, tc["num_positions"] - 2))
    flat = np.concatenate(texts).astype(np.int32) if n_texts else np.zeros(1, np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    d_ids = torch.from_numpy(flat).cuda()
    emb = torch.empty((args.batch + n_texts, proj), dtype=torch.float32, device="cuda")
    gathered = torch.empty((N * (args.batch + n_texts), proj), dtype=torch.float32, device="cuda") if N > 1 else None
    img_out = emb[: args.batch]
    txt_out = emb[args.batch:]

    def local_step():
        clip.encode_images_device(imgs.data_ptr(), args.batch, img_out.data_ptr(), True)
        if n_texts:
            clip_t.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True)
            if overlap:
                stream.wait_stream(tstream)      # join: everything after this point on the main stream sees both towers

    def step():
        local_step()
        if N > 1:
            dist.all_gather_into_tensor(gathered, emb)   # the single RCCL all-gather of the final embeddings

    def sync():
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.preheat > 0:     # device preconditioning (clocks, code objects, allocator), not part of W or K
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.preheat:   # rank-local work only: iteration counts differ between ranks
            local_step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if N > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per_step_units = N * (args.batch + n_texts)
    value = per_step_units * args.steps / dt
    assert bool(torch.isfinite(emb).all()), "non-finite embeddings"

    # separate image-only / text-only rates (rank-local, informative)
    def rate(fn, units):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            fn()
        torch.cuda.synchronize()
        return units * max(3, args.steps // 2) / (time.perf_counter() - t)

    img_rate = rate(lambda: clip.encode_images_device(imgs.data_ptr(), args.batch, img_out.data_ptr(), True), args.batch)
    txt_rate = rate(lambda: clip.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True), n_texts) if n_texts else 0.0

    roofline = None
    kernels = None
    if not args.no_roofline and rank == 0:
        clip.profile(True)
        for _ in range(args.steps):
            clip.encode_images_device(imgs.data_ptr(), args.batch, img_out.data_ptr(), True)
            if n_texts:
                clip.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True)
        torch.cuda.synchronize()
        rep = clip.profile_report(reset=True)
        clip.profile(False)
        # aggregate by kernel instantiation (= rocprofv3 kernel name), the roofline is quoted for the one with most time
        inst = {}
        for k, v in rep.items():
            if not k.startswith("gemm"):
                continue
            name = k.split("/")[0]
            a = inst.setdefault(name, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, shapes=set()))
            a["ms"] += v["ms"]; a["launches"] += v["launches"]; a["flops"] += v["flops"]; a["bytes"] += v["bytes"]
            a["shapes"].add(k.split(":")[1])
        if inst:
            dom = max(inst, key=lambda k: inst[k]["ms"])
            d = inst[dom]
            avg_ms = d["ms"] / d["launches"]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            total_ms = sum(v["ms"] for v in rep.values())
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc (scripts/gpu_pmc.sh)
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # the binding roofline of this kernel: whichever of (algorithmic FLOPs / MFMA peak) and (algorithmic bytes / HBM peak) is
            # the longer time.  The f32-residual GEMMs of ViT-B/32 sit at the ridge (37.7 GFLOP vs 128.6 MB per launch: 15.1 vs 16.1 us).
            fl_l, by_l = d["flops"] / d["launches"], d["bytes"] / d["launches"]
            t_mfma, t_hbm = fl_l / (MFMA_F16_PEAK_TFLOPS * 1e12), by_l / (HBM_PEAK_GBS * 1e9)
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            mfma_view = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4)}
            hbm_view = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
            main, other = (hbm_view, mfma_view) if t_hbm > t_mfma else (mfma_view, hbm_view)
            roofline = dict(main)
            roofline.update({"kernel": dom + " (dequant + fp16 MFMA GEMM; WT,BM,BN,EPI)",
                             "traffic": traffic, "avg_launch_us": round(avg_ms * 1e3, 2), "launches": d["launches"],
                             "algorithmic_flops_per_launch": fl_l, "algorithmic_bytes_per_launch": by_l,
                             "t_mfma_us": round(t_mfma * 1e6, 2), "t_hbm_us": round(t_hbm * 1e6, 2), "other_bound": other,
                             "shapes_MxNxK": sorted(d["shapes"]), "share_of_kernel_time": round(d["ms"] / total_ms, 3)})
            kernels = {k: {"ms_per_step": round(v["ms"] / args.steps, 4), "launches_per_step": v["launches"] // args.steps,
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] else None}
                       for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:14]}

    cpu_baseline = None
    if not args.no_cpu_baseline and rank == 0 and N == 1:
        from oracle import ref
        orc = ref.OracleModel(path)
        ns = max(1, args.cpu_sample)
        h_imgs = imgs[:ns].cpu().numpy()
        cores = ref.host_cores()
        t = time.perf_counter()
        want = orc.image_batch_encode(h_imgs, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=cores)
        for ids in (texts[:ns] if n_texts else []):
            orc.text_encode(ids, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=cores)
        cdt = time.perf_counter() - t
        got = img_out[:ns].cpu().numpy()
        clip.encode_images_device(imgs.data_ptr(), args.batch, img_out.data_ptr(), True)
        torch.cuda.synchronize()
        got = img_out[:ns].cpu().numpy()
        cosd = 1.0 - (got * want).sum(1)
        cpu_baseline = {"value": round((2 if n_texts else 1) * ns / cdt, 3), "unit": "embeddings/s", "cores": cores, "kind": "port",
                        "sample": "%d images + %d texts of the same workload, oracle in ggml-faithful numerics (CPU restatement of the ggml path; ggml @dd1d575 unavailable)" % (ns, ns),
                        "gpu_vs_cpu_1_minus_cos_max": float(cosd.max()), "gpu_vs_cpu_1_minus_cos_mean": float(cosd.mean())}

    if rank == 0:
        out = {
            "metric": "image+text embeddings/sec", "value": round(value, 1), "unit": "embeddings/s", "n_gpus": N,
            "steps": args.steps, "warmup": args.warmup, "preheat_s": args.preheat, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "CLIP ViT-%s %s two-tower: %d images (224x224, vision tower) + %d texts (1-75 tokens, text tower) per GPU per step, inputs resident in HBM, %s, RCCL all-gather of final embeddings when N>1"
                                   % (args.model.upper(), args.ftype, args.batch, n_texts, "towers on two HIP streams (two contexts)" if overlap else "towers back to back on one stream"),
                       "weights": "%s %s GGUF, seeded synthetic" % (args.model, args.ftype), "images_per_gpu": args.batch, "texts_per_gpu": n_texts,
                       "text_tokens_per_gpu": int(offsets[-1]), "parallelism": "dp%d" % N},
            "images_per_s_per_gpu": round(img_rate, 1), "texts_per_s_per_gpu": round(txt_rate, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels,
        }
        line = json.dumps(out)
        print(line, flush=True)
        if args.json_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
            with open(args.json_out, "w") as f:
                f.write(line + "\n")
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
