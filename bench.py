#!/usr/bin/env python
"""bench.py — image+text embeddings/sec of the MI355X-native CLIP encoder (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config NAME]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic input that is ALREADY resident in HBM:
`clip_amd_image_batch_encode_device` on B preprocessed images + `clip_amd_text_batch_encode_device` on B
ragged token sequences, both through the C ABI of libclip.so, then (N > 1) ONE RCCL all-gather of the
final embeddings.  Per-GPU work is fixed as N grows ("weak" scaling).  Weights are seeded synthetic
(no real checkpoints exist offline) in the named architecture and file type.

Default workload = the configuration BASELINE.json's metric is quoted on: ViT-B/32 q4_0, batch 256 (+ 256 texts).
`--config` selects the other rows of the north_star target matrix (ViT-B/32 q4_0 and ViT-L/14 f16 at batch 1 / 32 / 256)
and the BASELINE configs 2-5 in their single-GPU forms; see CONFIGS.

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  "roofline"            — dominant kernel (the GEMM instantiation with the largest total time), timed live with HIP events
                          on the launch stream in a second pass of the same K steps
  "whole_step_roofline" — EXECUTED FLOPs (SURVEY 8(d) figure minus the last layer's pruned rows) / algorithmic bytes of the whole step
                          against ms_per_step; the SURVEY-figure view of the same time under frac_survey_flops
  "matrix"              — the other north_star cells, each with gpu_vs_cpu_1_minus_cos_max of rows kept from its timed call
  "cpu_baseline"        — the CPU oracle (restatement of the ggml path; ggml itself is absent) on a bounded sample, all host
                          cores and the reference harness's chunk-of-4 / 4-thread form (tests/benchmark.cpp:50-51), with the
                          GPU-vs-oracle cosine deltas of BOTH towers on that sample
  "host_api_images_per_s" — the drop-in host-pointer API (clip_image_batch_encode from pageable caller buffers), PCIe included.
"""
import argparse
import hashlib
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

# name -> model, file type, images per GPU per step, texts per GPU per step (None = same as images), default steps
CONFIGS = {
    "b32_q4_0_b256": dict(model="b32", ftype="q4_0", batch=256, texts=None, steps=100),   # BASELINE metric (default)
    "b32_q4_0_b32": dict(model="b32", ftype="q4_0", batch=32, texts=None, steps=200),
    "b32_q4_0_b1": dict(model="b32", ftype="q4_0", batch=1, texts=None, steps=400),
    "l14_f16_b256": dict(model="l14", ftype="f16", batch=256, texts=None, steps=5),
    "l14_f16_b32": dict(model="l14", ftype="f16", batch=32, texts=None, steps=40),
    "l14_f16_b1": dict(model="l14", ftype="f16", batch=1, texts=None, steps=200),
    # BASELINE.json configs[1..4] (image encode only), single-GPU forms
    "cfg2_b32_q4_0_b32_img": dict(model="b32", ftype="q4_0", batch=32, texts=0, steps=200),
    "cfg3_l14_f16_b256_img": dict(model="l14", ftype="f16", batch=256, texts=0, steps=5),
    "cfg4_l14_q5_1_b128_img": dict(model="l14", ftype="q5_1", batch=128, texts=0, steps=5),   # 1024 / 8 GPUs
    "cfg5_h14_q8_0_b64_img": dict(model="h14", ftype="q8_0", batch=64, texts=0, steps=5),
    "b32_q4_0_b256_img": dict(model="b32", ftype="q4_0", batch=256, texts=0, steps=100),
}
BITS_PER_WEIGHT = {"f32": 32.0, "f16": 16.0, "q4_0": 4.5, "q4_1": 5.0, "q5_0": 5.5, "q5_1": 6.0, "q8_0": 8.5}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=-1)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="b32_q4_0_b256", choices=sorted(CONFIGS), help="workload preset (default = the BASELINE metric configuration)")
    ap.add_argument("--model", default=None)
    ap.add_argument("--ftype", default=None)
    ap.add_argument("--batch", type=int, default=-1, help="images per GPU per step")
    ap.add_argument("--texts", type=int, default=-1, help="texts per GPU per step (default = batch)")
    ap.add_argument("--vision-only", action="store_true", help="vision tower only (implies --texts 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="images (and texts) of the workload timed on the CPU oracle (default: ~20-30 core-seconds)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true")
    ap.add_argument("--preheat", type=float, default=1.0,
                    help="seconds of untimed stepping BEFORE the W warm-up steps (a cold MI355X needs ~0.5 s of load to reach its "
                         "sustained clocks: measured 7.6 ms/step for the first process on a fresh box vs 5.9 ms once warm)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the text tower behind the vision tower on ONE stream (default: two contexts on two HIP streams, so "
                         "the small text kernels fill the tails of the vision kernels)")
    ap.add_argument("--tower-priority", default="none", choices=["none", "text", "vision"],
                    help="experiment: give one tower's HIP stream high priority (profiles/r03_lnfold_and_text_tiles.txt section 5)")
    ap.add_argument("--single-process", action="store_true",
                    help="SURVEY 8(e) form: ONE process, clip_amd_model_load_multi (a replica context + stream + host thread per GPU), "
                         "device-resident shards, ONE grouped ncclAllGather per tower — run as `python bench.py --gpus N --single-process` "
                         "(no torch.distributed.run); the default N > 1 form is one torch process per GPU")
    ap.add_argument("--python-gc", action="store_true", help="leave CPython's cyclic garbage collector on (default: frozen + disabled for the run)")
    ap.add_argument("--no-matrix", action="store_true", help="default config only: skip the other cells of the north_star matrix")
    ap.add_argument("--matrix", action="store_true", help="run the matrix cells also with a non-default --config")
    ap.add_argument("--json-out", default=None)
    ap.add_argument("--no-rates", action="store_true", help="skip the separate image-only / text-only passes (PMC collection: every launch of the process then belongs to a W + K step)")
    ap.add_argument("--stub-encoder", action="store_true",
                    help="CPU test tier only (tests/test_bench_launch.py): torch CPU tensors, the gloo backend and a trivial stand-in for the two "
                         "towers, so that the launcher, the collective skeleton and the JSON contract of this file run without a GPU; the line says so in `data`")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher (the form the driver uses for N = 1): re-execute this command line under
    torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve).  Rank 0 of the child job prints
    the JSON line on the inherited stdout; this process only forwards the exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode this pool's host driver supports (RCCL over xGMI needs it)
    sys.stdout.flush()
    # stdout of the job carries exactly the JSON line: anything else a rank's libraries print there (gloo's "[Gloo] Rank ... connected"
    # banner in the CPU tier) is forwarded to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    raise SystemExit(proc.wait())


def frac4(x):
    """A roofline fraction for the JSON line: 4 decimals, but never rounded to 0 (tiny test models sit at 1e-5: 3 significant digits there)."""
    return round(x, 4) if x >= 0.01 else float("%.3g" % x)


def kernel_source_sha16():
    """Identity of the kernel sources a PMC traffic file was measured at (profiles/pmc_traffic.json carries it)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "clip_cpp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip") or f in ("kernels.h", "gemm_common.h", "attn_body.h"):    # everything that holds device code
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_grid_workgroups(name, shape):
    """Workgroups of one launch of a tiled GEMM kernel at M x N x K — the key that separates the shapes of one instantiation in the per-grid
    rocprofv3 files of scripts/gpu_round.sh.  None for kernels whose grid does not follow from the name alone (ring / small-M: split-K)."""
    import re
    M, N, K = (int(x) for x in shape.split("x"))
    m = re.match(r"gemm_dma_kernel<\d+,(\d+),(\d+),\d+>", name)
    if m and int(m.group(1)) > 64:
        bm, bn = int(m.group(1)), int(m.group(2))
        return -(-M // bm) * -(-N // bn)
    m = re.match(r"gemm8_kernel<(\d+),\d+>", name)
    if m:
        return -(-M // (32 * int(m.group(1)))) * -(-N // 256)
    m = re.match(r"gemm32_kernel<(\d+),(\d+),\d+>", name)      # k_gemm32.hip: (64 WM) x (64 WN) tiles
    if m:
        return -(-M // (64 * int(m.group(1)))) * -(-N // (64 * int(m.group(2))))
    return None


def weight_only_bytes(d, rep, dom, shape=None):
    """weight bytes per launch of the dominant kernel instantiation: N x K x bits / 8 from its shapes (MxNxK tags) — the 8(d) byte count."""
    tot = n = 0
    for k, v in rep.items():
        if k.split("/")[0] != dom or (shape and k.split(":")[1] != shape):
            continue
        M, N, K = (int(x) for x in k.split(":")[1].split("x"))
        tot += v["launches"] * N * K * d.get("bits_per_weight", 4.5) / 8.0
        n += v["launches"]
    return tot / max(1, n)


def algorithmic_work(vc, tc, ftype, n_img, text_lens):
    """FLOPs = 2 x MACs of the weight GEMMs, attention and patch embedding (elementwise excluded);
    bytes = weight bytes once + inputs + outputs (intermediates are not algorithmic)."""
    bpw = BITS_PER_WEIGHT[ftype] / 8.0
    fl = by = 0.0
    if n_img:
        S, P, h, ff, L, proj = vc["image_size"], vc["patch_size"], vc["hidden_size"], vc["n_intermediate"], vc["n_layer"], vc["projection_dim"]
        Np = (S // P) ** 2
        T = Np + 1
        per_img = 2.0 * Np * h * 3 * P * P + L * (2.0 * T * (4 * h * h + 2 * h * ff) + 4.0 * T * T * h) + 2.0 * h * proj
        fl += n_img * per_img
        wbytes = L * (4 * h * h + 2 * h * ff) * bpw + h * proj * bpw + T * h * bpw + h * 3 * P * P * 2 + (L * (9 * h + ff) + 5 * h) * 4
        by += wbytes + n_img * (S * S * 3 * 4 + proj * 4)
    if len(text_lens):
        h, ff, L, proj = tc["hidden_size"], tc["n_intermediate"], tc["n_layer"], tc["projection_dim"]
        for n in text_lens:
            fl += 2.0 * L * n * (4 * h * h + 2 * h * ff) + 4.0 * L * n * n * h + 2.0 * h * proj
        rows = float(sum(text_lens))
        wbytes = L * (4 * h * h + 2 * h * ff) * bpw + h * proj * bpw + tc["num_positions"] * h * bpw + (L * (9 * h + ff) + 2 * h) * 4
        by += wbytes + rows * (h * bpw + 4) + len(text_lens) * proj * 4     # token-embedding rows actually gathered + ids
    return fl, by


def pruned_flops(vc, tc, n_img, text_lens):
    """FLOPs the library does NOT execute although SURVEY 8(d) counts them: behind the last layer's attention only the pooled row of every
    sequence is needed (class token / last token), so the last out-projection and FFN run on one row per sequence (csrc/forward.cpp
    pooled_tail) whenever a tower has more than 64 token rows.  whole_step_roofline.frac counts executed FLOPs (SURVEY figure minus this)."""
    cut = 0.0
    if n_img:
        h, ff, T = vc["hidden_size"], vc["n_intermediate"], (vc["image_size"] // vc["patch_size"]) ** 2 + 1
        if n_img * T > 64:
            cut += n_img * (T - 1) * 2.0 * (h * h + 2 * h * ff)
    if len(text_lens) and sum(text_lens) > 64:
        h, ff = tc["hidden_size"], tc["n_intermediate"]
        cut += sum(n - 1 for n in text_lens) * 2.0 * (h * h + 2 * h * ff)
    return cut


def matrix_cell(torch, clip_cpp_amd, synth, cache, name, local_rank, steps=None, preheat=0.3, warmup=3):
    """One cell of the north_star target matrix, measured like the headline (inputs resident in HBM, the two towers of a step on two
    HIP streams, K timed steps between synchronisations) but rank-local and outside the headline's timed region."""
    cfg = CONFIGS[name]
    batch = cfg["batch"]
    n_texts = batch if cfg["texts"] is None else cfg["texts"]
    steps = steps or cfg["steps"]
    path = synth.cached_model(cache, cfg["model"], cfg["ftype"], text=n_texts > 0, vision=True, seed=1234)
    clip = clip_cpp_amd.Clip(path, verbosity=0, device=local_rank)
    vc, tc = clip.vision_config, clip.text_config
    if n_texts == 0:
        tc = dict(tc, num_positions=77)
    S, proj = vc["image_size"], vc["projection_dim"]
    stream = torch.cuda.Stream()
    clip.set_stream(stream.cuda_stream)
    clip_t = tstream = None
    if n_texts:
        clip_t = clip_cpp_amd.Clip(path, verbosity=0, device=local_rank)
        tstream = torch.cuda.Stream()
        clip_t.set_stream(tstream.cuda_stream)
    g = torch.Generator(device="cuda")
    g.manual_seed(4242)
    imgs = torch.randn((batch, S, S, 3), dtype=torch.float32, device="cuda", generator=g)
    texts = synth.token_ids(n_texts, seed=11, min_len=1, max_len=min(75, tc["num_positions"] - 2))
    flat = np.concatenate(texts).astype(np.int32) if n_texts else np.zeros(1, np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    d_ids = torch.from_numpy(flat).cuda()
    emb = torch.empty((batch + n_texts, proj), dtype=torch.float32, device="cuda")

    def step():
        with torch.cuda.stream(stream):
            if tstream is not None:
                tstream.wait_stream(stream)
            clip.encode_images_device(imgs.data_ptr(), batch, emb[:batch].data_ptr(), True)
            if n_texts:
                clip_t.encode_texts_device(d_ids.data_ptr(), offsets, emb[batch:].data_ptr(), True)
                stream.wait_stream(tstream)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preheat:
        step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record(stream)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sm = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    assert bool(torch.isfinite(emb).all()), "non-finite embeddings in matrix cell %s" % name
    lens = [len(t) for t in texts]
    fl, by = algorithmic_work(vc, tc, cfg["ftype"], batch, lens)
    fl_exec = fl - (pruned_flops(vc, tc, batch, lens) if os.environ.get("CLIP_AMD_PRUNE_LAST", "1") != "0" else 0.0)
    ms = dt / steps * 1e3
    t_mfma, t_hbm = fl_exec / (MFMA_F16_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)
    t_mfma_survey = fl / (MFMA_F16_PEAK_TFLOPS * 1e12)
    # whole_step_frac counts the FLOPs the kernels EXECUTE (the last layer behind the attention runs on the pooled rows only);
    # whole_step_frac_survey_flops is the same time against the SURVEY 8(d) per-item figure
    out = {"name": name, "value": round((batch + n_texts) * steps / dt, 1), "unit": "embeddings/s" if n_texts else "images/s",
           "ms_per_step": round(ms, 4), "ms_per_step_median": round(sm[len(sm) // 2], 4), "steps": steps, "bound": "mfma" if t_mfma >= t_hbm else "hbm",
           "t_bound_us": round(max(t_mfma, t_hbm) * 1e6, 2), "whole_step_frac": frac4(max(t_mfma, t_hbm) / (ms * 1e-3)),
           "whole_step_frac_survey_flops": frac4(max(t_mfma_survey, t_hbm) / (ms * 1e-3))}
    # rows of THIS cell's timed call kept for the oracle comparison of the cpu_baseline leg (first / last image, first / last text; the
    # wide models two rows, the narrow ones four): the comparison runs after every timed region, never inside one
    k = 2 if vc["hidden_size"] >= 1024 else 4
    rows = sorted(set([0, batch - 1] if k == 2 else [0, batch // 3, (2 * batch) // 3, batch - 1]))
    trows = sorted(set([0, n_texts - 1])) if n_texts else []
    out["_sample"] = {"path": path, "rows": rows, "imgs": imgs[rows].cpu().numpy(), "got": emb[:batch][rows].cpu().numpy(),
                      "trows": trows, "texts": [texts[i] for i in trows], "got_t": emb[batch:][trows].cpu().numpy() if trows else None}
    clip.close()
    if clip_t is not None:
        clip_t.close()
    del imgs, emb
    torch.cuda.empty_cache()
    return out


def matrix_oracle_deltas(ref, cells):
    """cpu_baseline leg: 1 - cos between the rows a matrix cell kept from its timed call and the oracle (ggml-faithful numerics) on the same
    bytes.  One oracle model per GGUF; identical input rows (the cells of one model share their first image) are encoded once."""
    import hashlib as _h
    models, memo = {}, {}
    cores = ref.host_cores()
    for c in cells:
        sm = c.get("_sample")
        if not sm:
            continue
        orc = models.get(sm["path"])
        if orc is None:
            orc = models[sm["path"]] = ref.OracleModel(sm["path"])
        worst = 0.0
        for i in range(len(sm["rows"])):
            key = (sm["path"], _h.sha256(sm["imgs"][i].tobytes()).hexdigest())
            if key not in memo:
                memo[key] = orc.image_batch_encode(sm["imgs"][i:i + 1], normalize=True, mode=ref.MODE_FAITHFUL, n_threads=cores)[0]
            worst = max(worst, 1.0 - float((sm["got"][i] * memo[key]).sum()))
        c["gpu_vs_cpu_1_minus_cos_max"] = worst
        c["gpu_vs_cpu_rows"] = sm["rows"]
        if sm["trows"]:
            wt = 0.0
            for i, ids in enumerate(sm["texts"]):
                want = orc.text_encode(ids, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=cores)
                wt = max(wt, 1.0 - float((sm["got_t"][i] * want).sum()))
            c["gpu_vs_cpu_text_1_minus_cos_max"] = wt


# the cells of the north_star matrix that the default run adds behind its timed region: the ViT-B/32 q4_0 column
# on the headline's GGUF, then ViT-L/14 f16 if its synthetic model can be generated within the wall-time guard
MATRIX_B32 = ["b32_q4_0_b1", "b32_q4_0_b32", "cfg2_b32_q4_0_b32_img"]
MATRIX_L14 = ["l14_f16_b1", "l14_f16_b32", "l14_f16_b256"]
L14_GEN_GUARD_S = 60.0


def single_process_main(args, cfg, steps, batch, n_texts, torch, clip_cpp_amd, synth):
    """The C-ABI multi-GPU path on the measured path: one process, N replicas behind clip_amd_model_load_multi, shard g of
    every batch resident on device g, the two towers of a step on two streams per device (replica context + twin context, as the
    one-process-per-GPU form), one grouped ncclAllGather (RCCL over xGMI) of both towers' embeddings per step.  Weak scaling: `batch` images + `n_texts` texts per GPU per step.  The calls are synchronous
    (they return when every replica stream has finished), so the timed region needs no further barrier."""
    N = args.gpus
    if clip_cpp_amd.device_count() < N:
        raise SystemExit("bench.py --single-process --gpus %d: only %d HIP devices visible" % (N, clip_cpp_amd.device_count()))
    cache = os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures")
    path = synth.cached_model(cache, cfg["model"], cfg["ftype"], text=n_texts > 0, vision=True, seed=1234)
    if N == 1:
        os.environ.setdefault("CLIP_AMD_MULTI_FORCE_RCCL", "1")       # one replica still goes through ncclCommInitAll + the grouped all-gather
    clip = clip_cpp_amd.Clip(path, verbosity=0, n_devices=N)
    vc, tc = clip.vision_config, clip.text_config
    if n_texts == 0:
        tc = dict(tc, num_positions=77)
    S, proj = vc["image_size"], vc["projection_dim"]
    imgs, ids, img_ptrs, id_ptrs, all_texts = [], [], [], [], []
    for g in range(N):
        gen = torch.Generator(device="cuda:%d" % g)
        gen.manual_seed(1000 + g)
        t = torch.randn((batch, S, S, 3), dtype=torch.float32, device="cuda:%d" % g, generator=gen)
        imgs.append(t)
        img_ptrs.append(t.data_ptr())
        texts = synth.token_ids(n_texts, seed=11 + g, min_len=1, max_len=min(75, tc["num_positions"] - 2))
        all_texts += texts
        flat = np.concatenate(texts).astype(np.int32) if n_texts else np.zeros(1, np.int32)
        ti = torch.from_numpy(flat).to("cuda:%d" % g)
        ids.append(ti)
        id_ptrs.append(ti.data_ptr())
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in all_texts])]).astype(np.int32)
    for g in range(N):
        torch.cuda.synchronize(g)

    def step():
        if n_texts:      # both towers of the step on two streams per device, one all-gather of both towers' rows
            clip.encode_pair_device_multi(img_ptrs, N * batch, id_ptrs, offsets, True)
        else:
            clip.encode_images_device_multi(img_ptrs, N * batch, True)

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preheat:
        step()
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    out_i = np.empty((N * batch, proj), dtype=np.float32)
    clip.encode_images_device_multi(img_ptrs, N * batch, True, out_i)
    assert np.all(np.isfinite(out_i)), "non-finite embeddings"
    fl_step, by_step = algorithmic_work(vc, tc, cfg["ftype"], batch, [len(t) for t in all_texts[:n_texts]])
    ms_step = dt / steps * 1e3
    fl_exec = fl_step - (pruned_flops(vc, tc, batch, [len(t) for t in all_texts[:n_texts]]) if os.environ.get("CLIP_AMD_PRUNE_LAST", "1") != "0" else 0.0)
    t_mfma_ws, t_hbm_ws = fl_exec / (MFMA_F16_PEAK_TFLOPS * 1e12), by_step / (HBM_PEAK_GBS * 1e9)
    out = {"metric": "image+text embeddings/sec", "value": round(N * (batch + n_texts) * steps / dt, 1), "unit": "embeddings/s", "n_gpus": N,
           "steps": steps, "warmup": args.warmup, "preheat_s": args.preheat, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "CLIP ViT-%s %s: %d images%s per GPU per step, shards resident in HBM, ONE process with a replica context + stream + "
                                  "host thread per GPU (clip_amd_model_load_multi), the two towers of a step on two streams per device "
                                  "(clip_amd_encode_pair_device_multi), one grouped ncclAllGather of the final embeddings per step" % (cfg["model"].upper(), cfg["ftype"], batch,
                                                                                           (" + %d texts" % n_texts) if n_texts else ""),
                      "name": args.config, "images_per_gpu": batch, "texts_per_gpu": n_texts, "parallelism": "dp%d single-process" % N},
           "whole_step_roofline": {"bound": "mfma" if t_mfma_ws >= t_hbm_ws else "hbm", "algorithmic_flops_per_step": fl_step, "executed_flops_per_step": fl_exec,
                                   "algorithmic_bytes_per_step": by_step, "frac": frac4(max(t_mfma_ws, t_hbm_ws) / (ms_step * 1e-3)),
                                   "frac_survey_flops": frac4(max(fl_step / (MFMA_F16_PEAK_TFLOPS * 1e12), t_hbm_ws) / (ms_step * 1e-3)),
                                   "note": "per GPU; frac counts executed FLOPs"},
           "roofline": None, "cpu_baseline": None,
           "parity": "partial (the oracle's op arithmetic is unpinned against ggml — the reference ships no vectors and its ggml submodule is absent; its graph wiring, loader, tokenizer and preprocessing are bit-identical to the reference's own clip.cpp run over oracle/ggml_shim)"}
    line = json.dumps(out)
    print(line, flush=True)
    if args.json_out:
        os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
        with open(args.json_out, "w") as f:
            f.write(line + "\n")
    clip.close()


def timed_steps(args, steps, step, local_step, sync, local_sync, max_over_ranks, mark=None):
    """The driver's timing contract, shared by the HIP path and the CPU-tier stub: untimed preheat (rank-local) + W warm-up steps, then EXACTLY
    K steps between two (barrier + device synchronise) brackets; the time returned is the MAX over ranks."""
    if args.preheat > 0:     # device preconditioning (clocks, code objects, allocator), not part of W or K
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.preheat:   # rank-local work only: iteration counts differ between ranks
            local_step()
            local_sync()
    for _ in range(args.warmup):
        step()
    sync()
    if mark:
        mark(-1)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        if mark:
            mark(i)
    sync()
    return max_over_ranks(time.perf_counter() - t0)


def stub_main(args, cfg, steps, batch, n_texts, torch, dist, rank, world, N):
    """--stub-encoder (CPU test tier): everything of the N-rank form of this file EXCEPT the HIP library — process group (gloo), per-rank
    seeded inputs, the ONE all_gather_into_tensor of the final embeddings per step, barrier brackets, max-over-ranks time, one JSON line
    from rank 0 — with a seeded linear map standing in for the towers.  Also checks the gathered buffer: block r must hold what rank r computes."""
    use_dist = N > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="gloo")
    proj, feat = 64, 192
    gw = torch.Generator()
    gw.manual_seed(7)
    W = torch.randn((feat, proj), generator=gw)

    def inputs(r):
        g = torch.Generator()
        g.manual_seed(1000 + r)
        return torch.randn((batch + n_texts, feat), generator=g)

    def towers(x):
        y = x @ W
        return y / y.norm(dim=1, keepdim=True)

    x = inputs(rank)
    emb = torch.empty((batch + n_texts, proj))
    gathered = torch.empty((N * (batch + n_texts), proj)) if use_dist else None

    def local_step():
        emb.copy_(towers(x))

    def step():
        local_step()
        if use_dist:
            dist.all_gather_into_tensor(gathered, emb)

    def sync():
        if use_dist:
            dist.barrier()

    def max_over_ranks(dt):
        if not use_dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = timed_steps(args, steps, step, local_step, sync, lambda: None, max_over_ranks)
    if use_dist:
        per = batch + n_texts
        for r in range(world if N > 1 else 1):
            assert torch.allclose(gathered[r * per:(r + 1) * per], towers(inputs(r)), atol=1e-6), "all-gather block %d is not rank %d's embeddings" % (r, r)
    if rank == 0:
        out = {"metric": "image+text embeddings/sec", "value": round(N * (batch + n_texts) * steps / dt, 1), "unit": "embeddings/s", "n_gpus": N,
               "steps": steps, "warmup": args.warmup, "preheat_s": args.preheat, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub (CPU test tier: gloo ranks, a linear map instead of the HIP towers; NOT a measurement)",
               "config": {"workload": "stub encoder, %d + %d rows per rank per step" % (batch, n_texts), "name": "stub", "parallelism": "dp%d" % N},
               "roofline": None, "cpu_baseline": None, "self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1"}
        line = json.dumps(out)
        print(line, flush=True)
        if args.json_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
            with open(args.json_out, "w") as f:
                f.write(line + "\n")
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import clip_cpp_amd
    from clip_cpp_amd import synth   # synthetic GGUF through the product's own writer + clip_model_quantize (no oracle/ in the measured path)

    if not args.python_gc:
        # CPython's cyclic collector, with torch's heap behind it, pauses the launching thread for 35-40 ms once every few hundred steps
        # (a full collection; scripts/step_spikes.py, profiles/r03_step_spikes.txt): a 200-step batch-32 cell read 53 k instead of 60 k
        # emb/s when the pause fell into it.  Reference counting still frees everything this process allocates per step.
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()

    cfg = dict(CONFIGS[args.config])
    if args.model: cfg["model"] = args.model
    if args.ftype: cfg["ftype"] = args.ftype
    if args.batch > 0: cfg["batch"] = args.batch
    if args.texts >= 0: cfg["texts"] = args.texts
    if args.vision_only: cfg["texts"] = 0
    steps = args.steps if args.steps > 0 else cfg["steps"]
    batch = cfg["batch"]
    n_texts = batch if cfg["texts"] is None else cfg["texts"]
    vision_only = n_texts == 0
    custom = bool(args.model or args.ftype or args.batch > 0 or args.texts >= 0 or args.vision_only)

    if args.single_process:
        return single_process_main(args, cfg, steps, batch, n_texts, torch, clip_cpp_amd, synth)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N:
        if N > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("BENCH_SELF_LAUNCHED") != "1":
            self_launch(N)                       # does not return
        if N > 1:
            raise SystemExit("bench.py --gpus %d: launched with WORLD_SIZE=%d (torch.distributed.run --nproc-per-node must equal --gpus)" % (N, world))
    stub = args.stub_encoder
    if not stub and (not torch.cuda.is_available() or clip_cpp_amd.device_count() < 1):
        raise SystemExit("bench.py: no HIP device — the HIP path has no CPU fallback")
    if stub:
        return stub_main(args, cfg, steps, batch, n_texts, torch, dist, rank, world, N)
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1: a single rank still initialises the RCCL process group and runs the all-gather / barrier / max-reduce of the
    # N > 1 path (one-rank collectives), so that path executes on a 1-GPU box (tests/test_bench_contract.py)
    use_dist = N > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    cache = os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures")
    path = synth.cached_model(cache, cfg["model"], cfg["ftype"], text=not vision_only, vision=True, seed=1234)
    clip = clip_cpp_amd.Clip(path, verbosity=0, device=local_rank)
    vc, tc = clip.vision_config, clip.text_config
    if vision_only:
        tc = dict(tc, num_positions=77)
    S, proj = vc["image_size"], vc["projection_dim"]
    # a dedicated (non-null) torch stream carries the HIP kernels AND the RCCL all-gather, so they are ordered
    stream = torch.cuda.Stream(priority=-1 if args.tower_priority == "vision" else 0)
    torch.cuda.set_stream(stream)
    clip.set_stream(stream.cuda_stream)
    # the two towers of a step are independent: the text tower gets its own context (own workspace; 2nd copy of the weights)
    # and its own stream, joined back into the main stream before the step ends
    overlap = n_texts > 0 and not args.no_overlap
    if overlap:
        clip_t = clip_cpp_amd.Clip(path, verbosity=0, device=local_rank)
        tstream = torch.cuda.Stream(priority=-1 if args.tower_priority == "text" else 0)
        clip_t.set_stream(tstream.cuda_stream)
    else:
        clip_t, tstream = clip, None

    # synthetic inputs, resident in HBM before the timed region
    g = torch.Generator(device="cuda")
    g.manual_seed(1000 + rank)
    imgs = torch.randn((batch, S, S, 3), dtype=torch.float32, device="cuda", generator=g)
    texts = synth.token_ids(n_texts, seed=11 + rank, min_len=1, max_len=min(75, tc["num_positions"] - 2))
    flat = np.concatenate(texts).astype(np.int32) if n_texts else np.zeros(1, np.int32)
    offsets = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
    d_ids = torch.from_numpy(flat).cuda()
    emb = torch.empty((batch + n_texts, proj), dtype=torch.float32, device="cuda")
    gathered = torch.empty((N * (batch + n_texts), proj), dtype=torch.float32, device="cuda") if use_dist else None
    img_out = emb[:batch]
    txt_out = emb[batch:]

    def share(on):
        # the two towers of a step run concurrently on two contexts: say so (clip_amd_set_device_shared: the heuristic keeps to the kernels
        # that share a CU); the single-tower rates below switch it off again
        if overlap:
            clip.set_device_shared(on)
            clip_t.set_device_shared(on)

    share(True)

    def local_step():
        if overlap:
            tstream.wait_stream(stream)          # fork: the text tower of this step starts with the vision tower
        clip.encode_images_device(imgs.data_ptr(), batch, img_out.data_ptr(), True)
        if n_texts:
            clip_t.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True)
            if overlap:
                stream.wait_stream(tstream)      # join: everything after this point on the main stream sees both towers

    def step():
        local_step()
        if use_dist:
            dist.all_gather_into_tensor(gathered, emb)   # the single RCCL all-gather of the final embeddings

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # device-side step boundaries: one HIP event per step on the main stream (behind the join of the two towers / the all-gather), so the
    # line can carry the MEDIAN step time next to the mean the contract asks for (median of >= 10 repetitions)
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]

    def max_over_ranks(dt):
        if not use_dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = timed_steps(args, steps, step, local_step, sync, torch.cuda.synchronize, max_over_ranks, mark=lambda i: step_events[i + 1].record(stream))
    step_ms = sorted(step_events[i].elapsed_time(step_events[i + 1]) for i in range(steps))
    ms_median = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    per_step_units = N * (batch + n_texts)
    value = per_step_units * steps / dt
    assert bool(torch.isfinite(emb).all()), "non-finite embeddings"

    # separate image-only / text-only rates (rank-local, informative): one tower at a time = the device is not shared
    def rate(fn, units):
        fn(); torch.cuda.synchronize()
        n = max(3, steps // 2)
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return units * n / (time.perf_counter() - t)

    img_rate = txt_rate = 0.0
    if not args.no_rates:
        share(False)
        img_rate = rate(lambda: clip.encode_images_device(imgs.data_ptr(), batch, img_out.data_ptr(), True), batch)
        txt_rate = rate(lambda: clip.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True), n_texts) if n_texts else 0.0
        share(True)       # (the per-kernel pass below times the kernels of the STEP: same heuristic)

    # whole-step roofline: algorithmic work of one GPU's step against the measured step time
    fl_step, by_step = algorithmic_work(vc, tc, cfg["ftype"], batch, [len(t) for t in texts])
    ms_step = dt / steps * 1e3
    t_mfma_ws, t_hbm_ws = fl_step / (MFMA_F16_PEAK_TFLOPS * 1e12), by_step / (HBM_PEAK_GBS * 1e9)
    ws_bound = "mfma" if t_mfma_ws >= t_hbm_ws else "hbm"
    # `frac` / `achieved_tflops` count the FLOPs the kernels EXECUTE: behind the last layer's attention only the pooled
    # row of every sequence is computed, so ~6 % of the SURVEY 8(d) per-item FLOPs are never multiplied.  The SURVEY-figure view of the
    # same step time is kept under `frac_survey_flops` / `achieved_tflops_survey_flops` (what `frac` was until round 4).
    fl_exec = fl_step - (pruned_flops(vc, tc, batch, [len(t) for t in texts]) if os.environ.get("CLIP_AMD_PRUNE_LAST", "1") != "0" else 0.0)
    t_mfma_ex = fl_exec / (MFMA_F16_PEAK_TFLOPS * 1e12)
    ws_bound = "mfma" if t_mfma_ex >= t_hbm_ws else "hbm"
    whole = {"bound": ws_bound, "algorithmic_flops_per_step": fl_step, "executed_flops_per_step": fl_exec, "algorithmic_bytes_per_step": by_step,
             "t_mfma_us": round(t_mfma_ex * 1e6, 2), "t_hbm_us": round(t_hbm_ws * 1e6, 2),
             "achieved_tflops": round(fl_exec / (ms_step * 1e-3) / 1e12, 2), "achieved_gbs": round(by_step / (ms_step * 1e-3) / 1e9, 1),
             "frac": frac4(max(t_mfma_ex, t_hbm_ws) / (ms_step * 1e-3)),
             "frac_survey_flops": frac4(max(t_mfma_ws, t_hbm_ws) / (ms_step * 1e-3)),
             "achieved_tflops_survey_flops": round(fl_step / (ms_step * 1e-3) / 1e12, 2),
             "executed_note": ("frac and achieved_tflops use executed_flops_per_step — what the kernels multiply: the last layer's out-projection + FFN run on the "
                               "pooled row of every sequence only (same embeddings); *_survey_flops use the SURVEY 8(d) per-item figure over the same step time")}

    roofline = None
    kernels = None
    if not args.no_roofline and rank == 0:
        clip.profile(True)
        for _ in range(steps):
            clip.encode_images_device(imgs.data_ptr(), batch, img_out.data_ptr(), True)
            if n_texts:
                clip.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True)
        torch.cuda.synchronize()
        rep = clip.profile_report(reset=True)
        clip.profile(False)
        # aggregate by kernel instantiation AND problem shape: the roofline is quoted for the (kernel, shape) with most time.  (Until round 4 the
        # key was the instantiation alone = the rocprofv3 kernel name; since the q/k/v and FFN-up GEMMs of BOTH towers share one 192 x 128
        # instantiation, a per-name average mixes a 12800 x 3072 x 768 launch with a 10290 x 2048 x 512 one.  scripts/gpu_round.sh therefore splits
        # the rocprofv3 kernel trace and the PMC passes by grid size too — `roofline.grid_workgroups` is the key into those files — and the
        # per-name view of the same instantiation stays in `roofline.instantiation_all_shapes`.)
        inst, by_name = {}, {}
        for k, v in rep.items():
            if not (k.startswith("gemm") or k.startswith("skinny_kernel")):     # the weight-GEMM kernels (tiled, ring, small-M)
                continue
            name, shape = k.split("/")[0], k.split(":")[1]
            for tab, key in ((inst, name + ":" + shape), (by_name, name)):
                a = tab.setdefault(key, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, shapes=set(), name=name))
                a["ms"] += v["ms"]; a["launches"] += v["launches"]; a["flops"] += v["flops"]; a["bytes"] += v["bytes"]
                a["shapes"].add(shape)
        if inst:
            dom_key = max(inst, key=lambda k: inst[k]["ms"])
            d = inst[dom_key]
            dom = d["name"]
            dom_shape = next(iter(d["shapes"]))
            grid_wgs = kernel_grid_workgroups(dom, dom_shape)
            d["bits_per_weight"] = BITS_PER_WEIGHT[cfg["ftype"]]
            avg_ms = d["ms"] / d["launches"]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            total_ms = sum(v["ms"] for v in rep.values())
            traffic = None
            traffic_note = "no profiles/pmc_traffic.json"
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc (scripts/gpu_round.sh)
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("_kernel_src_sha16") != kernel_source_sha16():
                        traffic_note = "stale: pmc_traffic.json was measured at kernel sources %s, current %s" % (tj.get("_kernel_src_sha16"), kernel_source_sha16())
                    elif args.config != tj.get("_config", "b32_q4_0_b256") or custom:
                        traffic_note = "pmc_traffic.json was measured on config %s" % tj.get("_config", "b32_q4_0_b256")
                    else:
                        traffic = (tj.get("%s@%d" % (dom, grid_wgs)) if grid_wgs else None) or (tj.get(dom) if len(by_name[dom]["shapes"]) == 1 else None) or {}
                        traffic = traffic.get("hbm_bytes_per_launch")
                        whole["traffic_bytes_per_step"] = tj.get("_whole_step_hbm_bytes")      # PMC sum over every launch of a step (tracked round over round)
                        if whole["traffic_bytes_per_step"]:
                            whole["traffic_over_algorithmic"] = round(whole["traffic_bytes_per_step"] / by_step, 1)
                        traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2 x FETCH_SIZE + WRITE_SIZE (profiles/pmc_traffic.json)"
                except Exception as e:   # noqa: BLE001
                    traffic_note = "unreadable pmc_traffic.json: %s" % e
            # intermediates are not algorithmic, so the bound of a weight GEMM follows from its FLOPs per WEIGHT byte against the
            # ridge (310 FLOP/B): MFMA-bound (algorithmic FLOPs / 2.5 PFLOP/s) above it, HBM-bound (weight bytes / 8 TB/s) below it.  That
            # view is `roofline.frac`; the kernel-level HBM view — whose byte count also holds the activations, outputs and residual rows the
            # launch really moves — is kept under `other_bound` (until r03 `frac` was whichever view took longer).
            fl_l, by_l = d["flops"] / d["launches"], d["bytes"] / d["launches"]
            wb_l = weight_only_bytes(d, rep, dom, dom_shape)
            t_mfma, t_hbm_w = fl_l / (MFMA_F16_PEAK_TFLOPS * 1e12), wb_l / (HBM_PEAK_GBS * 1e9)
            gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            gbs_w = wb_l / (avg_ms * 1e-3) / 1e9
            mfma_view = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": frac4(achieved / MFMA_F16_PEAK_TFLOPS)}
            hbm_w_view = {"bound": "hbm", "achieved": round(gbs_w, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac4(gbs_w / HBM_PEAK_GBS),
                          "note": "weight bytes only (SURVEY 8d)"}
            hbm_k_view = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac4(gbs / HBM_PEAK_GBS),
                          "note": "kernel-level view: weights + activations + outputs + residual rows of the launch"}
            mfma_bound = t_mfma >= t_hbm_w          # FLOPs per weight byte above the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B)
            roofline = dict(mfma_view if mfma_bound else hbm_w_view)
            other = hbm_k_view if mfma_bound else mfma_view
            roofline["frac_8d"] = roofline["frac"]          # (kept for readers of the r03 line: the same number)
            n_all = by_name[dom]
            roofline.update({"kernel": dom + " at " + dom_shape + " (M x N x K; fp16 MFMA weight GEMM; template args as in the rocprofv3 kernel name)",
                             "grid_workgroups": grid_wgs,
                             "instantiation_all_shapes": {"shapes_MxNxK": sorted(n_all["shapes"]), "launches": n_all["launches"],
                                                          "avg_launch_us": round(n_all["ms"] / n_all["launches"] * 1e3, 2),
                                                          "achieved_tflops": round(n_all["flops"] / (n_all["ms"] * 1e-3) / 1e12, 2),
                                                          "note": "what a per-kernel-NAME average (rocprofv3 --stats) of this instantiation shows"},
                             "traffic": traffic, "traffic_note": traffic_note, "avg_launch_us": round(avg_ms * 1e3, 2), "launches": d["launches"],
                             "algorithmic_flops_per_launch": fl_l, "algorithmic_bytes_per_launch": by_l, "weight_bytes_per_launch": wb_l,
                             "t_mfma_us": round(t_mfma * 1e6, 5), "t_hbm_us": round(t_hbm_w * 1e6, 5), "other_bound": other,
                             "shapes_MxNxK": sorted(d["shapes"]), "share_of_kernel_time": round(d["ms"] / total_ms, 3)})
            kernels = {k: {"ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": v["launches"] // steps,
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] and v["ms"] else None}
                       for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:16]}

    host_api = host_api_x4 = host_api_u8 = host_api_u8_x4 = None
    if not args.no_host_api and rank == 0 and N == 1:
        # the drop-in boundary itself: clip_image_batch_encode from the caller's pageable float buffers (H2D + D2H inside the call)
        h_imgs = imgs.cpu().numpy()
        clip.encode_images(h_imgs[: min(batch, 8)])
        reps = 8 if batch >= 64 else 20
        clip.encode_images(h_imgs)
        t = time.perf_counter()
        for _ in range(reps):
            clip.encode_images(h_imgs)
        host_api = round(batch * reps / (time.perf_counter() - t), 1)
        # ... and with 4 x the batch per call: a call of several 256-image chunks overlaps pack + H2D of chunk c+1 with the forward of
        # chunk c (host_pipeline.cpp), which one chunk per call cannot
        host_big = np.concatenate([h_imgs] * 4, axis=0)
        clip.encode_images(host_big)
        reps4 = reps
        t = time.perf_counter()
        for _ in range(reps4):
            clip.encode_images(host_big)
        host_api_x4 = round(4 * batch * reps4 / (time.perf_counter() - t), 1)
        del host_big
        del h_imgs
        # ... and the raw-pixel entry point: clip_amd_image_batch_encode_u8 from pageable u8 images of the model's input size —
        # 150 KB per image over PCIe instead of 602 KB, resize / crop / normalise on the GPU (bit-identical to clip_image_preprocess)
        u8 = np.random.default_rng(5).integers(0, 256, (batch, S, S, 3), dtype=np.uint8)
        clip.encode_images_u8(u8[: min(batch, 8)])
        clip.encode_images_u8(u8)
        t = time.perf_counter()
        for _ in range(reps):
            clip.encode_images_u8(u8)
        host_api_u8 = round(batch * reps / (time.perf_counter() - t), 1)
        u8_big = np.concatenate([u8] * 4, axis=0)
        clip.encode_images_u8(u8_big)
        t = time.perf_counter()
        for _ in range(reps):
            clip.encode_images_u8(u8_big)
        host_api_u8_x4 = round(4 * batch * reps / (time.perf_counter() - t), 1)
        del u8, u8_big

    cpu_baseline = None
    if not args.no_cpu_baseline and rank == 0 and N == 1:
        from oracle import ref
        orc = ref.OracleModel(path)
        big = vc["hidden_size"] >= 1024
        ns = args.cpu_sample if args.cpu_sample > 0 else (4 if big else 32)
        ns = max(1, min(ns, batch))
        nts = min(ns, n_texts)
        h_imgs = imgs[:ns].cpu().numpy()
        cores = ref.host_cores()
        t = time.perf_counter()
        want = orc.image_batch_encode(h_imgs, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=cores)
        want_t = [orc.text_encode(ids, normalize=True, mode=ref.MODE_FAITHFUL, n_threads=cores) for ids in texts[:nts]]
        cdt = time.perf_counter() - t
        # the reference harness's own form (tests/benchmark.cpp:50-51): batches of 4 images, 4 threads
        n4 = min(ns, 4 if big else 8)
        t = time.perf_counter()
        for b0 in range(0, n4, 4):
            orc.image_batch_encode(h_imgs[b0:b0 + 4], normalize=True, mode=ref.MODE_FAITHFUL, n_threads=4)
        c4 = n4 / (time.perf_counter() - t)
        clip.encode_images_device(imgs.data_ptr(), batch, img_out.data_ptr(), True)
        if n_texts:
            clip.encode_texts_device(d_ids.data_ptr(), offsets, txt_out.data_ptr(), True)
        torch.cuda.synchronize()
        got = img_out[:ns].cpu().numpy()
        cosd = 1.0 - (got * want).sum(1)
        cpu_baseline = {"value": round((ns + nts) / cdt, 3), "unit": "embeddings/s", "cores": cores,
                        "kind": "port",
                        "port_form": "SIMD integer dot products of the block-quantised mat-muls (%s; exact int32 block sums, the scalar loop's f32 operations in its order), cache-blocked; bit-identical to the scalar loop (tests/test_oracle_golden.py)" % ref.dot_simd_name(),
                        "sample": "%d images + %d texts of the same workload, oracle in ggml-faithful numerics (CPU restatement of the ggml path; ggml @dd1d575 unavailable; parity UNPINNED against ggml itself)" % (ns, nts),
                        "chunk4_threads4_images_per_s": round(c4, 3),
                        "chunk4_note": "reference harness form (tests/benchmark.cpp:50-51): batches of 4 images on 4 threads, %d images timed" % n4,
                        "gpu_vs_cpu_1_minus_cos_max": float(cosd.max()), "gpu_vs_cpu_1_minus_cos_mean": float(cosd.mean())}
        if nts:
            got_t = txt_out[:nts].cpu().numpy()
            cosd_t = 1.0 - (got_t * np.stack(want_t)).sum(1)
            cpu_baseline["gpu_vs_cpu_text_1_minus_cos_max"] = float(cosd_t.max())
            cpu_baseline["gpu_vs_cpu_text_1_minus_cos_mean"] = float(cosd_t.mean())

    matrix = None
    if rank == 0 and N == 1 and not args.no_matrix and ((args.config == "b32_q4_0_b256" and not custom) or args.matrix):
        matrix = []
        clip.close()
        if overlap:
            clip_t.close()
        del imgs, emb
        torch.cuda.empty_cache()
        for name in MATRIX_B32:
            matrix.append(matrix_cell(torch, clip_cpp_amd, synth, cache, name, local_rank))
        t_gen = time.perf_counter()
        try:
            synth.cached_model(cache, "l14", "f16", text=True, vision=True, seed=1234)
            gen_s = time.perf_counter() - t_gen
        except Exception as e:   # noqa: BLE001
            gen_s = None
            matrix.append({"name": "l14_f16_*", "skipped": "synthetic ViT-L/14 f16 model could not be generated: %s" % e})
        if gen_s is not None and gen_s > L14_GEN_GUARD_S:
            matrix.append({"name": "l14_f16_*", "skipped": "generating the synthetic ViT-L/14 f16 GGUF took %.0f s (> %.0f s guard)" % (gen_s, L14_GEN_GUARD_S)})
        elif gen_s is not None:
            for name in MATRIX_L14:
                matrix.append(matrix_cell(torch, clip_cpp_amd, synth, cache, name, local_rank, steps=3 if name.endswith("b256") else None))

    if matrix and not args.no_cpu_baseline:
        # cpu_baseline leg, continued (the only place bench.py touches oracle/): every matrix cell's kept rows against the oracle in
        # ggml-faithful numerics -> gpu_vs_cpu_1_minus_cos_max per cell.  Behind every timed region.
        from oracle import ref
        matrix_oracle_deltas(ref, matrix)
    for c in matrix or []:
        c.pop("_sample", None)

    if rank == 0:
        workload = "CLIP ViT-%s %s %s: %d images (%dx%d, vision tower)%s per GPU per step, inputs resident in HBM, %s, RCCL all-gather of final embeddings when N>1" % (
            cfg["model"].upper(), cfg["ftype"], "two-tower" if n_texts else "vision tower only", batch, S, S,
            (" + %d texts (1-75 tokens, text tower)" % n_texts) if n_texts else "",
            "towers on two HIP streams (two contexts)" if overlap else ("towers back to back on one stream" if n_texts else "one stream"))
        out = {
            "metric": "image+text embeddings/sec", "value": round(value, 1), "unit": "embeddings/s", "n_gpus": N,
            "steps": steps, "warmup": args.warmup, "preheat_s": args.preheat, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": workload, "name": args.config if not custom else "custom",
                       "weights": "%s %s GGUF, seeded synthetic" % (cfg["model"], cfg["ftype"]), "images_per_gpu": batch, "texts_per_gpu": n_texts,
                       "text_tokens_per_gpu": int(offsets[-1]), "parallelism": "dp%d" % N,
                       "last_layer": ("every row (CLIP_AMD_PRUNE_LAST=0)" if os.environ.get("CLIP_AMD_PRUNE_LAST", "1") == "0" else
                                      "out-projection + FFN of the LAST layer computed for the pooled row of each sequence only — the rows the embeddings are taken from "
                                      "(class token / last token, reference clip.cpp:1426-1431, 1154-1155); identical embeddings; whole_step_roofline.executed_flops_per_step")},
            "images_per_s_per_gpu": round(img_rate, 1), "texts_per_s_per_gpu": round(txt_rate, 1),
            "host_api_images_per_s": host_api,
            "host_api_images_per_s_4x_batch_per_call": host_api_x4,
            "host_api_u8_images_per_s": host_api_u8,
            "host_api_u8_images_per_s_4x_batch_per_call": host_api_u8_x4,
            "ms_per_step_median": round(ms_median, 4), "ms_per_step_min": round(step_ms[0], 4), "ms_per_step_max": round(step_ms[-1], 4),
            "ms_per_step_note": "ms_per_step = wall clock of the K steps / K (the contract); median / min / max = per-step HIP-event deltas on the main stream over the same K steps",
            "python_gc": "enabled" if args.python_gc else "disabled (gc.freeze + gc.disable for the run: a caller's process does not get this; --python-gc leaves it on)",
            "self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1",
            "roofline": roofline, "whole_step_roofline": whole, "cpu_baseline": cpu_baseline, "matrix": matrix, "kernels": kernels,
            "parity": "partial (the oracle's op arithmetic is unpinned against ggml — the reference ships no vectors and its ggml submodule is absent; its graph wiring, loader, tokenizer and preprocessing are bit-identical to the reference's own clip.cpp run over oracle/ggml_shim)",
        }
        line = json.dumps(out)
        print(line, flush=True)
        if args.json_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.json_out)), exist_ok=True)
            with open(args.json_out, "w") as f:
                f.write(line + "\n")
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
