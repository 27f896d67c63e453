"""`python bench.py --gpus N` must start by itself for N > 1 (the driver runs it without torch.distributed.run at N = 1 and may do the same
at N = 2 / 4 / 8): the file re-executes itself under torch.distributed.run, one rank per GPU, rank 0 prints the ONE JSON line.  CPU tier:
gloo ranks and `--stub-encoder` (a linear map instead of the HIP towers) — the launcher, the process group, the single all-gather per step,
the barrier brackets, the max-over-ranks time and the JSON contract are bench.py's own code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def run(args, env_drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_SELF_LAUNCHED"), extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)


def one_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_gpus_2_launches_itself_and_prints_one_line():
    d = one_line(run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--preheat", "0", "--batch", "8", "--stub-encoder"]))
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["self_launched"] is True and d["data"].startswith("stub") and d["config"]["parallelism"] == "dp2"


def test_bench_under_an_external_launcher_does_not_relaunch():
    """the driver's N > 1 form: torch.distributed.run starts the ranks; bench.py must take them as they are"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_SELF_LAUNCHED")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29653", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat", "0", "--batch", "8",
                        "--stub-encoder"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]     # (gloo prints a connection banner on stdout; RCCL does not)
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["self_launched"] is False


def test_bench_rejects_a_world_size_that_disagrees_with_gpus():
    r = run(["--gpus", "4", "--steps", "1", "--stub-encoder"], env_drop=(), extra_env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_bench_stub_single_rank():
    d = one_line(run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat", "0", "--batch", "8", "--stub-encoder"]))
    assert d["n_gpus"] == 1 and d["self_launched"] is False


def test_algorithmic_work_is_the_survey_8d_figure():
    """`roofline.achieved` must use SURVEY 8(d)'s per-unit figures: FLOPs per image 8.818 G (ViT-B/32), 162.03 G (ViT-L/14), 334.59 G (ViT-H/14);
    FLOPs per text 2 L N (4 h^2 + 2 h ff) + 4 L N^2 h + 2 h proj (0.606 G at N = 8 for ViT-B/32); weight bytes per vision forward 53.2 MB
    (ViT-B/32 q4_0), 229.8 MB (ViT-L/14 q5_1), 608.6 MB (ViT-L/14 f16), 673.9 MB (ViT-H/14 q8_0); config 2 = 72.5 MB algorithmic bytes."""
    import bench
    V = {"b32": dict(image_size=224, patch_size=32, hidden_size=768, n_intermediate=3072, n_layer=12, projection_dim=512),
         "l14": dict(image_size=224, patch_size=14, hidden_size=1024, n_intermediate=4096, n_layer=24, projection_dim=768),
         "h14": dict(image_size=224, patch_size=14, hidden_size=1280, n_intermediate=5120, n_layer=32, projection_dim=1024)}
    T = {"b32": dict(hidden_size=512, n_intermediate=2048, n_layer=12, projection_dim=512, num_positions=77)}
    for name, gflop in (("b32", 8.818), ("l14", 162.03), ("h14", 334.59)):
        fl, _ = bench.algorithmic_work(V[name], None, "f16", 1, [])
        assert abs(fl / 1e9 - gflop) < 0.01 * gflop / 8.8, (name, fl / 1e9)
    fl, _ = bench.algorithmic_work(None, T["b32"], "q4_0", 0, [8])
    assert abs(fl / 1e9 - 0.606) < 0.002
    for name, ftype, mb in (("b32", "q4_0", 53.2), ("l14", "q5_1", 229.8), ("l14", "f16", 608.6), ("h14", "q8_0", 673.9)):
        _, by1 = bench.algorithmic_work(V[name], None, ftype, 1, [])         # one image; its input and output bytes are subtracted below
        S, proj = V[name]["image_size"], V[name]["projection_dim"]
        wbytes = by1 - (S * S * 3 * 4 + proj * 4)
        assert abs(wbytes / 1e6 - mb) < 0.01 * mb, (name, ftype, wbytes / 1e6)
    _, by = bench.algorithmic_work(V["b32"], None, "q4_0", 32, [])
    assert abs(by / 1e6 - 72.5) < 0.4
    # what the pooled last layer leaves out is reported next to the SURVEY figure, never subtracted from it
    cut = bench.pruned_flops(V["b32"], T["b32"], 256, [40] * 256)
    fl, _ = bench.algorithmic_work(V["b32"], T["b32"], "q4_0", 256, [40] * 256)
    assert 0.04 < cut / fl < 0.08
