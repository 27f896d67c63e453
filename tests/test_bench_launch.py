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
