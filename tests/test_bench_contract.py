"""bench.py prints ONE JSON line with the contract's fields (incl. `roofline` and `cpu_baseline`); smoke() runs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat", "0.2", "--model", "tiny",
                        "--ftype", "q4_0", "--batch", "16", "--cpu-sample", "4"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                          # exactly one line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # (5 decimals in the line: with the dominant kernel taken per SHAPE, a tiny model's dominant GEMM has both times far below 0.01 us)
    assert rf["other_bound"]["bound"] != rf["bound"] and (rf["t_hbm_us"] > rf["t_mfma_us"]) == (rf["bound"] == "hbm")
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["gpu_vs_cpu_1_minus_cos_max"] <= 1e-3
    assert cb["gpu_vs_cpu_text_1_minus_cos_max"] <= 1e-3          # the text half of the metric is checked against the oracle too
    assert cb["chunk4_threads4_images_per_s"] > 0                 # reference harness form (tests/benchmark.cpp:50-51)
    ws = d["whole_step_roofline"]
    assert ws["bound"] in ("mfma", "hbm") and 0 < ws["frac"] <= 1.0 and ws["algorithmic_flops_per_step"] > 0
    assert d["host_api_images_per_s"] > 0 and d["config"]["name"] == "custom"
    assert "traffic" in rf and "traffic_note" in rf and 0 < rf["frac_8d"] <= 1.0 and rf["frac_8d"] == rf["frac"]       # frac IS the SURVEY 8(d) view since round 4
    assert d["host_api_u8_images_per_s"] > 0 and d["host_api_u8_images_per_s_4x_batch_per_call"] > 0                   # the raw-u8 entry point beside the f32 one
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"] and d["python_gc"].startswith("disabled") and d["self_launched"] is False
    # the contract field counts the FLOPs the kernels execute; the SURVEY-figure view sits under a second key
    assert ws["executed_flops_per_step"] <= ws["algorithmic_flops_per_step"] and ws["frac"] <= ws["frac_survey_flops"]
    if ws["bound"] == "mfma":
        assert abs(ws["frac"] - ws["executed_flops_per_step"] / 2.5e15 / (d["ms_per_step"] * 1e-3)) < 2e-3


def test_bench_collective_path_runs_with_one_rank():
    """The N > 1 code path of bench.py — RCCL process group on the dedicated stream, all_gather_into_tensor of the embeddings, barrier,
    max-reduce of the time — executed with ONE rank (BENCH_FORCE_DIST=1): the 1-GPU box cannot run two ranks, but the calls, their
    stream ordering with the HIP kernels and the gathered buffer are exercised on hardware."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat", "0.2", "--model", "tiny",
                        "--ftype", "q4_0", "--batch", "16", "--no-cpu-baseline", "--no-roofline", "--no-host-api"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"


def test_bench_single_process_multi_gpu_form_runs_with_one_replica():
    """bench.py --single-process: the SURVEY 8(e) implementation on the measured path — clip_amd_model_load_multi, device-resident
    shards, the grouped ncclAllGather (one replica here: CLIP_AMD_MULTI_FORCE_RCCL) — prints the same contract line."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--single-process", "--steps", "3", "--warmup", "1", "--preheat", "0.2", "--model", "tiny",
                        "--ftype", "q4_0", "--batch", "16"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak" and "single-process" in d["config"]["parallelism"]


def test_graft_entry_smoke_runs():
    r = subprocess.run([sys.executable, "__graft_entry__.py", "--smoke"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]

