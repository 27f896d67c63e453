"""Prints the two result tables of DESIGN.md section 7 from the committed measurement files (profiles/cfg_rNN/*.json, profiles/rNN_bench.json,
profiles/rNN_kernel_stats_by_grid_*.csv, profiles/pmc_traffic.json).  usage: python scripts/design_tables.py r05"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles")
order = ["b32_q4_0_b256", "b32_q4_0_b32", "b32_q4_0_b1", "l14_f16_b256", "l14_f16_b32", "l14_f16_b1", "cfg2_b32_q4_0_b32_img", "cfg3_l14_f16_b256_img",
         "cfg4_l14_q5_1_b128_img", "cfg5_h14_q8_0_b64_img"]
print("| config (`--config`) | embeddings/s (images/s for `*_img`) | ms/step mean | median | binding | executed FLOPs / time / 2.5 PFLOP/s | GPU vs oracle 1 - cos max (img, txt) |")
print("|---|---|---|---|---|---|---|")
for n in order:
    d = json.load(open(os.path.join(P, "cfg_%s" % tag, "%scfg_%s.json" % (tag, n))))
    w, c = d["whole_step_roofline"], d.get("cpu_baseline") or {}
    t = c.get("gpu_vs_cpu_text_1_minus_cos_max")
    print("| `%s` | %.0f | %.3f | %.3f | %s | %.1f %% (%.0f TFLOP/s) | %.1e, %s |" % (n, d["value"], d["ms_per_step"], d["ms_per_step_median"], w["bound"].upper(), 100 * w["frac"],
                                                                               w["achieved_tflops"], c.get("gpu_vs_cpu_1_minus_cos_max", float("nan")), ("%.1e" % t) if t is not None else "-"))
b = json.load(open(os.path.join(P, "%s_bench.json" % tag)))
grid = {}
for r in csv.DictReader(open(os.path.join(P, "%s_kernel_stats_by_grid_b32_q4_0_b256.csv" % tag))):
    grid[(r["Name"].replace(" ", ""), int(r["Workgroups"]))] = float(r["AverageNs"]) / 1e3
pmc = json.load(open(os.path.join(P, "pmc_traffic.json")))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
print()
print("| kernel at M x N x K (role) | launches / step | us per launch: HIP events / rocprofv3 (two-tower trace) | TFLOP/s (/ 2500) | PMC bytes per launch |")
print("|---|---|---|---|---|")
for k, v in list(b["kernels"].items())[:12]:
    name, rest = k.split("/")[0], k.split("/")[1] if "/" in k else ""
    shape = k.split(":")[1]
    us = v["ms_per_step"] / v["launches_per_step"] * 1e3
    g = bench.kernel_grid_workgroups(name, shape) if name.startswith("gemm") else None
    if name == "attention":
        M = int(shape.split("x")[0]); g = M; rp = grid.get(("attn_kernel<4,2,4,64>", g)) or grid.get(("attn_kernel<5,2,4,64>", g))
    else:
        rp = grid.get((name, g)) if g else None
    tr = pmc.get("%s@%d" % (name, g), {}).get("hbm_bytes_per_launch") if g else None
    print("| `%s` %s (%s) | %d | %.1f / %s | %s | %s |" % (name, shape.replace("x", " x "), rest.split(":")[0], v["launches_per_step"], us, ("%.1f" % rp) if rp else "-",
                                                        ("%.0f (%.2f)" % (v["tflops"], v["tflops"] / 2500.0)) if v["tflops"] else "-", ("%.1f MB" % (tr / 1e6)) if tr else "-"))
r = b["roofline"]
print("\nroofline:", r["kernel"], r["avg_launch_us"], r["achieved"], r["frac"], "traffic", r["traffic"], "| whole step frac", b["whole_step_roofline"]["frac"],
      "traffic/step", b["whole_step_roofline"].get("traffic_bytes_per_step"))
print("value", b["value"], b["ms_per_step"], "img/s", b["images_per_s_per_gpu"], "txt/s", b["texts_per_s_per_gpu"], "host", b["host_api_images_per_s"], b["host_api_images_per_s_4x_batch_per_call"],
      b["host_api_u8_images_per_s"], b["host_api_u8_images_per_s_4x_batch_per_call"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["chunk4_threads4_images_per_s"])
