#!/bin/bash
# north_star target matrix + BASELINE.json configs (single-GPU forms) through bench.py --config; one JSON line each under gpurun_out/,
# every one WITH the cpu_baseline leg (the oracle on a bounded sample: 32 images for ViT-B/32, 4 for ViT-L/14 / H/14), so that each carries
# gpu_vs_cpu_1_minus_cos_max beside its rate.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-cfg}
run() { name=$1; shift; echo "== $name"; timeout 900 python bench.py --no-matrix --config $name "$@" --json-out gpurun_out/${TAG}_$name.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); w=d['whole_step_roofline']; c=d.get('cpu_baseline') or {}; print('%s: %.1f emb/s  %.3f ms/step  whole-step %s-bound frac %.4f  (%.1f TF/s, %.1f GB/s)  host API %s img/s  1-cos max img %s txt %s' % (d['config']['name'], d['value'], d['ms_per_step'], w['bound'], w['frac'], w['achieved_tflops'], w['achieved_gbs'], d.get('host_api_images_per_s'), c.get('gpu_vs_cpu_1_minus_cos_max'), c.get('gpu_vs_cpu_text_1_minus_cos_max')))"; }
run b32_q4_0_b1
run b32_q4_0_b32
run b32_q4_0_b256
run l14_f16_b1
run l14_f16_b32
run l14_f16_b256 --no-host-api
run cfg2_b32_q4_0_b32_img
run cfg3_l14_f16_b256_img --no-host-api
run cfg4_l14_q5_1_b128_img --no-host-api
run cfg5_h14_q8_0_b64_img --no-host-api
