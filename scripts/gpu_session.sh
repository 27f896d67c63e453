#!/bin/bash
# One parametrised GPU session (replaces the per-call scripts of rounds 4-5).  Run through gpurun:
#   gpurun -- 'bash scripts/gpu_session.sh TAG <step> [<step> ...]'      logs: gpurun_out/TAG_<step>.log
# steps (each may be given several times; arguments after ':' are passed on, '+' stands for a space):
#   tests[:pytest+args]        GPU test tier (default: tests/ -m gpu)
#   smoke                      __graft_entry__.smoke()
#   bench[:bench.py+args]      bench.py (JSON line kept as gpurun_out/TAG_bench.json when no args are given)
#   ab:ROUNDS:ENV_A:ENV_B[:bench+args]   interleaved same-box A/B of two environment settings (VAR=value,VAR2=value; '-' = none; values cannot hold ',' or ':' —
#                              CLIP_AMD_TILE_OVERRIDE lists need a hand-written loop) on bench.py
#   gemm:gemm_bench.py+args    isolated GEMM timings (scripts/gemm_bench.py)
#   stamps:VARIANT:gemm+args   per-workgroup phase stamps of a -DCLIPAMD_G8_TIMING variant library (scripts/build_variant.sh) -> scripts/g32_stamps.py
#   round                      scripts/gpu_round.sh TAG (tests + smoke + bench + rocprofv3 traces + PMC passes)
#   configs                    scripts/gpu_configs.sh TAG
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=$1; shift
QUICK="--no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates"
for step in "$@"; do
  IFS=':' read -r what a1 a2 a3 a4 <<< "$step"
  sp() { echo "${1//+/ }"; }
  case "$what" in
    tests) timeout 3000 python -m pytest $(sp "${a1:-tests/+-m+gpu}") -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/${TAG}_tests.log ;;
    smoke) timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log ;;
    bench) if [ -z "${a1:-}" ]; then timeout 1200 python bench.py --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -1 | cut -c1-3000 | tee gpurun_out/${TAG}_bench.log
           else timeout 1200 python bench.py $(sp "$a1") 2>&1 | tail -1 | cut -c1-3000 | tee -a gpurun_out/${TAG}_bench_args.log; fi ;;
    ab) for r in $(seq 1 "$a1"); do for e in "$a2" "$a3"; do
          envs=""; [ "$e" != "-" ] && envs="${e//,/ }"; envs="${envs//@/$PWD}"     # (@ = the repo root: CLIP_AMD_LIB=@/clip_cpp_amd/variants/libclip_X.so)
          echo -n "[$e] " | tee -a gpurun_out/${TAG}_ab.log
          env $envs timeout 900 python bench.py --steps 200 --warmup 20 $QUICK $(sp "${a4:-}") 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_ab.log
        done; done ;;
    gemm) GEMM_ITERS=${GEMM_ITERS:-200} timeout 900 python scripts/gemm_bench.py $(sp "$a1") 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_gemm.log ;;
    stamps) rm -f /tmp/st_$a1.txt
            CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_$a1.so CLIPAMD_G8_STAMPS=/tmp/st_$a1.txt GEMM_ITERS=50 timeout 300 python scripts/gemm_bench.py $(sp "$a2") 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_stamps.log
            python scripts/g32_stamps.py /tmp/st_$a1.txt | cut -c1-220 | tee -a gpurun_out/${TAG}_stamps.log ;;
    round) bash scripts/gpu_round.sh "$TAG" ;;
    configs) bash scripts/gpu_configs.sh "$TAG" ;;
    *) echo "unknown step $what" ;;
  esac
done
