#!/bin/bash
# final set of round 5 in ONE session (one box): tests + smoke + bench + traces + PMC (gpu_round.sh), then — with the fresh PMC file in place — the bench line again and every config
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_round.sh r05
cp gpurun_out/r05_pmc_traffic.json profiles/pmc_traffic.json
bash scripts/r5/s12.sh
