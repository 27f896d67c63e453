#!/bin/bash
# kernel trace A/B of the LayerNorm fold (CLIP_AMD_LNFOLD=0 / 1): per-kernel durations from rocprofv3 start/end stamps + inter-kernel gaps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r03t}; CFG=${2:-b32_q4_0_b256_img}
for fold in 1 0; do
  (cd /tmp && CLIP_AMD_LNFOLD=$fold timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_${TAG}_$fold -o tr -- python "$R/bench.py" --config $CFG --steps 10 --warmup 3 --preheat 0.5 --no-cpu-baseline --no-roofline --no-host-api > /tmp/tr_${TAG}_$fold.log 2>&1; tail -1 /tmp/tr_${TAG}_$fold.log | cut -c1-200)
done
python - <<PY | tee gpurun_out/${TAG}_trace_${CFG}.txt
import csv, glob, collections
for fold in (1, 0):
    fs = glob.glob("/tmp/tr_${TAG}_%d/**/*kernel_trace.csv" % fold, recursive=True)
    rows = []
    import shutil
    for f in fs: shutil.copy(f, "gpurun_out/${TAG}_fold%d_kernel_trace.csv" % fold)
    print("files", fs)
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "clipamd" not in k: continue
            k = k.replace("void clipamd::(anonymous namespace)::", "").split("(")[0].replace(" ", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    # last 4 steps' worth: find per-step period by counting im2col launches
    starts = [i for i, r in enumerate(rows) if r[2].startswith("im2col")]
    if len(starts) < 6: print("fold", fold, "too few steps", len(starts)); continue
    a, b = starts[-5], starts[-1]
    seg = rows[a:b]
    nst = 4
    dur = collections.defaultdict(float); cnt = collections.Counter(); gap = collections.defaultdict(float)
    for i, (s, e, k) in enumerate(seg):
        dur[k] += e - s; cnt[k] += 1
        if i + 1 < len(seg): gap[k] += max(0, seg[i + 1][0] - e)
    tot = (seg[-1][1] - seg[0][0]) / nst
    print("== fold=%d  step %.1f us  (sum of kernel durations %.1f, sum of gaps %.1f)" % (fold, tot / 1e3, sum(dur.values()) / nst / 1e3, sum(gap.values()) / nst / 1e3))
    for k in sorted(dur, key=lambda k: -dur[k]):
        print("   %-58s x%-3d avg %8.2f us   gap after avg %6.2f us   per step %8.1f us" % (k[:58], cnt[k] // nst, dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3, (dur[k] + gap[k]) / nst / 1e3))
PY
