#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import clip_cpp_amd
from clip_cpp_amd import synth
path = synth.cached_model("/tmp/clip_amd_fixtures", "b32", "q4_0", text=False, vision=True)
clip = clip_cpp_amd.Clip(path, device=0)
imgs = np.random.default_rng(0).standard_normal((256, 224, 224, 3), dtype=np.float32)
for i in range(4):
    t = time.perf_counter(); clip.encode_images(imgs, n_threads=8); print("call %d %.2f ms" % (i, (time.perf_counter() - t) * 1e3))
PY
cd /tmp && CLIP_AMD_HOST_SUBCHUNK=128 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ht -o ht -- python /tmp/one.py 2>&1 | grep -E "call|rror" 
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/ht/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("bytes", "?"))))
ks = []
for f in glob.glob("/tmp/ht/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
ks.sort()
# last call: events after the last big gap
copies = sorted(ev)
big = [c for c in copies if "HOST_TO_DEVICE" in c[2] or "H2D" in c[2]]
last = [c for c in copies if c[0] >= big[-2][0] - 1000] if len(big) >= 2 else copies[-6:]
t0 = last[0][0]
print("-- copies of the last call (us relative to its first copy)")
for s, e, n in last:
    print("  %9.1f -> %9.1f  (%7.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
kk = [k for k in ks if k[0] >= t0]
if kk:
    print("-- kernels: first start %.1f us, last end %.1f us, count %d, busy %.1f us" % ((kk[0][0] - t0) / 1e3, (kk[-1][1] - t0) / 1e3, len(kk), sum(e - s for s, e, _ in kk) / 1e3))
    # gaps > 100us between consecutive kernels
    for a, b in zip(kk, kk[1:]):
        if b[0] - a[1] > 100000: print("   gap %.1f us after %s at %.1f us" % ((b[0] - a[1]) / 1e3, a[2], (a[1] - t0) / 1e3))
PY
