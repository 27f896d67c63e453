"""Summarise the per-workgroup phase stamps of the 8-wave GEMM (variant built with -DCLIPAMD_G8_TIMING, CLIPAMD_G8_STAMPS=<file>)."""
import sys
import numpy as np
blocks = {}
cur = None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = line[1:].strip(); blocks[cur] = []
    else:
        blocks[cur].append([int(v) for v in line.split()])
for k, rows in blocks.items():
    a = np.array(rows, dtype=np.float64)
    if not len(a): continue
    t0 = a[:, 1].min()
    start, loop, epi, end, ack = (a[:, i] - t0 for i in (1, 2, 3, 4, 5))
    rt0 = a[:, 6].min()
    rstart, rend = a[:, 6] - rt0, a[:, 7] - rt0      # 100 MHz real-time counter
    order = np.argsort(start)
    print(k, "| %d workgroups" % len(a))
    print("  prologue (start -> first tile landed)  mean %7.0f cyc   main loop mean %7.0f   epilogue (issue) mean %7.0f   stores acked +%7.0f" % (
        (loop - start).mean(), (epi - loop).mean(), (end - epi).mean(), (ack - end).mean()))
    print("  workgroup start times (cycles): p0 %.0f p25 %.0f p50 %.0f p75 %.0f p100 %.0f ; kernel span %.0f cyc ; realtime span %.2f us" % (
        *np.percentile(start, [0, 25, 50, 75, 100]), ack.max(), (rend.max()) / 100.0))
