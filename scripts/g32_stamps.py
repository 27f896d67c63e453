"""Phase stamps of a -DCLIPAMD_G8_TIMING build (k_gemm32.hip / k_gemm4.hip): per launch, the median cycles of a workgroup's prologue / K loop / epilogue and the
shader clock during the kernel (s_memtime ticks per 100 MHz s_memrealtime tick).  usage: CLIPAMD_G8_STAMPS=FILE python scripts/gemm_bench.py ...; python scripts/g32_stamps.py FILE"""
import sys, collections
cur=None; groups=collections.OrderedDict()
for line in open(sys.argv[1]):
    if line.startswith('#'): cur=line.strip(); groups[cur]=[]; continue
    v=[int(x) for x in line.split()]
    groups[cur].append(v)
for k,rows in groups.items():
    if not rows: continue
    # cols: id s0 s1 s2 s3 s5 s4(real start) s6(real end)
    import statistics as st
    t0=min(r[1] for r in rows); 
    tot=[r[4]-r[1] for r in rows]; loop=[r[3]-r[2] for r in rows]; pro=[r[2]-r[1] for r in rows]; epi=[r[4]-r[3] for r in rows]; ack=[r[5]-r[4] for r in rows]
    clk=[(r[5]-r[1])/max(1,(r[7]-r[6]))*100 for r in rows]   # MHz
    rspan=(max(r[7] for r in rows)-min(r[6] for r in rows))/100.0
    print(k); print("  wgs %d | cycles: total med %d  prologue %d  loop %d  epilogue %d  ack %d | clock med %.0f MHz | first start to last end %.1f us" % (len(rows), st.median(tot), st.median(pro), st.median(loop), st.median(epi), st.median(ack), st.median(clk), rspan))
