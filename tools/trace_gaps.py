"""Summarise a rocprofv3 kernel trace: busy time, idle gaps between consecutive kernels, per-kernel totals."""
import csv, sys, glob, collections
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows))
# keep the last third (steady state)
ev = ev[len(ev) * 2 // 3:]
busy = sum(e - s for s, e, _ in ev)
wall = ev[-1][1] - ev[0][0]
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
print(f"kernels={len(ev)} wall_ms={wall/1e6:.3f} busy_ms={busy/1e6:.3f} idle_ms={(wall-busy)/1e6:.3f}")
gaps_pos = [g for g in gaps if g > 0]
print("gap_us: mean=%.1f median=%.1f max=%.1f n=%d" % (sum(gaps_pos)/max(1,len(gaps_pos))/1e3, sorted(gaps_pos)[len(gaps_pos)//2]/1e3 if gaps_pos else 0, max(gaps_pos)/1e3 if gaps_pos else 0, len(gaps_pos)))
by = collections.defaultdict(lambda: [0, 0])
gap_after = collections.defaultdict(lambda: [0, 0])
for i, (s, e, k) in enumerate(ev):
    by[k][0] += 1; by[k][1] += e - s
    if i + 1 < len(ev):
        gap_after[k][0] += 1; gap_after[k][1] += max(0, ev[i + 1][0] - e)
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    ga = gap_after[k]
    print(f"  {k:60s} n={c:4d} avg_us={t/c/1e3:8.1f} total_ms={t/1e6:7.3f} gap_after_avg_us={ga[1]/max(1,ga[0])/1e3:7.1f}")
