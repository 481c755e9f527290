"""Timeline of the LAST iteration in a rocprofv3 kernel (+memory copy) trace: start offset, duration and the idle gap
before every event.  usage: timeline.py <dir> [n_events_per_iter]"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void dfx::", "")[:48]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# iterations are separated by the synth/fill at start; take events after the last big idle gap > 300 us or last N
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tail = ev[-n:]
t0 = tail[0][0]
prev_end = None
busy = 0
for s, e, k in tail:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap_before {gap:7.1f}  {k}")
    prev_end = max(prev_end or e, e)
    busy += e - s
print(f"span {(tail[-1][1] - t0) / 1e6:.3f} ms busy {busy / 1e6:.3f} ms")
