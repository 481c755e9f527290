"""Is the sporadic 6-34 ms result download a property of the copy path or of the host (CPU quota)?  Times 300 x (D2H of 8 MB into
pinned memory + synchronize) with torch, prints the distribution and the cgroup CPU limits of the box."""
import os, time, torch
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e: print(f, "n/a")
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
d = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
h = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
ts = []
for i in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if i % 3 == 0: time.sleep(0.005)
ts.sort(); print("D2H 8 MB ms: min %.3f median %.3f p99 %.3f max %.3f" % (ts[0], ts[150], ts[296], ts[-1]), "n>2ms:", sum(t > 2 for t in ts))
# busy host loop: does the scheduler take the CPU away?
gaps = []; t_prev = time.perf_counter(); end = t_prev + 3.0
while t_prev < end:
    t = time.perf_counter(); gaps.append(t - t_prev); t_prev = t
gaps.sort(); print("busy loop 3 s: max gap %.3f ms, gaps > 1 ms: %d" % (gaps[-1] * 1e3, sum(g > 1e-3 for g in gaps)))
print("cpu.stat after:", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | ") if os.path.exists("/sys/fs/cgroup/cpu.stat") else "n/a")
