"""One forward step's kernel timeline from a rocprofv3 kernel trace (csv): kernels in start order with duration and the idle
gap before each.   python tools/step_timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at each phase-scan frame-sum kernel
starts = [i for i, n in enumerate(names) if "frame_sum" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = starts[-k - 1], starts[-k]
t0 = int(rows[a]["Start_Timestamp"])
busy = gap = 0
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = 0 if prev_end is None else s - prev_end
    busy += e - s
    gap += max(g, 0)
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {g / 1e3:6.1f}  {r['Kernel_Name'][:100]}")
    prev_end = max(e, prev_end or 0)
t1 = int(rows[b]["Start_Timestamp"])
print(f"step {(t1 - t0) / 1e3:.1f} us: kernels {b - a}, busy {busy / 1e3:.1f} us, gaps {gap / 1e3:.1f} us")
