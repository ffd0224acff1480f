"""Summarises one rocprofv3 --pmc pass of SQ counters per kernel (averages over launches).
Usage: python tools/pmc_sq.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    vals = {c: sum(v) / len(v) for c, v in cs.items()}
    if vals.get("SQ_WAVE_CYCLES", 0) < 1e5:
        continue
    print(k)
    print("   " + "  ".join(f"{c}={v:.3g}" for c, v in sorted(vals.items())))
