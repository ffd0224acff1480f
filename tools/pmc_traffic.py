"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH 'HBM') into per-launch
HBM traffic of the kernels bench.py reports on.  Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE is doubled: on gfx950 it reports half the bytes of 16-byte-per-lane streaming reads (both the LDS-DMA
loads of the GEMM and the float4 loads of the other kernels are of that kind)."""
import collections, csv, glob, json, sys

def load(d, tag):
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == tag:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:96]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if k in write:
        out[k] = {"launches_sampled": fetch[k][1], "fetch_bytes_per_launch": 2 * 1024 * fetch[k][0],
                  "write_bytes_per_launch": 1024 * write[k][0],
                  "hbm_bytes_per_launch": 2 * 1024 * fetch[k][0] + 1024 * write[k][0]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}, indent=1))
