"""Per-kernel SQ counter table from one `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace` pass.
Usage: python tools/pmc_sq_table.py <dir> > profiles/rNN_pmc_sq_pipeline.txt"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:78]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("Per launch (average): cycles = SQ_BUSY_CYCLES / 32 shader engines; LDS = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles;")
print("MFMA = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles; VALU = SQ_ACTIVE_INST_VALU * 4 / 1024 / cycles (4 issue cycles per")
print("wave64 instruction assumed; fp64 and transcendental instructions take longer, so values above 1 occur);")
print("wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of resident wave cycles spent waiting for an instruction's operands).")
print()
print(f"{'kernel':78s} {'launches':>8s} {'k cycles':>9s} {'LDS':>5s} {'MFMA':>5s} {'VALU':>5s} {'wait':>5s} {'bank confl':>10s}")
rows = []
for k, cs in agg.items():
    v = {c: sum(x) / len(x) for c, x in cs.items()}
    n = len(next(iter(cs.values())))
    cyc = v.get("SQ_BUSY_CYCLES", 0) / 32
    if cyc < 3000:
        continue
    rows.append((cyc, k, n, v))
for cyc, k, n, v in sorted(rows, reverse=True):
    lds = v.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc
    valu = v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc
    wait = v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)
    bank = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    print(f"{k:78s} {n:8d} {cyc / 1e3:9.1f} {lds:5.2f} {mfma:5.2f} {valu:5.2f} {wait:5.2f} {bank:10.3f}")
