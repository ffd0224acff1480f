#!/bin/bash
# One GPU session that produces the per-round evidence under gpurun_out/ev_<tag>/ (copy what is to be judged into
# profiles/): rocprofv3 kernel stats of the bench command, the two PMC traffic passes, the bench line itself.
#   usage (on the GPU box, from the repo root):  bash tools/evidence.sh r02_a [--skip-pmc]
set -e
TAG=${1:-r02_x}
OUT=gpurun_out/ev_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || true
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
if [ "$2" != "--skip-pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2>> $OUT/rocprof.err || true
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2>> $OUT/rocprof.err || true
  python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/hbm_traffic_per_launch.json > $OUT/hbm_traffic_summary.txt 2>> $OUT/rocprof.err || true
fi
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rm -rf $OUT/trace/*/*.db $OUT/pmc_fetch/*/*.db $OUT/pmc_write/*/*.db 2>/dev/null || true
du -sh $OUT
head -c 600 $OUT/bench.json
