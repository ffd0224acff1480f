#!/bin/bash
# Round-3 evidence session (one GPU call): rocprofv3 kernel stats + PMC passes of the headline bench, kernel stats of BASELINE
# configs #3 (Sins-256 B=64), #4 (training step B=32), #5 (real-time stream) and of the enhancer at the shipped geometry, the
# SQ counter pass of the headline forward, the bench lines themselves.  Everything lands under gpurun_out/ev_<tag>/; copy what
# is to be judged into profiles/.     usage (on the GPU box, repo root):  bash tools/evidence_r3.sh r03_a
set -e
TAG=${1:-r03_x}
OUT=gpurun_out/ev_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
stats() {   # stats <name> <command...>: kernel-trace stats of one command
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -o t -- "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}.err || true
  find $OUT/tr_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  rm -rf $OUT/tr_$name
}
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
stats combsub_b64 $BENCH
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > /dev/null 2>> $OUT/rocprof.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > /dev/null 2>> $OUT/rocprof.err || true
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/hbm_traffic_per_launch.json > $OUT/hbm_traffic_summary.txt 2>> $OUT/rocprof.err || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o pmc -- python3 tools/fwd_loop.py 12 > /dev/null 2>> $OUT/rocprof.err || true
python3 tools/pmc_sq_table.py $OUT/pmc_sq > $OUT/pmc_sq_synth.txt 2>> $OUT/rocprof.err || true
stats sins256_b64 python3 bench.py --model Sins256 --steps 20 --warmup 5 --no-cpu-baseline
stats train_b32 python3 bench.py --mode train --steps 10 --warmup 3
stats realtime python3 bench.py --mode realtime --steps 50 --warmup 10
stats enhancer_860 python3 tools/enhancer_time.py 860
stats causal_b64 python3 tools/causal_time.py 64
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --mode train --steps 20 --warmup 5 > $OUT/bench_train.json 2>> $OUT/bench.err
python3 bench.py --mode train --model Sins256 --steps 20 --warmup 5 > $OUT/bench_train_sins256.json 2>> $OUT/bench.err
python3 bench.py --mode realtime --steps 200 --warmup 20 > $OUT/bench_realtime.json 2>> $OUT/bench.err
python3 tools/secondary_bench.py > $OUT/secondary_configs.json 2>> $OUT/bench.err || true
python3 tools/enhancer_time.py 860 > $OUT/enhancer_time.txt 2>> $OUT/bench.err || true
python3 tools/causal_time.py 64 > $OUT/causal_time.txt 2>> $OUT/bench.err || true
python3 tools/phase_scan_time.py > $OUT/phase_scan_time.txt 2>> $OUT/bench.err || true
rm -rf $OUT/pmc_fetch/*/*.db $OUT/pmc_write/*/*.db $OUT/pmc_sq/*/*.db 2>/dev/null || true
find $OUT -name "*.db" -delete
du -sh $OUT
head -c 400 $OUT/bench.json
