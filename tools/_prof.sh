cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -o t -- python3 bench.py --mode train --steps 10 --warmup 3 > /dev/null 2>&1
f=$(find gpurun_out/pf -name "*kernel_stats.csv"); cp $f gpurun_out/pf_stats.csv; rm -rf gpurun_out/pf
