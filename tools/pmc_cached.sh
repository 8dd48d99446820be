# HBM traffic of the assembly kernel with cached geometry blocks: two launches (warm-up: all blocks, timed: A_uu + rhs only);
# the cached launch = 2 x mean - the uncached figure of the --warmup 0 pass (tools/prof_round.sh)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01j
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01j/fetch2 -o f -- python $R/bench.py --steps 1 --warmup 1 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01j/fetch2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01j/write2 -o w -- python $R/bench.py --steps 1 --warmup 1 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01j/write2.err
cd $R
for d in fetch2 write2; do db=$(find gpurun_out/r01j/$d -name "*.db" | head -1); python tools/rocpd_summary.py $db gpurun_out/r01j/$d; done
find gpurun_out/r01j -name "*.db" -delete
grep assemble3 gpurun_out/r01j/fetch2_pmc.csv gpurun_out/r01j/write2_pmc.csv
