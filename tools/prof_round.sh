set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01j
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01j/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --cpu-cells 0 --tuned 0 > $R/gpurun_out/r01j/bench_under_rocprof.jsonl 2> $R/gpurun_out/r01j/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01j/fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01j/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01j/write -o w -- python $R/bench.py --steps 1 --warmup 0 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01j/write.err
cd $R
find gpurun_out/r01j -name "*.db" | head
for d in kt fetch write; do db=$(find gpurun_out/r01j/$d -name "*.db" | head -1); python tools/rocpd_summary.py $db gpurun_out/r01j/$d; done
python bench.py > gpurun_out/r01j/bench_default.jsonl 2> gpurun_out/r01j/bench_default.err
tail -1 gpurun_out/r01j/bench_default.jsonl | cut -c1-600
find gpurun_out/r01j -name "*.db" -delete
ls -la gpurun_out/r01j
