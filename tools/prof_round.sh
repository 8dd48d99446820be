set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01i
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01i/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --cpu-cells 0 --tuned 0 > $R/gpurun_out/r01i/bench_under_rocprof.jsonl 2> $R/gpurun_out/r01i/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01i/fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01i/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01i/write -o w -- python $R/bench.py --steps 1 --warmup 0 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01i/write.err
cd $R
find gpurun_out/r01i -name "*.db" | head
for d in kt fetch write; do db=$(find gpurun_out/r01i/$d -name "*.db" | head -1); python tools/rocpd_summary.py $db gpurun_out/r01i/$d; done
python bench.py > gpurun_out/r01i/bench_default.jsonl 2> gpurun_out/r01i/bench_default.err
tail -1 gpurun_out/r01i/bench_default.jsonl | cut -c1-600
find gpurun_out/r01i -name "*.db" -delete
ls -la gpurun_out/r01i
