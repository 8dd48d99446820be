# usage: tools/prof_round.sh <tag> [quick]   (on the GPU box; writes gpurun_out/<tag>/)
# kernel trace + stats of the default bench, FETCH_SIZE / WRITE_SIZE passes (two launches: warm-up integrates every block,
# the timed one A_uu + rhs only), then the default bench line without the profiler.
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --cylinder-legs 0 > $O/bench_under_rocprof.jsonl 2> $O/kt.err
if [ "$2" != "quick" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --cylinder-legs 0 > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- python $R/bench.py --steps 1 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --cylinder-legs 0 > /dev/null 2> $O/write.err
fi
cd $R
for d in kt fetch write; do db=$(find $O/$d -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/$d; done
if [ "$2" != "quick" ]; then
python bench.py > $O/bench_default.jsonl 2> $O/bench_default.err
tail -1 $O/bench_default.jsonl | cut -c1-600
fi
find $O -name "*.db" -delete
rm -rf $O/kt $O/fetch $O/write
ls -la $O
