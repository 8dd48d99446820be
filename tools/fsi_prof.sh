set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02fsi
mkdir -p $O
cd $R
timeout 300 python tools/fsibench.py --cells 64 --cpu-cells 64 > $O/fsibench64.json 2> $O/fsibench64.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/fsibench.py --cells 128 --cpu-cells 0 > $O/fsibench128.json 2> $O/kt.err
cd $R
db=$(find $O/kt -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/kt
find $O -name "*.db" -delete
ls $O $O/kt
cat $O/fsibench64.json $O/fsibench128.json
grep -i "k_fsi" -r $O/kt/*.csv | head -20
