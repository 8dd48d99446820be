# per-STEP HBM traffic of every kernel family of the default bench: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs) of
# bench.py with 1 and with 3 timed steps, differenced by tools/pmc_families.py -> profiles/pmc_traffic.json "families".
# usage: tools/prof_families.sh <tag>     (on the GPU box; writes gpurun_out/<tag>/)
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for K in 1 3; do
    rocprofv3 --kernel-trace --pmc $C -d $O/pmc_${C}_$K -o p -- python $R/bench.py --steps $K --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --cylinder-legs 0 > /dev/null 2> $O/pmc_${C}_$K.err
    db=$(find $O/pmc_${C}_$K -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/fam_${C}_$K > /dev/null
    find $O/pmc_${C}_$K -name "*.db" -delete; rm -rf $O/pmc_${C}_$K
  done
done
cd $R
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_families.py $O/fam_FETCH_SIZE_1_pmc.csv $O/fam_FETCH_SIZE_3_pmc.csv $O/fam_WRITE_SIZE_1_pmc.csv $O/fam_WRITE_SIZE_3_pmc.csv $O/pmc_traffic.json 128 | tee $O/families.txt
ls -la $O
