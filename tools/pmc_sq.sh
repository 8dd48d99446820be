# stall breakdown of the heavy kernels: one rocprofv3 PMC pass over the SQ counters (separate from the HBM passes)
# usage: tools/pmc_sq.sh <tag>
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/sq -o sq -- python $R/bench.py --steps 1 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --cylinder-legs 0 > /dev/null 2> $O/sq.err
cd $R
db=$(find $O/sq -name "*.db" | head -1); python tools/rocpd_summary.py $db $O/sq
find $O -name "*.db" -delete; rm -rf $O/sq
grep -E "assemble3|apply_uu_mf2|spmv_uu_pipe|k_mf_gather<3, float, true" $O/sq_pmc.csv | cut -c1-40,140-400
