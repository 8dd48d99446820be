# stall breakdown of the heavy kernels: one rocprofv3 PMC pass over the SQ counters (separate from the HBM passes)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01j
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $R/gpurun_out/r01j/sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --cpu-cells 0 --tuned 0 > /dev/null 2> $R/gpurun_out/r01j/sq.err
cd $R
db=$(find gpurun_out/r01j/sq -name "*.db" | head -1); python tools/rocpd_summary.py $db gpurun_out/r01j/sq
find gpurun_out/r01j -name "*.db" -delete
grep -E "assemble3|apply_uu_mf2|spmv_planar<1, 1, 32, float|k_spmv_uu<3" gpurun_out/r01j/sq_pmc.csv | cut -c1-40,140-400
