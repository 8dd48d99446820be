# HBM traffic and stall counters of the steady-state 128^3 assembly launch alone (no multigrid levels, no solve):
# usage: tools/prof_asm.sh <tag> [variant:waves]     (on the GPU box; writes gpurun_out/<tag>/asm_*)
set -x
TAG=${1:-r03}
V=${2:-2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/akt -o k -- python $R/tools/asmbench.py 128 $V --warm-only > $O/asm_under_rocprof.log 2> $O/akt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/afetch -o f -- python $R/tools/asmbench.py 128 $V --warm-only > /dev/null 2> $O/afetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/awrite -o w -- python $R/tools/asmbench.py 128 $V --warm-only > /dev/null 2> $O/awrite.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/asq -o s -- python $R/tools/asmbench.py 128 $V --warm-only > /dev/null 2> $O/asq.err
cd $R
for d in akt afetch awrite asq; do db=$(find $O/$d -name "*.db" 2>/dev/null | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db $O/asm_$d; done
find $O -name "*.db" -delete
rm -rf $O/akt $O/afetch $O/awrite $O/asq
grep -h assemble3 $O/asm_*_pmc.csv $O/asm_akt_kernel_stats.csv | cut -c1-60,150-400
