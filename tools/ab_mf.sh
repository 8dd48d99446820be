# same-box A/B of the matrix-free A_uu kernel: bash tools/ab_mf.sh old new old new   (libraries openifem_amd/lib/libifem_hip_<tag>.so)
cd openifem_amd/lib; cp libifem_hip.so /tmp/keep.so
for v in "$@"; do cp libifem_hip_$v.so libifem_hip.so; echo "== $v"; (cd ../..; timeout 300 python tools/mfbench.py ${MFV:-4} 2>&1 | tail -5); done
cp /tmp/keep.so libifem_hip.so
