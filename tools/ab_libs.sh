# same-box A/B of kernel variants: build each variant into openifem_amd/lib/libifem_hip_<tag>.so (in-tree, travels with gpurun), then
#   gpurun -- 'bash tools/ab_libs.sh A B A B'
# swaps them in one after the other and prints the warm 128^3 assembly kernel time of each (box-to-box spread is 3-5 %, run-to-run
# on one box 0.3 %: differences below 3 % need this).  The shipped library is restored at the end.
cd openifem_amd/lib; cp libifem_hip.so /tmp/keep.so
for v in "$@"; do cp libifem_hip_$v.so libifem_hip.so; echo -n "$v: "; (cd ../..; timeout 300 python tools/asmbench.py 128 ${ASMB:-2} --warm-only 2>&1 | tail -1 | sed 's/.*warm kernel/warm kernel/'); done
cp /tmp/keep.so libifem_hip.so
