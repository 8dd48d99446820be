# inner-iteration table of the block preconditioner over the mesh size: plain Krylov inner solves
# (--ainv 3 --mg 0: Jacobi-preconditioned GMRES / CG, round 1) against the multigrid ones (default).  usage: tools/mg_table.sh <outdir>
O=${1:-gpurun_out/mg_table}
mkdir -p $O
for n in 32 64 128; do
  for cfg in "mg" "plain"; do
    if [ $cfg = mg ]; then fl=""; else fl="--ainv 3 --mg 0"; fi
    timeout 400 python bench.py --cells $n --steps 2 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 $fl > $O/b_${cfg}_$n.json 2> $O/b_${cfg}_$n.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/b_${cfg}_$n.json").read().strip().splitlines()[-1]); c = d["config"]
    print("n $n $cfg: ms/step %.1f | FGMRES %d | CG(M_p) %d | CG(S_m) %d | inner A_uu %d | t_mp %.1f t_sm %.1f t_ainv %.1f ms" % (
        d["ms_per_step"], c["fgmres_iters"], c["cg_mp_iters"], c["cg_sm_iters"], c["inner_iters"], c["t_cg_mp_ms"], c["t_cg_sm_ms"], c["t_ainv_ms"]))
except Exception as e:
    print("n $n $cfg: ERR", e)
PY
  done
done
