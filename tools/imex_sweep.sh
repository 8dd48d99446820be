cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02imex
for v in "--inner-maxit 0" "--inner-rel 1e-1" "--inner-rel 1e-3" "--inner-maxit 0 --mg-smooth-u 3"; do
  echo "== $v"; timeout 200 python bench.py --solver insimex --steps 2 $v 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:d['config'][k] for k in ('fgmres_iters','cg_mp_iters','cg_sm_iters','inner_iters','inner_rel','inner_maxit','mg_smooth_u')})"
done
