# outer residual history of the bench step for several inner tolerances of A~^-1 (is one outer iteration within reach?)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"
  timeout 300 python bench.py --steps 2 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --verbosity 1 $v 2> /tmp/err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); c=d['config']; print(d['ms_per_step'], c['assemble_ms'], c['solve_ms'], c['fgmres_iters'], c['cg_mp_iters'], c['cg_sm_iters'], c['inner_iters'], c['t_ainv_ms'])"
  grep "relative residual" /tmp/err.txt | tail -1
done
