mkdir -p gpurun_out/r03t
for cfg in "2 2 4" "1 2 4" "2 1 4" "1 1 4" "3 1 4" "1 3 4" "2 2 3" "2 2 6" "3 3 4"; do
  set -- $cfg
  timeout 200 python bench.py --steps 2 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --mg-smooth-u $1 --mg-post-u $2 --mg-ratio-u $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('pre $1 post $2 ratio $3: step %.1f solve %.1f ainv %.1f fgmres %d inner %d res %.2e' % (d['ms_per_step'], c['solve_ms'], c['t_ainv_ms'], c['fgmres_iters'], c['inner_iters'], c['true_rel_residual']))" | tee -a gpurun_out/r03t/smooth_sweep.txt
done
