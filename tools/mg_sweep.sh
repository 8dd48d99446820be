mkdir -p gpurun_out/r02e
for cfg in "1 3 400" "1 4 400" "2 3 400" "2 4 400" "2 6 400" "1 4 0" "2 4 0" "2 6 0" "3 6 0" "3 8 0"; do set -- $cfg
python bench.py --cells 128 --steps 3 --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --ainv 4 --mg-smooth-u $1 --mg-ratio-u $2 --inner-maxit $3 > gpurun_out/r02e/b_nu$1_r$2_m$3.json 2> gpurun_out/r02e/b_nu$1_r$2_m$3.err
python -c "
import json,sys
try:
  d=json.loads(open('gpurun_out/r02e/b_nu$1_r$2_m$3.json').read().strip().splitlines()[-1]); c=d['config']
  print('nu $1 ratio $2 maxit $3', 'ms/step %.1f'%d['ms_per_step'], 'asm %.1f solve %.1f'%(c['assemble_ms'],c['solve_ms']), 'fgmres',c['fgmres_iters'],'mp',c['cg_mp_iters'],'sm',c['cg_sm_iters'],'inner',c['inner_iters'], 'tmp %.1f tsm %.1f tainv %.1f'%(c['t_cg_mp_ms'],c['t_cg_sm_ms'],c['t_ainv_ms']))
except Exception as e: print('nu $1 ratio $2 maxit $3 ERR', e, open('gpurun_out/r02e/b_nu$1_r$2_m$3.err').read()[-300:])
"
done
