"""print the per-kernel-family table (roofline.kernels[]) of a bench line:  python tools/kprof_show.py <file.jsonl>"""
import json
import sys

for path in sys.argv[1:]:
    line = [x for x in open(path) if x.startswith("{")][-1]
    o = json.loads(line)
    c, r = o["config"], o["roofline"]
    print(f"# {path}: {o['value'] / 1e6:.1f} M DoF/s, {o['ms_per_step']:.2f} ms per step (assemble {c['assemble_ms']:.1f} + solve {c['solve_ms']:.1f}); "
          f"kernel sum {r['kernel_ms_sum']:.2f} ms, unattributed {r['unattributed_ms']:.2f} ms")
    print(f"# FGMRES {c['fgmres_iters']}, inner {c['inner_iters']}, CG(M_p) {c['cg_mp_iters']} ({c['t_cg_mp_ms']:.2f} ms), CG(S_m) {c['cg_sm_iters']} ({c['t_cg_sm_ms']:.2f} ms), "
          f"A~^-1 {c['t_ainv_ms']:.2f} ms, true relative residual {c['true_rel_residual']:.3e}")
    print(f"{'family':16s} {'launches':>9s} {'ms/step':>9s} {'GB alg.':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s} {'compute':>8s}")
    for k in r["kernels"]:
        print(f"{k['family']:16s} {k['launches_per_step']:9.1f} {k['ms_per_step']:9.3f} {k['algorithmic_bytes'] / 1e9:9.2f} {k['gb_s']:8.0f} {k['hbm_frac']:9.3f} "
              f"{k.get('compute_frac', 0):8.3f}")
