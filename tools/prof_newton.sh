# per-kernel totals of ONE whole Newton loop (InsIM::run_one_step(true), the time_step leg of bench.py) at 128^3: rocprofv3 kernel traces of
# tools/newton_sweep.py with 2 and with 4 loops, differenced (set-up and the first-use work cancel).
# usage: tools/prof_newton.sh <tag>     (on the GPU box; writes gpurun_out/<tag>/newton_kernels.csv)
TAG=${1:-r04n}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for K in 1 2; do
  ARGS=$(for i in $(seq $K); do echo -n "1e-2 "; done)
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kn$K -o k -- python $R/tools/newton_sweep.py 128 $ARGS > $O/newton_$K.txt 2> $O/kn$K.err
  db=$(find $O/kn$K -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/kn$K > /dev/null
  find $O/kn$K -name "*.db" -delete; rm -rf $O/kn$K
done
cd $R
python - $O <<'PY'
import csv, sys
O = sys.argv[1]
def load(k):
    d = {}
    for r in list(csv.reader(open(f"{O}/kn{k}_by_grid.csv")))[1:]:
        d[(r[0], r[1])] = (int(r[2]), float(r[3]))
    return d
a, b = load(1), load(2)
rows = []
for key, (c2, t2) in b.items():
    c1, t1 = a.get(key, (0, 0.0))
    if c2 > c1:
        rows.append((key[0], key[1], (c2 - c1) / 2, (t2 - t1) / 2))
rows.sort(key=lambda r: -r[3])
with open(f"{O}/newton_kernels.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "launches_per_newton_loop", "ms_per_newton_loop", "avg_ms"])
    for k, g, c, t in rows:
        w.writerow([k, g, f"{c:g}", f"{t:.3f}", f"{t / c:.4f}"])
print("sum of kernel time per Newton loop: %.1f ms" % sum(r[3] for r in rows))
for k, g, c, t in rows[:45]:
    print(f"{k[:70]:70s} {g:>10s} {c:7g} {t:9.3f} {t / c:8.4f}")
PY
