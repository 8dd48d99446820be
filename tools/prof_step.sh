# per-STEP kernel totals of the default bench: rocprofv3 kernel traces of a run with 1 and one with 3 timed steps, differenced
# (set-up, warm-up, eigenvalue estimates and the per-kernel pass behind the timed region cancel).
# usage: tools/prof_step.sh <tag> [extra bench.py flags]    (on the GPU box; writes gpurun_out/<tag>/step_kernels.csv)
TAG=${1:-r03}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for K in 1 3; do
  rocprofv3 --kernel-trace --stats -d $O/ks$K -o k -- python $R/bench.py --steps $K --warmup 1 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 "$@" > $O/step_bench_$K.jsonl 2> $O/ks$K.err
  db=$(find $O/ks$K -name "*.db" | head -1); python $R/tools/rocpd_summary.py $db $O/ks$K > /dev/null
  find $O/ks$K -name "*.db" -delete; rm -rf $O/ks$K
done
cd $R
python - $O <<'PY'
import csv, sys
O = sys.argv[1]
def load(k):
    d = {}
    for r in list(csv.reader(open(f"{O}/ks{k}_by_grid.csv")))[1:]:
        d[(r[0], r[1])] = (int(r[2]), float(r[3]))
    return d
a, b = load(1), load(3)
rows = []
for key, (c3, t3) in b.items():
    c1, t1 = a.get(key, (0, 0.0))
    if c3 > c1:
        rows.append((key[0], key[1], (c3 - c1) / 2, (t3 - t1) / 2))
rows.sort(key=lambda r: -r[3])
with open(f"{O}/step_kernels.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "launches_per_step", "ms_per_step", "avg_ms"])
    for k, g, c, t in rows:
        w.writerow([k, g, f"{c:g}", f"{t:.3f}", f"{t / c:.4f}"])
print("sum of kernel time per step: %.1f ms" % sum(r[3] for r in rows))
for k, g, c, t in rows[:40]:
    print(f"{k[:70]:70s} {g:>10s} {c:7g} {t:9.3f} {t / c:8.4f}")
PY
