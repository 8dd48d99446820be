# timeline of the big kernels of a few steps with ifem_tuning::auu_double_buffer on (rocprofv3 kernel trace): when does the fill of the next
# A_uu array run relative to the cell kernel and the solve?   usage (GPU box): bash tools/fill_timeline.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fill_tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 5 --cpu-cells 0 --tuned 0 --extras 0 --fsi 0 --cylinder-legs 0 > /dev/null 2> $O/err.txt
db=$(find $O/kt -name "*.db" | head -1)
python - $db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute('select name, start, "end", stream_id, queue_id from kernels order by start')) if "stream_id" in cols else list(cur.execute('select name, start, "end", 0, 0 from kernels order by start'))
big = [(n, s, e, st, q) for n, s, e, st, q in rows if (e - s) > 3e6 or "fillBuffer" in n and (e - s) > 1e6]
t0 = big[-40][1] if len(big) > 40 else big[0][1]
for n, s, e, st, q in big[-40:]:
    print(f"{(s - t0) / 1e6:9.2f} -> {(e - t0) / 1e6:9.2f} ms  ({(e - s) / 1e6:6.2f})  stream {st} queue {q}  {n[:60]}")
PY
find $O -name "*.db" -delete; rm -rf $O/kt
