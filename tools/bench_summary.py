#!/usr/bin/env python
"""Print the key fields of bench.py JSON lines read from stdin (one per line), prefixed by argv[1:]."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c, r = d["config"], d["roofline"]
    print(*sys.argv[1:], "MDoF/s", round(d["value"] / 1e6, 2), "ms", round(d["ms_per_step"]), "asm", round(c["assemble_ms"]),
          "solve", round(c["solve_ms"]), "its", c["fgmres_iters"], c["cg_mp_iters"], c["cg_sm_iters"], c["inner_iters"],
          "t_mp", round(c["t_cg_mp_ms"]), "t_sm", round(c["t_cg_sm_ms"]), "t_ainv", round(c["t_ainv_ms"]),
          "spmv_ms", round(r["launch_ms"], 2), "GB/s", round(r["achieved"]), "frac", round(r["frac"], 3))
