#!/usr/bin/env python
"""Print the key fields of bench.py JSON lines (one per line) of the files named on the command line (stdin only when none is
named and stdin is not a terminal: a forgotten redirection must not hang a GPU job)."""
import json
import sys

import os
files = [a for a in sys.argv[1:] if os.path.isfile(a)]
tags = [a for a in sys.argv[1:] if not os.path.isfile(a)]
lines = [l for f in files for l in open(f)] if files else ([] if sys.stdin.isatty() else list(sys.stdin))
for line in lines:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c, r = d["config"], d["roofline"]
    print(*tags, "MDoF/s", round(d["value"] / 1e6, 2), "ms", round(d["ms_per_step"]), "asm", round(c["assemble_ms"]),
          "solve", round(c["solve_ms"]), "its", c["fgmres_iters"], c["cg_mp_iters"], c["cg_sm_iters"], c["inner_iters"],
          "t_mp", round(c["t_cg_mp_ms"]), "t_sm", round(c["t_cg_sm_ms"]), "t_ainv", round(c["t_ainv_ms"]),
          "spmv_ms", round(r["launch_ms"], 2), "GB/s", round(r["achieved"]), "frac", round(r["frac"], 3))
