"""How many 64-byte segments does the staged A_uu scatter of k_ins_assemble3 touch per cell?  (CPU only: the C++ host mirror's
DoF tables, the block-interleaved layout of ctx.hpp and the lane -> (pair, entry) mapping of assemble3.hip, replayed in numpy.)
The memory-side atomic path retires ~24 G such segments per second whatever the data type, scope, footprint or number of CUs
(profiles/r03_atomics_types.txt), so this count is the kernel's floor.

    python tools/scatter_sim.py [n] [--lex-nodes]

replays, on the n^3 channel mesh of the host mirror,
  rows "col"   : blocks of a row in column order;           rows "cells": in the order (last cell, its tile, first cell, column)
  rows "cells2": (last cell, first cell, column) (setup.hip, round 4)
  tiles "lex"  : tile columns in the element's node order;  tiles "id"  : in the order of the cell's node ids (perm)
  slots "lane" : stage slot = lane;                         slots "rank": the 16 pairs of a matrix row by their position in it;
  slots "aligned": as "rank", every staged row shifted by the position of its first block inside a 64-byte segment (152 lanes per
                   row; the kernel of round 3: two wavefronts per cell, a row's two column tiles scattered at different times)
  slots "full" : the kernel of round 4 -- a matrix row's 27 blocks staged as the image of their memory (blocks by position, image
                 shifted to the row's alignment), 64 lanes per instruction along the image; "full_unaligned": without the shift
and prints segments per cell next to the floor of the row order (one ideal instruction per cell row) and of the layout (820)."""
import os
import sys
from collections import defaultdict

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openifem_amd import host


class Replay:
    """the DoF tables of the n^3 channel mesh of the host mirror and the replay of the scatter on them"""

    def __init__(self, n, lex_nodes=False):
        s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
        s.set_multigrid(False)
        if lex_nodes:
            s.set_node_order(False)  # nodes numbered lexicographically (x fastest) instead of along the Morton curve
        s.setup_host_only(0)
        self.cu, _, _, _ = s.cell_tables()
        self.nc, self.NU = self.cu.shape
        self.nn = int(self.cu.max()) + 1
        cu = self.cu
        self.rank_in_cell = {}  # (cell, node) -> rank of the node's id among the cell's 27: ranks 16..26 are the second MFMA tile's columns
        for c in range(self.nc):
            for k, j in enumerate(np.argsort(cu[c], kind="stable")):
                self.rank_in_cell[(c, cu[c][j])] = k
        self.rowcells = [defaultdict(list) for _ in range(self.nn)]  # row -> column -> cells that hold both
        for c in range(self.nc):
            nd = cu[c]
            for a in nd:
                rc = self.rowcells[a]
                for b in nd:
                    rc[b].append(c)

    def build_rows(self, kind):
        pos, rowptr = [None] * self.nn, np.zeros(self.nn + 1, np.int64)
        for r in range(self.nn):
            rc = self.rowcells[r]
            key = {"col": lambda b: b, "cells": lambda b: (max(rc[b]), self.rank_in_cell[(max(rc[b]), b)] >= 16, min(rc[b]), b),
                   "cells2": lambda b: (max(rc[b]), min(rc[b]), b)}[kind]  # "cells2" (round 4): a row's 27 blocks are scattered together
            cols = sorted(rc.keys(), key=key)
            pos[r] = {b: k for k, b in enumerate(cols)}
            rowptr[r + 1] = rowptr[r] + len(cols)
        return pos, rowptr

    def replay(self, pos, rowptr, cells, tiles, slots):
        cu = self.cu
        floor = tot = inst = 0
        lanes = np.arange(64)
        for c in cells:
            nd = cu[c]
            for a in nd:
                addr = (np.array([rowptr[a] + pos[a][b] for b in nd]) * 72)[:, None] + 8 * np.arange(9)[None, :]
                floor += len(np.unique(addr // 64))
            perm = (list(range(27)) if tiles == "lex" else list(np.argsort(nd, kind="stable"))) + [-1] * 5
            if slots in ("full", "full_unaligned"):
                # round 4: both column tiles of a row tile are integrated together, a matrix row's 27 blocks leave in one piece: blocks in
                # the order of their positions in the row, the row's image shifted to its alignment inside a 64-byte segment, 64 lanes
                # per instruction along the image (4 instructions per row)
                for a in range(27):
                    o = np.sort(np.array([rowptr[nd[a]] + pos[nd[a]][b] for b in nd]) * 72)
                    s0 = ((o[0] // 8) & 7) if slots == "full" else 0
                    img = np.full(256 + 8, -1, np.int64)
                    for k in range(27):
                        img[s0 + 9 * k:s0 + 9 * k + 9] = o[k] + 8 * np.arange(9)
                    for rr in range(4):
                        seg = img[64 * rr:64 * rr + 64]
                        v = seg >= 0
                        if v.any():
                            tot += len(np.unique(seg[v] // 64))
                            inst += 1
                continue
            for ti in range(2):
                for tj in range(2):
                    for r in range(4):
                        off = np.full(64, -1, np.int64)  # per stage slot
                        for g in range(4):
                            a = 16 * ti + g + 4 * r
                            cols = [perm[16 * tj + j] for j in range(16)]
                            p = [pos[nd[a]][nd[b]] if (a < 27 and b >= 0) else None for b in cols]
                            key = [(q << 4 | j) if q is not None else (0x100000 | j) for j, q in enumerate(p)]
                            rank = np.arange(16) if slots == "lane" else np.argsort(np.argsort(key))
                            for j in range(16):
                                if p[j] is not None:
                                    off[16 * g + rank[j]] = (rowptr[nd[a]] + p[j]) * 72
                        if slots in ("static7", "static7row"):
                            # round 4: stage slots compacted per tile (16 or 11 columns per row), instruction = 7 consecutive slots x 9
                            # entries on 63 lanes ("static7row": every matrix row starts a new instruction)
                            ncol = 16 if tj == 0 else 11
                            seq = []
                            for g in range(4):
                                row = [off[16 * g + k] for k in range(ncol)]
                                if slots == "static7row":
                                    row += [-1] * ((-len(row)) % 7)
                                seq += row
                            seq += [-1] * ((-len(seq)) % 7)
                            for k in range(0, len(seq), 7):
                                o = np.repeat(np.array(seq[k:k + 7], np.int64), 9)
                                v = o >= 0
                                if v.any():
                                    tot += len(np.unique((o[v] + 8 * np.tile(np.arange(9), 7)[v]) // 64))
                                    inst += 1
                            continue
                        span = 152 if slots == "aligned" else 144
                        for rr in range(10 if slots == "aligned" else 9):
                            t = lanes + 64 * rr
                            g = t // span
                            gc = np.minimum(g, 3)
                            o0 = off[16 * gc]
                            u = t - g * span - (np.where(o0 >= 0, (o0 // 8) & 7, 0) if slots == "aligned" else 0)
                            ok = (g < 4) & (u >= 0) & (u < 144)
                            uc = np.where(ok, u, 0)
                            o = np.where(ok, off[16 * gc + uc // 9], -1)
                            v = o >= 0
                            if v.any():
                                tot += len(np.unique((o[v] + 8 * (uc % 9)[v]) // 64))
                                inst += 1
        k = len(cells)
        return floor / k, tot / k, inst / k


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 12
    R = Replay(n, "--lex-nodes" in sys.argv)
    rng = np.random.default_rng(1)
    cells = rng.choice(R.nc, size=min(R.nc, 200), replace=False)
    print(f"{n}^3 cells, {len(cells)} sampled; layout floor 729 * 72 / 64 = 820 segments per cell")
    for rows, combos in (("col", (("lex", "lane"), ("id", "lane"))), ("cells", (("id", "rank"), ("id", "aligned"))),
                         ("cells2", (("id", "full_unaligned"), ("id", "full")))):
        pos, rowptr = R.build_rows(rows)
        for tiles, slots in combos:
            fl, seg, inst = R.replay(pos, rowptr, cells, tiles, slots)
            tag = {("col", "lex", "lane"): "  <- rounds 1-2", ("cells", "id", "aligned"): "  <- round 3", ("cells2", "id", "full"): "  <- shipped"}.get((rows, tiles, slots), "")
            print(f"rows {rows:6s} tiles {tiles:3s} slots {slots:14s}: {seg:7.1f} segments per cell ({seg / 729:.3f} per block; floor of this row order {fl:6.1f}), {inst:.0f} atomic instructions{tag}", flush=True)

if __name__ == "__main__":
    main()
