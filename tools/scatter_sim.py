"""How many 64-byte segments does the staged A_uu scatter of k_ins_assemble3 touch per cell?  (CPU only: the C++ host mirror's
DoF tables, the block-interleaved layout of ctx.hpp and the lane -> (pair, entry) mapping of assemble3.hip, replayed in numpy.)
The memory-side atomic path retires ~24 G such segments per second whatever the data type, scope, footprint or number of CUs
(profiles/r03_atomics_types.txt), so this count is the kernel's floor.  Compares column orders of the MFMA tiles.

    python tools/scatter_sim.py [n] [order ...]     order: lex (shipped), morton, sorted (by global node id per cell)"""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openifem_amd import host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
orders = [a for a in sys.argv[2:] if not a.startswith("--")] or ["lex", "morton", "sorted"]
s = host.InsIM(host.channel_prm(3), (n, n, n), (0, 0, 0), (2.0, 0.2, 0.2))
s.set_multigrid(False)
if "--lex-nodes" in sys.argv:
    s.set_node_order(False)  # nodes numbered lexicographically (x fastest) instead of along the Morton curve
s.setup_host_only(0)
cu, _, _, _ = s.cell_tables()
nc, NU = cu.shape
nn = cu.max() + 1
# row patterns: sorted unique neighbours
pairs = np.unique((cu[:, :, None].astype(np.int64) * nn + cu[:, None, :]).ravel())
rows, cols = pairs // nn, pairs % nn
rowptr = np.zeros(nn + 1, np.int64)
np.add.at(rowptr, rows + 1, 1)
rowptr = np.cumsum(rowptr)
key = pairs  # sorted: position of (r, c) = searchsorted(key, r * nn + c)


def block_addr(a_nodes, b_nodes):
    """byte offset of block (a, b) in the block-interleaved value array"""
    return np.searchsorted(key, a_nodes.astype(np.int64) * nn + b_nodes) * 72


lexmorton = sorted(range(27), key=lambda b: (sum(((b // 3 ** d) % 3 >> 1) << d for d in range(3)) << 3 | sum(((b // 3 ** d) % 3 & 1) << d for d in range(3))))


def simulate(order, cells):
    total_seg = total_inst = total_atom = 0
    for c in cells:
        nd = cu[c]
        if order == "lex":
            perm = list(range(27))
        elif order == "morton":
            perm = lexmorton
        else:
            perm = list(np.argsort(nd, kind="stable"))
        perm = perm + [-1] * 5  # 32 tile columns
        for ti in range(2):
            for tj in range(2):
                for r in range(4):
                    a = 16 * ti + (np.arange(64) >> 4) + 4 * r
                    bcol = 16 * tj + (np.arange(64) & 15)
                    b = np.array([perm[x] for x in bcol])
                    ok = (a < 27) & (b >= 0)
                    off = np.full(64, -1, np.int64)
                    off[ok] = block_addr(nd[a[ok]], nd[b[ok]])
                    for rr in range(9):
                        t = np.arange(64) + 64 * rr
                        pl, e = t // 9, t % 9
                        o = off[pl]
                        v = o >= 0
                        if not v.any():
                            continue
                        addr = o[v] + 8 * e[v]
                        total_seg += len(np.unique(addr // 64))
                        total_inst += 1
                        total_atom += int(v.sum())
    return total_seg / len(cells), total_inst / len(cells), total_atom / len(cells)


def row_floor(cells):
    """segments per cell if every (cell, row) were scattered by one ideal instruction: what the numbering alone allows"""
    tot = 0
    for c in cells:
        nd = cu[c]
        for a in range(27):
            addr = block_addr(np.full(27, nd[a]), nd)[:, None] + 8 * np.arange(9)[None, :]
            tot += len(np.unique(addr // 64))
    return tot / len(cells)


rng = np.random.default_rng(1)
cells = rng.choice(nc, size=min(nc, 400), replace=False)
print(f"{n}^3 cells, {len(cells)} sampled; layout floor 729 * 72 / 64 = {729 * 72 / 64:.0f} segments per cell")
print(f"floor of this node numbering (one ideal instruction per cell row): {row_floor(cells):.1f} segments per cell")
for o in orders:
    seg, inst, atom = simulate(o, cells)
    print(f"column order {o:7s}: {seg:7.1f} segments per cell ({seg / 729:.3f} per block), {inst:.1f} atomic instructions, {atom:.0f} atomics")
