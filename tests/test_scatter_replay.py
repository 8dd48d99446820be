"""The segment count of the assembly kernel's staged A_uu scatter, replayed on the CPU (tools/scatter_sim.py: the C++ host mirror's
DoF tables, the block-interleaved layout, the lane -> (pair, entry) mapping of assemble3.hip).  DESIGN section 4 rests on these
counts: the memory-side atomic path retires a fixed number of 64-byte segments per second, so the order of the tile columns,
the storage order of the blocks inside a row and the alignment of the staged rows each have to lower the count they are there
for.  (The kernel itself is checked entrywise against the oracle in tests/test_gpu_parity.py; this pins the reasoning.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_each_step_of_the_scatter_reorganisation_lowers_the_segment_count():
    import scatter_sim
    R = scatter_sim.Replay(5)
    cells = np.random.default_rng(3).choice(R.nc, size=60, replace=False)
    col = R.build_rows("col")
    fl_col, lex, inst = R.replay(*col, cells, "lex", "lane")
    _, by_id, _ = R.replay(*col, cells, "id", "lane")
    _, by_id_rank, _ = R.replay(*col, cells, "id", "rank")
    assert inst == 122  # 4 tile pairs x 4 steps x 9 rounds, minus the rounds that hold padding only
    assert by_id_rank == by_id  # rows in column order: the node-id order of the tile columns IS the order of the positions
    cel = R.build_rows("cells")
    fl_cel, ranked, _ = R.replay(*cel, cells, "id", "rank")
    _, shipped, inst_a = R.replay(*cel, cells, "id", "aligned")
    # layout floor: 729 blocks of 72 bytes = 820 segments; every variant lies above its row order's floor
    assert 820 <= fl_cel < fl_col <= by_id and fl_cel <= shipped
    assert shipped < ranked < by_id < lex
    assert lex > 1150 and by_id < 0.9 * lex and shipped < 0.81 * lex, (lex, by_id, ranked, shipped)
    assert inst_a <= 130  # the alignment shift costs at most one more round per step
    # round 4: a row's 27 blocks leave together, staged as the image of their memory
    cel2 = R.build_rows("cells2")
    fl2, full, inst_f = R.replay(*cel2, cells, "id", "full")
    _, full_u, _ = R.replay(*cel2, cells, "id", "full_unaligned")
    assert inst_f == 108  # 27 rows x 4 instructions
    assert 820 <= fl2 <= fl_cel and fl2 <= full < full_u
    assert full < 0.95 * shipped and full < 1.03 * fl2, (full, shipped, fl2)
