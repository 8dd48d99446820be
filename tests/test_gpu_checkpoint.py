"""FluidSolver::save_checkpoint / load_checkpoint of the host mirror (reference protocol: source/mpi_fluid_solver.cpp:582-713,
call sites mpi_insim.cpp:477-480,499-507): a run interrupted after a checkpoint and restarted from it must end in the same
state as the uninterrupted run, and a checkpoint written by a partitioned run restores on a different number of ranks."""
import ctypes as C
import glob
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _prm(dim, end_time, save):
    from openifem_amd import host
    return host.channel_prm(dim, end_time=end_time).replace("Save interval = 100", f"Save interval = {save}")


@pytest.mark.parametrize("kind", ["InsIM", "InsIMEX"])
def test_restart_from_checkpoint_reproduces_the_uninterrupted_run(tmp_path, kind):
    from openifem_amd import host
    cls = getattr(host, kind)
    d = str(tmp_path)
    a = cls(_prm(2, 4e-3, 2e-3), (8, 4), (0, 0), (2.0, 0.2))
    a.set_output_dir(d)
    a.run()  # steps 1..4, checkpoints after steps 2 and 4
    assert a.time()[0] == 4
    va, pa = a.get_current_solution()
    names = sorted(os.path.basename(f) for f in glob.glob(d + "/*.fluid_checkpoint"))
    assert names == ["000002.fluid_checkpoint", "000004.fluid_checkpoint"]
    info = open(d + "/000002.fluid_checkpoint.info").read().split("\n")[1].split()
    assert info[:4] == ["1", "2", "2", "1"] and info[7] == "2"
    a.close()
    for f in glob.glob(d + "/000004.*"):  # "the job died after step 2"
        os.remove(f)
    b = cls(_prm(2, 4e-3, 2e-3), (8, 4), (0, 0), (2.0, 0.2))
    b.set_output_dir(d)
    b.run()  # loads step 2, runs steps 3 and 4
    n, t = b.time()
    assert n == 4 and abs(t - 4e-3) < 1e-15
    vb, pb = b.get_current_solution()
    # Newton / Krylov tolerances are relative, so the restarted steps repeat the same arithmetic on the restored state
    assert np.abs(vb - va).max() <= 1e-9 * np.abs(va).max()
    assert np.abs(pb - pa).max() <= 1e-9 * np.abs(pa).max()
    # the older checkpoint is the only one kept besides the new one
    names = sorted(os.path.basename(f) for f in glob.glob(d + "/*.fluid_checkpoint"))
    assert names == ["000002.fluid_checkpoint", "000004.fluid_checkpoint"]
    b.close()


def test_checkpoint_of_two_ranks_restores_on_one(tmp_path):
    from openifem_amd import host, capi
    L = capi.load()
    d = str(tmp_path)
    reps, P = (4, 4, 4), (2, 1, 1)
    w = C.c_void_p(L.ifem_local_world_create(2))
    out, errs = [None, None], []

    def work(rank):
        try:
            s = host.InsIM(host.channel_prm(3), reps, (0, 0, 0), (2.0, 0.2, 0.2))
            s.set_partition(P, rank, local_world=w)
            s.setup(0)
            s.run_one_step(True)
            s.save_checkpoint(d, 1)
            t = s.partition_tables()
            v, p = s.get_current_solution()
            out[rank] = (t, v, p)
            s.close()
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    L.ifem_local_world_destroy(w)
    assert not errs, errs
    assert sorted(os.path.basename(f) for f in glob.glob(d + "/000001.*")) == [
        "000001.fluid_checkpoint", "000001.fluid_checkpoint.info", "000001.fluid_checkpoint_fixed.data.0",
        "000001.fluid_checkpoint_fixed.data.1"]
    n_ug, n_pg = out[0][0]["n_unodes_global"], out[0][0]["n_pnodes_global"]
    vg, pg = np.full((n_ug, 3), np.nan), np.full(n_pg, np.nan)
    for t, v, p in out:
        nuo, npo = t["n_unodes_owned"], t["n_pnodes_owned"]
        vg[t["l2g_u"][:nuo]] = np.asarray(v).reshape(-1, 3)[:nuo]
        pg[t["l2g_p"][:npo]] = np.asarray(p)[:npo]
    assert not np.isnan(vg).any() and not np.isnan(pg).any()
    s = host.InsIM(host.channel_prm(3), reps, (0, 0, 0), (2.0, 0.2, 0.2))
    assert s.load_checkpoint(d)
    assert s.time()[0] == 1
    t = s.partition_tables()
    v, p = s.get_current_solution()
    assert np.array_equal(np.asarray(v).reshape(-1, 3), vg[t["l2g_u"]])
    assert np.array_equal(np.asarray(p), pg[t["l2g_p"]])
    s.close()
