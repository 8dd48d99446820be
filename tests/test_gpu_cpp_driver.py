"""A compiled C++ caller of the host mirror (tests/cpp/channel3d.cpp, shaped like the reference's test drivers,
tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:19-105): InsIM<3>::run() from C++ reaches the multigrid-preconditioned inner
solver without any Python, reproduces the Poiseuille answer, and its bench mode sees the iteration counts of the ctypes
path bench.py uses (the same C++ objects underneath)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    subprocess.check_call(["make", "-C", CPP, "channel3d"])
    from openifem_amd import host
    d = tmp_path_factory.mktemp("cpp")
    prm = d / "parameters.prm"
    prm.write_text(host.channel_prm(3, dt=2e-3, end_time=4e-2))
    return os.path.join(CPP, "channel3d"), str(prm), str(d)


def _run(exe, prm, cwd, *args):
    out = subprocess.run([exe, prm] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_cpp_run_reaches_the_multigrid_path_and_the_poiseuille_answer(driver):
    exe, prm, cwd = driver
    r = _run(exe, prm, cwd, "run", 16)
    assert r["multigrid_levels"] >= 2 and r["ainv_kind"] == 4
    assert r["rel_error"] < 1e-3, r


def test_cpp_bench_mode_matches_the_ctypes_path(driver):
    exe, prm, cwd = driver
    n = int(os.environ.get("IFEM_TEST_CPP_N", "32"))
    r = _run(exe, prm, cwd, "bench", n, 2, 1)
    from openifem_amd import multigpu
    S, _, _ = multigpu.make_channel_solver(n, 0, 1, 0, None)
    # bench.py sets nothing on top of the mirror's defaults: the C++ driver with no knobs and the ctypes path
    # bench.py uses run the same solver options
    assert (S.opts.ainv_kind, S.opts.inner_restart, S.opts.inner_rel, S.opts.inner_rel_first) == (4, 16, 1e-2, 5e-5)
    assert (r["ainv_kind"], r["inner_restart"], r["inner_rel"], r["inner_rel_first"]) == (4, 16, 1e-2, 5e-5)
    S.channel_state()
    for _ in range(3):
        S.assemble(False)
        st = S.solve(False)
    res, b = S.true_residual()
    assert r["multigrid_levels"] == len(S.mg_levels()) and r["ainv_kind"] == S.opts.ainv_kind and r["inner_restart"] == 16
    assert (r["fgmres_iters"], r["inner_iters"], r["cg_mp_iters"], r["cg_sm_iters"]) == (st.fgmres_iters, st.inner_iters, st.cg_mp_iters, st.cg_sm_iters)
    # (the converged residual is the tail of a Krylov process fed by an atomically summed matrix: equal to a few per cent)
    assert abs(r["true_rel_residual"] - res / b) <= 0.25 * res / b
    assert r["true_rel_residual"] <= 1.05e-4
    S.close()
