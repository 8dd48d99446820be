"""The process environment of `bench.py --gpus N`: one process per GPU launched by torch.distributed.run, torch imported
(for the gloo rendezvous) BEFORE libifem_hip.so, so the library binds to the HIP runtime and the RCCL that torch bundles.

 * on any GPU box: in a child process with that import order, the RCCL round trip of comm.hip (communicator, all-reduce,
   grouped send/recv) and one assemble + solve checked against the oracle (__graft_entry__.smoke);
 * on a box with >= 2 GPUs: a real 2-rank run of bench.py over RCCL (skipped on the 1-GPU test boxes) -- the first
   consumer of comm.hip's ncclSend/ncclRecv between two devices."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_library_runs_on_the_runtime_torch_bundles():
    code = ("import torch, torch.distributed as dist\n"
            "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from openifem_amd import capi\n"
            "L = capi.load()\n"
            "assert L.ifem_comm_selftest(0) == 0, L.ifem_last_error().decode()\n"
            "import __graft_entry__ as g\n"
            "g.smoke()\n"
            "print('CHILD_OK')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_two_process_rccl_bench_run(tmp_path):
    """bench.py on two GPUs over RCCL against the same global mesh in ONE context: RCCL saw two ranks, the Newton update of
    the two ranks equals the single-context update to 1e-6 (tight tolerances, as the virtual-rank tests), same outer
    iteration count, the true residual meets the tolerance -- a wrong halo fails all three"""
    import numpy as np
    from openifem_amd import capi
    if capi.load().ifem_device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    dump = str(tmp_path / "upd")
    common = ["--cells", "16", "--steps", "1", "--warmup", "1", "--cpu-cells", "0", "--fgmres-rel", "1e-9", "--inner-rel", "1e-4",
              "--inner-rel-first", "0", "--extras", "0", "--tuned", "0", "--fsi", "0"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump-update", dump] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    cfg = out["config"]
    assert out["n_gpus"] == 2 and out["value"] > 0 and cfg["fgmres_iters"] > 0
    assert cfg["n_dofs"] == 3 * (2 * 32 + 1) * 33 * 33 + 33 * 17 * 17
    assert cfg["rccl_nranks"] == 2 and cfg["comm_transport"] == "rccl" and cfg["halo_neighbors"] == 1
    assert cfg["halo_exchanges_per_step"] > 0 and cfg["true_rel_residual"] <= 1.05e-9
    # the same 32 x 16 x 16 channel in one context, same tolerances, same seeded state (keyed by global dof)
    from openifem_amd import host
    S = host.InsIM(host.channel_prm(3), (32, 16, 16), (0, 0, 0), (2.0, 0.2, 0.2))
    S.setup(0)
    S.opts.inner_rel, S.opts.inner_rel_first, S.opts.fgmres_rel = 1e-4, 0.0, 1e-9
    S.channel_state()
    for _ in range(2):
        S.assemble(False)
        st = S.solve(False)
    _, n_u, n_p = S.sizes()
    x1 = np.zeros(n_u + n_p)
    assert S.L.ifem_vec_get(S.ctx, capi.VEC_UPDATE, x1.ctypes.data_as(__import__("ctypes").c_void_p)) == 0
    t1 = S.partition_tables()
    ref = np.full(n_u + n_p, np.nan)
    ref[(t1["l2g_u"][:, None] * 3 + np.arange(3)[None, :]).ravel()] = x1[:n_u]
    ref[3 * t1["n_unodes_global"] + t1["l2g_p"]] = x1[n_u:]
    got = np.full(n_u + n_p, np.nan)
    for rk in range(2):
        z = np.load(f"{dump}.rank{rk}.npz")
        nuo = len(z["l2g_u"])
        got[(z["l2g_u"][:, None] * 3 + np.arange(3)[None, :]).ravel()] = z["update"][:3 * nuo]
        got[3 * int(z["n_unodes_global"]) + z["l2g_p"]] = z["update"][3 * nuo:]
    assert not np.isnan(got).any() and not np.isnan(ref).any()
    assert np.linalg.norm(got - ref) <= 1e-6 * np.linalg.norm(ref)
    assert abs(cfg["fgmres_iters"] - st.fgmres_iters) <= 1
    S.close()
