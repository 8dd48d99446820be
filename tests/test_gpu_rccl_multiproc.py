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


def test_two_process_rccl_bench_run():
    from openifem_amd import capi
    if capi.load().ifem_device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "16", "--steps", "1",
           "--warmup", "1", "--cpu-cells", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["fgmres_iters"] > 0
    assert out["config"]["n_dofs"] == 3 * (2 * 32 + 1) * 33 * 33 + 33 * 17 * 17
