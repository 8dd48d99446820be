"""`python bench.py --gpus N` started plainly (the way the driver starts `--gpus 1`) launches its own N ranks: on a box without a
GPU both ranks reach the gloo rendezvous and stop at IFEM_E_NODEVICE -- not at the launcher.  The reference runs every MPI test
on two ranks (/root/reference/tests/CMakeLists.txt:52,71: `mpirun -n 2`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    sys.path.insert(0, ROOT)
    from openifem_amd import capi
    return capi.load().ifem_device_count() == 0


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(not _no_gpu(), reason="a GPU box runs the real thing (tests/test_gpu_rccl_multiproc.py)")
@pytest.mark.parametrize("n", [2, 4])
def test_plain_launch_reaches_the_rendezvous_and_stops_at_nodevice(n):
    from openifem_amd import capi
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode == capi.E_NODEVICE_EXIT, r.stderr[-2000:]
    for rank in range(n):
        assert f"[bench rank {rank}] gloo rendezvous of {n} ranks complete" in r.stderr, r.stderr[-2000:]
        assert f"[bench rank {rank}] IFEM_E_NODEVICE" in r.stderr, r.stderr[-2000:]
    assert "must be launched" not in r.stderr and '"metric"' not in r.stdout  # no JSON line without a device


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without a GPU")
def test_launcher_environment_is_honoured():
    """started by torch.distributed.run (RANK / WORLD_SIZE set) bench.py does not launch again; a mismatch is an error"""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without a GPU")
def test_single_rank_without_device_fails_loudly():
    from openifem_amd import capi
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"])
    assert r.returncode == capi.E_NODEVICE_EXIT and "IFEM_E_NODEVICE" in r.stderr and '"metric"' not in r.stdout
