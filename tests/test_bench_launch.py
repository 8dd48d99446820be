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


def _stub_record(fat=1):
    """a bench record shaped like the real one (profiles/r05_bench128_default.jsonl), padded with `fat` copies of the big tables"""
    fams = ["assemble_cells", "mf_cell", "mf_gather", "spmv_uu", "zero_fill", "spmv_sm", "vector_ops", "mg_transfer", "spmv_b_bt",
            "spmv_mp", "maxpy", "smoother_setup", "cg_recurrence", "mdot", "other"]
    kernels = [{"family": f, "kernel": f + " " + "x" * 80, "ms_per_step": 1.2345678901 * (i + 1), "algorithmic_bytes": 1.23456789e10,
                "traffic_source": "profiles/pmc_traffic.json " * 3} for i, f in enumerate(fams)] * fat
    return {"metric": "DoF/s per Newton step (assemble+solve), 3D INS Q2/Q1", "value": 266518123.456789, "unit": "DoF/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 199.12345678, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D channel flow 128x128x128 Q2/Q1, mpi_insim Newton step " + "y" * 300, "n_dofs": 53070468,
                       "assemble_ms": 100.4, "solve_ms": 98.6, "fgmres_iters": 1, "true_rel_residual": 7.77e-5, "rccl_nranks": 0,
                       "halo_exchanges_per_step": 0.0, "coarse_levels": [[128, 64, 64]] * 40 * fat, "solver_opts": {"inner_rel": 1e-2}},
            "value_cold": 1.97e8, "value_sustained": 1.52e8,
            "time_step": {"ms": 1043.0, "newton_iterations": 3, "fgmres_iters": [1, 6, 3], "kernel_ms_per_loop": {f: 1.0 for f in fams}},
            "roofline": {"bound": "mfma", "achieved": 25.35, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.3226, "traffic": 122626662400,
                         "kernel": "k_ins_assemble3 " + "z" * 300, "launch_ms": 86.28, "algorithmic_bytes": 8.3e10,
                         "algorithmic_flops": 2187746869248, "hbm_frac": 0.12, "atomic_segment_frac": 0.88, "kernels": kernels,
                         "kernel_ms_per_step": {f: 1.23456789 for f in fams * fat}},
            "cylinder_workloads": {w: {"value": 1.7e6, "unit": "DoF/s", "ms_per_step": 31.2, "config": {"note": "n" * 2000}}
                                   for w in ("cylinder2d", "cylinder2d_scnsim", "cylinder3d")},
            "fsi_inputs": {"note": "f" * 3000 * fat},
            "cpu_baseline": {"value": 138072.0, "unit": "DoF/s", "cores": 12, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                             "sample": "1 Newton step " + "s" * 900, "scaling_table": {"rows": [{"threads": t} for t in range(64 * fat)]}}}


@pytest.mark.parametrize("fat", [1, 8])
def test_stdout_line_is_compact_and_parses(fat):
    """the driver's capture keeps ~8 kB of stdout; the ONE line stays under 4 kB whatever the side legs hold
    and carries the contract's keys, `roofline` and `cpu_baseline`"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    rec = _stub_record(fat)
    line = bench.compact_line(rec, "bench_detail.json")
    assert "\n" not in line and len(line) < 4096, len(line)
    o = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in o, k
    assert o["value"] == pytest.approx(rec["value"], rel=1e-5) and o["config"]["n_dofs"] == rec["config"]["n_dofs"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(o["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(o["cpu_baseline"]) or fat > 1
    assert {"value", "unit", "cores", "kind"} <= set(o["cpu_baseline"])
    assert o["value_cold"] == rec["value_cold"] and o["value_sustained"] == rec["value_sustained"]
    assert "kernels" not in o["roofline"] and "cylinder_workloads" not in o and "fsi_inputs" not in o
