"""world_size-2 (and 4) CPU tests of the multi-GPU host logic over torch.distributed/gloo (no GPU):
the block partition, ownership, ghost numbering and halo plans the RCCL path consumes."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P, reps, dim, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        from openifem_amd import host
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        p1 = (2.0, 0.2, 0.2)[:dim]
        cylinder = reps == "cylinder"
        if cylinder:  # unstructured mesh cut into strips (partition_unstructured)
            prm = open(os.path.join(ROOT, "tests", "golden", "prm", "fluid_cylinder_mpi.prm")).read()
            s = host.InsIM(prm, mesh="cylinder")
            s.set_partition(P, rank)
            s.setup_host_only(1)
        else:
            s = host.InsIM(host.channel_prm(dim), reps, (0,) * dim, p1)
            s.set_partition(P, rank)
            s.setup_host_only(0)
        t = s.partition_tables()
        nUo, nUl, nPo, nPl = t["n_unodes_owned"], t["n_unodes_local"], t["n_pnodes_owned"], t["n_pnodes_local"]
        if tuple(P) == (2, 2, 2):
            # the north-star octant split: every rank talks to 3 face + 3 edge + 1 corner neighbours
            assert sorted(int(v) for v in t["neighbors"]) == [r for r in range(8) if r != rank], t["neighbors"]
        # (a) every global node is owned by exactly one rank
        for key, no, ng in (("l2g_u", nUo, t["n_unodes_global"]), ("l2g_p", nPo, t["n_pnodes_global"])):
            cnt = torch.zeros(ng, dtype=torch.int64)
            cnt[torch.from_numpy(t[key][:no])] += 1
            dist.all_reduce(cnt)
            assert int(cnt.min()) == 1 and int(cnt.max()) == 1, f"{key}: ownership is not a partition"
        # (b) halo exchange through the send/recv plans reproduces f(global id) on the ghosts
        for which, l2g, no, nl, bs in (("u", t["l2g_u"], nUo, nUl, dim), ("p", t["l2g_p"], nPo, nPl, 1)):
            f = lambda g: np.sin(0.37 * g[:, None] + np.arange(bs)[None, :])
            x = np.zeros((nl, bs))
            x[:no] = f(l2g[:no])
            sp, si, rp = t[f"send_{which}_ptr"], t[f"send_{which}_idx"], t[f"recv_{which}_ptr"]
            reqs, bufs = [], []
            for k, nb in enumerate(t["neighbors"]):
                sbuf = torch.from_numpy(np.ascontiguousarray(x[si[sp[k]:sp[k + 1]]]))
                rbuf = torch.zeros((rp[k + 1] - rp[k], bs), dtype=torch.float64)
                bufs.append((k, rbuf))
                if sbuf.numel():
                    reqs.append(dist.isend(sbuf, int(nb)))
                if rbuf.numel():
                    reqs.append(dist.irecv(rbuf, int(nb)))
            for r in reqs:
                r.wait()
            for k, rbuf in bufs:
                x[no + rp[k]:no + rp[k + 1]] = rbuf.numpy()
            assert np.abs(x - f(l2g)).max() == 0.0, f"halo {which}: ghost values differ from the owners'"
        # (b2) the 2-deep pressure halo plan of the distributed explicit S_m: after the exchange every node of the lattice box
        #      "owned range +-2" carries f(global id) at its S_m column id
        sm = s.sm_plan()
        if sm is not None:
            lo, bn, N = sm["box_lo"], sm["box_n"], sm["lattice_n"]
            zz, yy, xx = np.meshgrid(np.arange(bn[2]) + lo[2], np.arange(bn[1]) + lo[1], np.arange(bn[0]) + lo[0], indexing="ij")
            gid = ((zz * N[1] + yy) * N[0] + xx).ravel()
            f1 = lambda g: np.sin(0.37 * g)
            xs = np.zeros(nPo + sm["n_far"])
            xs[:nPo] = f1(t["l2g_p"][:nPo])
            sp, si, rp = sm["send_s_ptr"], sm["send_s_idx"], sm["recv_s_ptr"]
            reqs, bufs = [], []
            for k, nb in enumerate(t["neighbors"]):
                sbuf = torch.from_numpy(np.ascontiguousarray(xs[si[sp[k]:sp[k + 1]]]))
                rbuf = torch.zeros(int(rp[k + 1] - rp[k]), dtype=torch.float64)
                bufs.append((k, rbuf))
                if sbuf.numel():
                    reqs.append(dist.isend(sbuf, int(nb)))
                if rbuf.numel():
                    reqs.append(dist.irecv(rbuf, int(nb)))
            for r in reqs:
                r.wait()
            for k, rbuf in bufs:
                xs[nPo + rp[k]:nPo + rp[k + 1]] = rbuf.numpy()
            assert sm["box_id"].min() >= 0 and len(np.unique(sm["box_id"])) == len(sm["box_id"]) == len(xs)
            assert np.abs(xs[sm["box_id"]] - f1(gid)).max() == 0.0, "2-deep pressure halo: values differ from the owners'"
        if cylinder:
            # (c') every global cell appears on at least one rank, local tables reproduce the global geometry
            from cylmesh import CylinderMesh
            m = CylinderMesh(1)
            cu, cp, fb, vc = s.cell_tables()
            uc, pc = s.node_coords()
            # the vertex nodes of every local cell sit on the cell's vertices (node numbering itself is an implementation detail)
            assert np.abs(pc[cp] - vc).max() == 0.0 and np.abs(uc[cu][:, [0, 2, 6, 8]] - vc).max() == 0.0
            key = {tuple(np.round(r, 12)): i for i, r in enumerate(m.vcoords.reshape(m.n_cells, -1))}
            mine = torch.zeros(m.n_cells, dtype=torch.int64)
            for r in vc.reshape(len(vc), -1):
                mine[key[tuple(np.round(r, 12))]] = 1
            dist.all_reduce(mine)
            assert int(mine.min()) >= 1
            dist.barrier()
            dist.destroy_process_group()
            q.put((rank, "ok"))
            return
        # (c) local connectivity maps to the global lattice connectivity; every global cell touching an owned node is local
        from boxmesh import BoxMesh
        m = BoxMesh(reps, (0,) * dim, p1, kv=2)
        cu, cp, fb, vc = s.cell_tables()
        gu = t["l2g_u"][cu]
        key = {tuple(row): i for i, row in enumerate(m.cell_unodes.tolist())}
        gcells = np.array([key[tuple(r)] for r in gu.tolist()])
        assert np.array_equal(t["l2g_p"][cp], m.cell_pnodes[gcells])
        assert np.array_equal(fb, m.cell_face_bid[gcells]) and np.abs(vc - m.vcoords[gcells]).max() < 1e-15
        owned = set(t["l2g_u"][:nUo].tolist())
        need = {c for c in range(m.n_cells) if owned & set(m.cell_unodes[c].tolist())}
        assert need <= set(gcells.tolist()), "a cell touching an owned row is not local"
        # (d) constraints cover exactly the constrained global dofs among the local ones
        d, v = s.constraints()
        bcs = {2: (3, [0, 0]), 3: (3, [0, 0])} if dim == 2 else {2: (7, [0, 0, 0]), 3: (7, [0, 0, 0]), 4: (4, [0]), 5: (4, [0])}
        d0, _ = m.dirichlet(bcs)
        gl = set((t["l2g_u"][d // dim] * dim + d % dim).tolist())
        local_all = set((t["l2g_u"][:, None] * dim + np.arange(dim)[None, :]).ravel().tolist())
        assert gl == (set(d0.tolist()) & local_all)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,P,reps,dim", [(2, (2, 1, 1), (4, 2, 2), 3), (4, (2, 2, 1), (4, 4, 2), 3), (2, (2, 1), (6, 3), 2),
                                               (4, (2, 2, 1), (8, 6, 3), 3), (2, (2, 1, 1), "cylinder", 2), (3, (3, 1, 1), "cylinder", 2),
                                               (8, (2, 2, 2), (4, 4, 4), 3)])
def test_block_partition_over_gloo(world, P, reps, dim):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, reps, dim, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
