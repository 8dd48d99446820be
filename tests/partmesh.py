"""Test-side partitioner of a general (unstructured / hanging-node) mesh into per-rank local tables + halo plans in the
layout include/ifem_hip.h asks for (ifem_mesh_desc, ifem_partition, local hanging lines, local Dirichlet lines).

Stand-in for what p4est + DoFTools::extract_locally_relevant_dofs give the reference (mpi_fluid_solver.cpp:140-152): a row
belongs to the lowest rank touching its node, every rank assembles the cells touching its owned rows, ghosts are the other
nodes of those cells plus the masters of its local hanging nodes, grouped by owner rank in global order.  Deliberately
independent of the product's own partitioners (csrc/host/grid.cpp) so that the two can be cross-checked."""
import threading

import numpy as np


class LocalPart:
    pass


def _owners(n_nodes, cell_nodes, cell_rank, nranks):
    own = np.full(n_nodes, nranks, np.int64)
    np.minimum.at(own, cell_nodes.ravel(), np.repeat(cell_rank, cell_nodes.shape[1]))
    assert own.max() < nranks, "a node belongs to no cell"
    return own


def partition_mesh(m, cell_rank, nranks):
    """m: BoxMesh / HangingMesh-like global mesh; cell_rank[c] in [0, nranks).  Returns [LocalPart] * nranks."""
    dim = m.dim
    cell_rank = np.asarray(cell_rank, np.int64)
    own_u = _owners(m.n_unodes, m.cell_unodes, cell_rank, nranks)
    own_p = _owners(m.n_pnodes, m.cell_pnodes, cell_rank, nranks)
    n_u_glob = dim * m.n_unodes
    hang_dof = getattr(m, "hang_dof", np.zeros(0, np.int32))
    lines = {}
    for i, d in enumerate(hang_dof):
        lines[int(d)] = (m.hang_master[m.hang_ptr[i]:m.hang_ptr[i + 1]], m.hang_weight[m.hang_ptr[i]:m.hang_ptr[i + 1]])
    hang_unodes = {int(d) // dim: None for d in hang_dof if d < n_u_glob}
    hang_pnodes = {int(d) - n_u_glob: None for d in hang_dof if d >= n_u_glob}
    parts = []
    for r in range(nranks):
        P = LocalPart()
        touch = (own_u[m.cell_unodes] == r).any(axis=1) | (own_p[m.cell_pnodes] == r).any(axis=1)
        P.cells = np.nonzero(touch)[0]
        for kind, own, cell_nodes, hang_nodes in (("u", own_u, m.cell_unodes, hang_unodes), ("p", own_p, m.cell_pnodes, hang_pnodes)):
            owned = np.nonzero(own == r)[0]
            local = set(int(x) for x in np.unique(cell_nodes[P.cells]))
            local |= set(int(x) for x in owned)
            # masters of the local hanging nodes join the ghost layer
            extra = set()
            for nd in local:
                if nd in hang_nodes:
                    for c in range(dim if kind == "u" else 1):
                        dof = dim * nd + c if kind == "u" else n_u_glob + nd
                        for mdof in lines[dof][0]:
                            extra.add(int(mdof) // dim if kind == "u" else int(mdof) - n_u_glob)
            local |= extra
            ghosts = np.array(sorted(local - set(int(x) for x in owned), key=lambda g: (own[g], g)), np.int64)
            setattr(P, "owned_" + kind, owned)
            setattr(P, "ghost_" + kind, ghosts)
            l2g = np.concatenate([owned, ghosts]).astype(np.int64)
            setattr(P, "l2g_" + kind, l2g)
            g2l = -np.ones(len(own), np.int64)
            g2l[l2g] = np.arange(len(l2g))
            setattr(P, "g2l_" + kind, g2l)
        parts.append(P)
    # neighbours (symmetric, one list for both node kinds) and the halo plans
    nbr = [set() for _ in range(nranks)]
    for r, P in enumerate(parts):
        for kind, own in (("u", own_u), ("p", own_p)):
            for q in np.unique(own[getattr(P, "ghost_" + kind)]):
                nbr[r].add(int(q))
                nbr[int(q)].add(r)
    for r, P in enumerate(parts):
        P.rank, P.nranks = r, nranks
        P.neighbors = np.array(sorted(nbr[r]), np.int32)
    for r, P in enumerate(parts):
        for kind, own in (("u", own_u), ("p", own_p)):
            ghosts = getattr(P, "ghost_" + kind)
            rptr = [0]
            for q in P.neighbors:
                rptr.append(rptr[-1] + int((own[ghosts] == q).sum()))
            assert rptr[-1] == len(ghosts)
            setattr(P, "recv_%s_ptr" % kind, np.array(rptr, np.int32))
            sptr, sidx = [0], []
            for q in P.neighbors:
                qg = getattr(parts[q], "ghost_" + kind)
                mine = qg[own[qg] == r]  # in q's ghost order
                sidx += [int(x) for x in getattr(P, "g2l_" + kind)[mine]]
                sptr.append(len(sidx))
            setattr(P, "send_%s_ptr" % kind, np.array(sptr, np.int32))
            setattr(P, "send_%s_idx" % kind, np.array(sidx, np.int32))
    # local tables
    for r, P in enumerate(parts):
        P.dim, P.kv = dim, m.kv
        P.cell_unodes = P.g2l_u[m.cell_unodes[P.cells]].astype(np.int32)
        P.cell_pnodes = P.g2l_p[m.cell_pnodes[P.cells]].astype(np.int32)
        assert P.cell_unodes.min() >= 0 and P.cell_pnodes.min() >= 0
        P.vcoords = np.ascontiguousarray(m.vcoords[P.cells])
        P.cell_face_bid = np.ascontiguousarray(m.cell_face_bid[P.cells])
        P.n_unodes_owned, P.n_unodes = len(P.owned_u), len(P.l2g_u)
        P.n_pnodes_owned, P.n_pnodes = len(P.owned_p), len(P.l2g_p)
        P.n_u_ext = dim * P.n_unodes
        P.n_local = P.n_u_ext + P.n_pnodes
        P.n_owned = dim * P.n_unodes_owned + P.n_pnodes_owned
        # global dof ids of the local (extended) vector and of the compact owned vector
        P.ext_gdof = np.concatenate([(dim * P.l2g_u[:, None] + np.arange(dim)[None, :]).ravel(), n_u_glob + P.l2g_p])
        P.own_gdof = np.concatenate([(dim * P.owned_u[:, None] + np.arange(dim)[None, :]).ravel(), n_u_glob + P.owned_p])
        g2l_dof = -np.ones(n_u_glob + m.n_pnodes, np.int64)
        g2l_dof[P.ext_gdof] = np.arange(P.n_local)
        P.g2l_dof = g2l_dof
        # hanging lines of every local hanging dof (owned or ghost), local ids
        hd, hp, hm, hw = [], [0], [], []
        for d in hang_dof:
            if g2l_dof[d] < 0:
                continue
            ms, ws = lines[int(d)]
            lm = g2l_dof[ms]
            assert lm.min() >= 0, "master of a local hanging dof is not local"
            hd.append(int(g2l_dof[d]))
            hm += [int(x) for x in lm]
            hw += [float(x) for x in ws]
            hp.append(len(hm))
        P.hang_dof, P.hang_ptr = np.array(hd, np.int32), np.array(hp, np.int32)
        P.hang_master, P.hang_weight = np.array(hm, np.int32), np.array(hw, float)
    return parts


def local_dirichlet(P, dofs, vals):
    """global Dirichlet lines -> the lines of the dofs that are local on P (owned and ghost), local ids"""
    l = P.g2l_dof[np.asarray(dofs, np.int64)]
    keep = l >= 0
    return l[keep].astype(np.int32), np.asarray(vals, float)[keep]


def make_context(capi, P, world, device=0):
    part, keep = capi.make_partition(P.rank, P.nranks, P.neighbors, P.send_u_ptr, P.send_u_idx, P.recv_u_ptr,
                                     P.send_p_ptr, P.send_p_idx, P.recv_p_ptr, local_world=world)
    ctx = capi.Context(P.dim, P.kv, P.vcoords, P.cell_unodes, P.cell_pnodes, P.cell_face_bid, P.n_unodes, P.n_pnodes,
                       n_unodes_owned=P.n_unodes_owned, n_pnodes_owned=P.n_pnodes_owned, partition=part, device=device)
    ctx._part_keep = (part, keep)
    return ctx


def run_virtual_ranks(capi, parts, work, timeout=600):
    """work(rank, part, ctx) -> result, on one host thread per virtual rank sharing a local world (one GPU)."""
    L = capi.load()
    import ctypes as C
    nranks = len(parts)
    if nranks == 1:
        ctx = capi.Context(parts[0].dim, parts[0].kv, parts[0].vcoords, parts[0].cell_unodes, parts[0].cell_pnodes,
                           parts[0].cell_face_bid, parts[0].n_unodes, parts[0].n_pnodes)
        try:
            return [work(0, parts[0], ctx)]
        finally:
            ctx.close()
    w = C.c_void_p(L.ifem_local_world_create(nranks))
    out, errs = [None] * nranks, []

    def run(rank):
        try:
            ctx = make_context(capi, parts[rank], w)
            try:
                out[rank] = work(rank, parts[rank], ctx)
            finally:
                ctx.close()
        except Exception:  # noqa
            import traceback
            errs.append((rank, traceback.format_exc()))

    th = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=timeout)
    assert not errs, errs
    assert all(not t.is_alive() for t in th), "a virtual rank hangs"
    L.ifem_local_world_destroy(w)
    return out


def gather_owned(parts, results, n_global):
    """owned (compact) result vectors of all ranks -> one global vector"""
    x = np.full(n_global, np.nan)
    for P, r in zip(parts, results):
        x[P.own_gdof] = r
    assert not np.isnan(x).any(), "some dof is owned by no rank"
    return x
