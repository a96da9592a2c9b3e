"""world_size-2 (and 4) runs on CPU over gloo: the slab decomposition and the
communication schedule the HIP library uses (operator outputs exchanged once,
vector updates carried on the ghost planes, one all-reduce per dot product)
reproduce the undivided solve."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for q in (ROOT, HERE):
    if q not in sys.path:
        sys.path.insert(0, q)


class GlooComm:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def allreduce(self, v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0])

    def exchange(self, lo, send_lo, hi, send_hi):
        reqs, got_lo, got_hi = [], None, None
        if lo is not None:
            got_lo = torch.empty(send_lo.shape, dtype=torch.float64)
            reqs += [dist.isend(torch.from_numpy(send_lo), lo), dist.irecv(got_lo, lo)]
        if hi is not None:
            got_hi = torch.empty(send_hi.shape, dtype=torch.float64)
            reqs += [dist.isend(torch.from_numpy(send_hi), hi), dist.irecv(got_hi, hi)]
        for r in reqs:
            r.wait()
        return (None if got_lo is None else got_lo.numpy(), None if got_hi is None else got_hi.numpy())


def make_problem(kind):
    from krylov_amd import problems as P
    rng = np.random.default_rng(9)
    if kind == "const":
        p = P.reference_test1(16)
        p.rhs = P.interior_only(rng.uniform(-1, 1, p.rhs.shape))
    elif kind == "var":
        p = P.smooth_var(16)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    else:
        p = P.offdiag_var(16)
        p.rhs = rng.uniform(-1, 1, p.rhs.shape)
    return p


def worker(rank, world, port, kind, L, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from krylov_amd import problems as P
    from oracle import oracle as orc
    import slab_reference as SR
    p = make_problem(kind)
    local = SR.split_problem(P, p, world, rank)
    phi, hist, nrhs = SR.SlabSolver(orc, GlooComm(rank, world), rank, world, local, L, 1e-30, 4).solve()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), phi=phi, hist=np.array(hist), nrhs=nrhs)
    dist.barrier()
    dist.destroy_process_group()


def brick_worker(rank, world, port, kind, L, nproc, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from krylov_amd import problems as P
    from oracle import oracle as orc
    import slab_reference as SR
    p = make_problem(kind)
    coord = (rank // (nproc[1] * nproc[2]), (rank // nproc[2]) % nproc[1], rank % nproc[2])
    local = SR.split_brick(P, p, nproc, coord)
    phi, hist, nrhs = SR.BrickSolver(orc, GlooComm(rank, world), nproc, coord, local, L, 1e-30, 4).solve()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), phi=phi, hist=np.array(hist), nrhs=nrhs)
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("kind", ["const", "var", "offdiag"])
def test_slab_schedule_reproduces_the_undivided_solve(orc, tmp_path, kind, L, world):
    from helpers import oracle_solve
    p = make_problem(kind)
    want = oracle_solve(orc, p, L=L, eps=1e-30, max_outer=4)
    mp.start_processes(worker, args=(world, free_port(), kind, L, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    mzl = p.mx[2] // world
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert float(d["nrhs"]) == pytest.approx(want["norm_rhs"], rel=1e-13)
        assert np.max(np.abs(d["hist"] - want["history"][:4])) <= 1e-10 * want["norm_rhs"]
        # owned planes 0..mzl of the slab against the undivided solution
        mine = d["phi"][1:mzl + 2, 1:-1, 1:-1]
        ref = want["phi"][r * mzl + 1: r * mzl + mzl + 2, 1:-1, 1:-1]
        assert np.max(np.abs(mine - ref)) <= 1e-10 * np.max(np.abs(ref))
        # ghost planes carried without exchange still equal the neighbour's planes
        if r > 0:
            assert np.max(np.abs(d["phi"][0, 1:-1, 1:-1] - want["phi"][r * mzl, 1:-1, 1:-1])) \
                <= 1e-10 * np.max(np.abs(ref))


@pytest.mark.parametrize("nproc", [(2, 1, 1), (1, 2, 2), (2, 2, 2)])
@pytest.mark.parametrize("kind,L", [("const", 2), ("var", 1), ("offdiag", 2)])
def test_cartesian_schedule_reproduces_the_undivided_solve(orc, tmp_path, kind, L, nproc):
    """Sub-domains in x and y too (up to the reference driver's 2 x 2 x 2): exchange
    order x, y, z over full tangential boxes, vector updates carried on all ghost
    layers, interface points shared with weights 1/2, 1/4, 1/8."""
    from helpers import oracle_solve
    p = make_problem(kind)
    want = oracle_solve(orc, p, L=L, eps=1e-30, max_outer=4)
    world = nproc[0] * nproc[1] * nproc[2]
    mp.start_processes(brick_worker, args=(world, free_port(), kind, L, nproc, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    m = [p.mx[d] // nproc[d] for d in range(3)]
    for r in range(world):
        c = (r // (nproc[1] * nproc[2]), (r // nproc[2]) % nproc[1], r % nproc[2])
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert float(d["nrhs"]) == pytest.approx(want["norm_rhs"], rel=1e-13)
        assert np.max(np.abs(d["hist"] - want["history"][:4])) <= 1e-10 * want["norm_rhs"]
        mine = d["phi"][1:-1, 1:-1, 1:-1]
        ref = want["phi"][c[2] * m[2] + 1: c[2] * m[2] + m[2] + 2, c[1] * m[1] + 1: c[1] * m[1] + m[1] + 2,
                          c[0] * m[0] + 1: c[0] * m[0] + m[0] + 2]
        assert np.max(np.abs(mine - ref)) <= 1e-10 * np.max(np.abs(ref))
