"""Worker of tests/test_gpu_multiproc.py: one process = one z slab, all
processes share GPU 0.  The inter-process transport is the callback form of
the C ABI, implemented with device<->host copies and torch.distributed (gloo);
everything else is the library's own multi-process code path (the same one the
RCCL transport drives on a multi-GPU node).

usage: mp_slab_worker.py RANK WORLD PORT KIND L MAX_OUTER EPS OUTDIR [NX,NY,NZ]
(with a split NX,NY,NZ the processes form a Cartesian grid, rank = (cx*NY + cy)*NZ + cz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kind, L, max_outer, eps, outdir = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]), sys.argv[8]
    nproc = tuple(int(v) for v in sys.argv[9].split(",")) if len(sys.argv) > 9 else None

    # the solver library (and with it the ROCm HIP runtime) before torch
    from krylov_amd import capi, problems as P
    capi.lib()
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from krylov_amd import gloo_transport
    allreduce, halo = gloo_transport.make(rank, world)
    use_rccl = os.environ.get("KRY_TEST_RCCL") == "1"      # one GPU per process, RCCL as the transport
    dev = rank if use_rccl else 0

    import slab_reference as SR
    from test_gpu_multiproc import make_problem
    p = make_problem(kind)
    flags = capi.KRY_FORCE_GENERIC if kind.endswith("_generic") else 0
    if nproc is None:
        loc = SR.split_problem(P, p, world, rank)
        solver = capi.Solver(loc.mx, loc.delx, loc.bc, L=L, eps=eps, max_outer=max_outer, flags=flags, nslabs=world,
                             slab_begin=rank, slab_count=1, devices=(dev,))
    else:
        coord = (rank // (nproc[1] * nproc[2]), (rank // nproc[2]) % nproc[1], rank % nproc[2])
        loc = SR.split_brick(P, p, nproc, coord)
        solver = capi.Solver(loc.mx, loc.delx, loc.bc, L=L, eps=eps, max_outer=max_outer, flags=flags, devices=(dev,),
                             nproc=nproc, coord=coord)
    with solver as s:
        if use_rccl:
            uid = [capi.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            s.comm_init(uid[0], rank, world)
        elif os.environ.get("KRY_TEST_NO_TRANSPORT") != "1":
            s.comm_init_callbacks(rank, world, allreduce, halo)
        mail = int(os.environ.get("KRY_TEST_MAIL", "0"))   # 1: dot products through device mailboxes, 2: ghost
        if mail:                                            # planes through mapped memory, 3: both
            handles = [None] * world
            dist.all_gather_object(handles, s.mail_export())
            s.mail_connect(rank, world, handles)
            if os.environ.get("KRY_TEST_NO_TRANSPORT") != "1":
                s.mail_enable(mail)
        if os.environ.get("KRY_TEST_MAIL_ABSENT") == "2":  # the last rank vanishes after the load: every exchange of the
            import time                                     # batch of passes queued behind it finds no partner
            s.load_var(loc.phi.copy(), loc.rhs, loc.lam, loc.Dxx, loc.Dyy, loc.Dzz, loc.Dxy, loc.use_offdiag)
            dist.barrier()
            if rank == world - 1:
                time.sleep(6.0)                              # (its mailbox stays mapped meanwhile)
                os._exit(0)
            t0 = time.time()
            try:
                s.iterate(8)
                msg = "no error"
            except capi.KryError as e:
                msg = str(e)
            with open(os.path.join(outdir, "r%d.err" % rank), "w") as f:
                f.write("%.3f s: %s" % (time.time() - t0, msg))
            os._exit(0)
        if os.environ.get("KRY_TEST_MAIL_ABSENT") == "1":  # the last rank never shows up at the first dot product
            import time
            if rank == world - 1:
                time.sleep(4.0)                              # (its mailbox stays mapped meanwhile)
                os._exit(0)
            try:
                s.solve_var(loc.phi.copy(), loc.rhs, loc.lam, loc.Dxx, loc.Dyy, loc.Dzz, loc.Dxy, loc.use_offdiag)
                msg = "no error"
            except capi.KryError as e:
                msg = str(e)
            with open(os.path.join(outdir, "r%d.err" % rank), "w") as f:
                f.write(msg)
            os._exit(0)
        phi = loc.phi.copy()
        if loc.var_coef:
            st = s.solve_var(phi, loc.rhs, loc.lam, loc.Dxx, loc.Dyy, loc.Dzz, loc.Dxy, loc.use_offdiag)
        else:
            st = s.solve_const(phi, loc.rhs, loc.lam, loc.Dc)
        stats = s.stats()
        hist = s.history()
    np.savez(os.path.join(outdir, "r%d.npz" % rank), phi=phi, hist=hist, status=st, iters=stats.iterations,
             fused=stats.fused, norm_rhs=stats.norm_rhs, overlapped=stats.overlapped_exchanges)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
