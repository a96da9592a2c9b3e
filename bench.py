#!/usr/bin/env python3
"""Throughput of the BiCGStab(l) stencil solve on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is one pass of the outer do{}while loop of the reference's
solve_int (BICGStab3D/solveLin_BICGStab.C:327-496): 2*l operator
applications, the vector updates and all dot products.  The workload is the
512^3 fp64 variable-coefficient stencil (Dxy = 0) with Dirichlet faces; with N
GPUs the grid is split into N z-slabs (strong scaling: total work fixed).
Fields are generated on the host, staged to HBM and the solve's prologue is
run before the timed region; the timed region is K passes on resident data.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (HIP events
recorded on the solver's own stream around every launch of the timed region);
`cpu_baseline` is the CPU oracle (oracle/, a line-faithful restatement of the
reference's serial path) timed on one host core on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

KIND_NAMES = {0: "fused_A0", 1: "fused_A1", 2: "fused_B0", 3: "fused_B0L", 4: "fused_B1L",
              5: "fused_residual", 6: "mgs2", 7: "final"}


def kernel_words(L, var, offd):
    """Doubles read + written per grid point by each kernel of one pass (every
    array touched once per kernel, stencil neighbours served on chip)."""
    d = (4 if offd else 3) if var else 0
    if L == 1:
        return {0: 7 + d,          # r0,u0,u1(old),lam,rt -> u0,u1
                3: 5 + d,          # r0,u1,lam -> r0,r1
                7: 7}              # phi,u0,r0,r1,rt -> phi,r0   (the u0 update rides in the next pass's first kernel)
    return {0: 6 + d,              # r0,u0,lam,rt -> u0,u1
            2: 6 + d,              # r0,u1,lam,rt -> r0,r1
            1: 6 + d,              # r1,u1,lam,rt -> u1,u2
            4: 5 + d,              # r1,u2,lam -> r1,r2
            6: 4,                  # r2,r1,r0,u1 -> three dots (r2' and r0'' are recomputed by the last kernel)
            7: 11}                 # phi,r0..r2,u0,u1,u2,rt -> phi,r0,u0


def measured_traffic(kernel_name, n, L, workload, world):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic.json), if they were taken on this very workload."""
    import glob
    if workload == "mixed":
        return None, None
    want = "%d^3 fp64 %s-coefficient, l=%d, %d GPU" % (n, "variable" if workload == "var" else "constant", L, world)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == want and kernel_name in d.get("kernels", {}):
            return d["kernels"][kernel_name]["hbm_bytes_per_launch"], os.path.basename(path)
    return None, None


def slab_problem(n, nslabs, slab, workload, nz=None):
    """Fields of z-slab `slab` of the n x n x nz problem (nz = n unless given),
    ghost planes included, evaluated from closed forms at global positions."""
    from krylov_amd import problems as P
    nz = n if nz is None else nz
    mzl = nz // nslabs
    delx = (1.0 / n, 1.0 / n, 1.0 / nz)
    ax = [np.arange(-1, n + 2, dtype=np.float64) * delx[0],
          np.arange(-1, n + 2, dtype=np.float64) * delx[1],
          (np.arange(-1, mzl + 2, dtype=np.float64) + slab * mzl) * delx[2]]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij", sparse=True)
    pi = np.pi
    shape = (mzl + 3, n + 3, n + 3)
    full = lambda a: np.ascontiguousarray(np.broadcast_to(a, shape))
    f = {}
    f["lam"] = full(0.2 * x * (y * y) * z)
    sinx, siny, sinz = np.sin(pi * x), np.sin(pi * y), np.sin(pi * z)
    f["bc"] = (0,) * 6
    if workload == "mixed":
        # anisotropic high-contrast coefficients (contrast 1e4, anisotropy 100) with mixed
        # faces: x Dirichlet, y extrapolating, z low Dirichlet, z high extrapolating
        e = 10.0 ** (2.0 * np.sin(2 * pi * x) * np.sin(2 * pi * y) * np.sin(2 * pi * z))
        f["Dxx"] = full(e)
        f["Dyy"] = full(0.1 * e)
        f["Dzz"] = full(10.0 * e)
        f["lam"] = np.full(shape, 1e-3)
        f["rhs"] = full(sinx * siny * sinz)
        f["Dc"] = (1., 1., 1.)
        f["bc"] = (0, 0, 1, 1, 0, 1)
    elif workload == "var":
        f["Dxx"] = full(1.0 + 0.5 * np.sin(2 * pi * x) * np.cos(2 * pi * y))
        f["Dyy"] = full(1.0 + 0.5 * np.sin(2 * pi * y) * np.cos(2 * pi * z))
        f["Dzz"] = full(1.0 + 0.5 * np.sin(2 * pi * z) * np.cos(2 * pi * x))
        f["rhs"] = full(sinx * siny * sinz)
        f["Dc"] = (1., 1., 1.)
    else:
        # the reference driver's test 1 (BICGStab3D/main.C:223-229,277,282-284)
        Dc = (1., 1., 10.)
        f["rhs"] = full(-((pi * pi) * (Dc[0] + Dc[1] + Dc[2]) + f["lam"]) * (sinx * siny * sinz))
        f["Dc"] = Dc
    f["phi"] = np.zeros(shape)
    f["mx"] = (n, n, mzl)
    f["delx"] = delx
    return f


def _cpu_run(n_sample, L, workload, n_full, budget_iters, threads):
    """One timed oracle solve in a child process (the OpenMP thread count is fixed at start-up)."""
    import subprocess
    code = (
        "import sys, json, time, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "from oracle import oracle as orc\n"
        "orc.build()\n"
        "%s\n"
        "f = bench.slab_problem(%d, 1, 0, %r)\n"
        "var = %r != 'const'\n"
        "cfg = orc.make_cfg(f['mx'], f['delx'], f['bc'], L=%d, eps=0.0, max_outer=%d, var_coef=var, "
        "use_offdiag=False, Dc=f['Dc'])\n"
        "phi = f['phi'].copy()\n"
        "z = np.zeros_like(phi) if var else None\n"
        "t0 = time.perf_counter()\n"
        "r = orc.solve(cfg, phi, f['rhs'].copy(), f['lam'].copy(), f.get('Dxx'), f.get('Dyy'), f.get('Dzz'), z)\n"
        "print(json.dumps({'iters': r['iters'], 'seconds': time.perf_counter() - t0}))\n"
    ) % (ROOT, "orc.use_openmp_build()" if threads > 1 else "", n_sample, workload, workload, L, budget_iters)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    rate = max(d["iters"], 1) / d["seconds"]
    return rate, rate * ((n_sample + 1) ** 3) / float((n_full + 1) ** 3), max(d["iters"], 1)


def cpu_baseline(n_sample, L, workload, n_full, budget_iters):
    """The CPU oracle (oracle/krylov_oracle.c, a line-faithful restatement of the
    reference's serial solver with its un-fused pass structure) on the same
    workload at n_sample^3 for a few passes, scaled by grid points to n_full^3.
    Reported: the multi-core figure (the same source with its loops spread over
    8 threads by OpenMP, standing in for the reference's 2x2x2 MPI ranks) and,
    in `single_core`, the plain serial build."""
    cores = min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    r1, v1, it1 = _cpu_run(n_sample, L, workload, n_full, budget_iters, 1)
    out = {"value": v1, "unit": "iter/s", "cores": 1, "kind": "port",
           "sample": "%d passes (plus prologue/epilogue residuals) of the same %s-coefficient l=%d solve at %d^3: "
                     "%.3f iter/s measured on 1 core, scaled by grid points to %d^3"
                     % (it1, workload, L, n_sample, r1, n_full)}
    if cores > 1:
        try:
            rc, vc, itc = _cpu_run(n_sample, L, workload, n_full, 2 * budget_iters, cores)
            out = {"value": vc, "unit": "iter/s", "cores": cores, "kind": "port",
                   "sample": "%d passes (plus prologue/epilogue residuals) of the same %s-coefficient l=%d solve at "
                             "%d^3: %.3f iter/s measured on %d cores (OpenMP build of the oracle), scaled by grid "
                             "points to %d^3" % (itc, workload, L, n_sample, rc, cores, n_full),
                   "single_core": {"value": v1, "measured_at_sample": r1}}
        except Exception as e:          # keep the single-core figure
            out["sample"] += " (multi-core run failed: %s)" % e
    return out


def claim_stdout():
    """Routes file descriptor 1 to stderr for the rest of the run and returns a
    descriptor of the real stdout: libraries underneath (gloo, RCCL) print start-up
    notes on fd 1, and stdout must carry the one JSON line only."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=512, help="grid size (points 0..n per direction)")
    ap.add_argument("--ell", dest="L", type=int, default=2, help="BiCGStab(l) degree; the reference drivers use 2")
    ap.add_argument("--workload", choices=["var", "const", "mixed"], default="var",
                    help="var: smooth field coefficients; const: constant coefficients (both with Dirichlet faces); "
                         "mixed: high-contrast anisotropic field coefficients with mixed face types")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: a GRID x GRID x 128 block per GPU (GRID = 1024 gives the 1024^3 problem on 8 GPUs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-grid", dest="cpu_n", type=int, default=192)
    ap.add_argument("--cpu-iters", type=int, default=4)
    args = ap.parse_args()
    real_stdout = claim_stdout()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    # per-kernel HIP events only where the roofline is reported (one GPU)
    os.environ.setdefault("KRY_PROFILE", "1" if world == 1 else "0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: the control plane stays on loopback

    # the solver library first (it binds the ROCm installation's HIP runtime),
    # torch.distributed afterwards and for the control plane only
    from krylov_amd import capi
    capi.lib()
    capi.device_check(int(os.environ.get("KRY_BENCH_DEVICE", local_rank)))

    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)

    n, L, var = args.n, args.L, args.workload != "const"
    nz = 128 * world if args.weak else n
    if nz % world != 0:
        sys.exit("grid size %d is not divisible by %d slabs" % (nz, world))
    f = slab_problem(n, world, rank, args.workload, nz)

    device = int(os.environ.get("KRY_BENCH_DEVICE", local_rank))
    solver = capi.Solver(f["mx"], f["delx"], f["bc"], L=L, eps=0.0, max_outer=0, nslabs=world,
                         slab_begin=rank, slab_count=1, devices=(device,))
    transport = "none"
    if world > 1:
        import torch
        transport = "rccl"
        try:
            uid = [capi.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            solver.comm_init(uid[0], rank, world)
            ok_local = 1
        except Exception as e:           # no usable RCCL: keep the run alive on a host-staged path
            sys.stderr.write("rank %d: RCCL communicator unavailable (%s)\n" % (rank, e))
            ok_local = 0
        flag = torch.tensor([ok_local])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            from krylov_amd import gloo_transport
            transport = "gloo-staged (RCCL unavailable)"
            solver.close()
            solver = capi.Solver(f["mx"], f["delx"], f["bc"], L=L, eps=0.0, max_outer=0, nslabs=world,
                                 slab_begin=rank, slab_count=1, devices=(device,))
            solver.comm_init_callbacks(rank, world, *gloo_transport.make(rank, world))

    def barrier():
        if dist is not None:
            dist.barrier()

    def load():
        if var:
            solver.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
        else:
            solver.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])

    def all_ok(mine):
        flag = torch.tensor([1 if mine else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag[0]) == 1

    # Exchange between the ranks: through the transport (per dot product one RCCL all-reduce and
    # two small kernels, per operator output one grouped send/receive), with the dot products
    # all-reduced by the reduction kernels themselves through device mailboxes ("mailbox"), or
    # with the ghost planes travelling through mapped device memory as well ("mapped memory").
    # All are tried on a few passes of this very problem; the fastest one that reproduces the
    # transport's residuals is used for the measurement.  KRY_BENCH_REDUCE=transport skips the
    # trial, =mailbox / ="mapped memory" prefers that route when it is valid.
    reduction, trial = "none", None
    if world > 1:
        reduction = "transport"
        want = os.environ.get("KRY_BENCH_REDUCE", "auto")
        if want != "transport":
            os.environ.setdefault("KRY_MAIL_TIMEOUT_MS", "5000")
            try:
                mine = solver.mail_export()
            except Exception as e:
                sys.stderr.write("rank %d: no mailbox (%s)\n" % (rank, e))
                mine = None
            handles = [None] * world
            dist.all_gather_object(handles, mine)
            connected = False
            if all(h is not None for h in handles):
                try:
                    solver.mail_connect(rank, world, handles)
                    connected = True
                except Exception as e:
                    sys.stderr.write("rank %d: mailboxes not connected (%s)\n" % (rank, e))
            if not all_ok(connected):
                if connected:
                    solver.mail_enable(0)
            else:
                trial, hists = {}, {}
                modes = {"mailbox": capi.KRY_MAIL_SUMS, "mapped memory": capi.KRY_MAIL_SUMS | capi.KRY_MAIL_PLANES,
                         "transport": 0}
                for name, on in modes.items():
                    solver.mail_enable(on)
                    barrier()
                    good, dt = True, float("inf")
                    try:
                        load()
                        solver.iterate(2)
                        barrier()
                        t0 = time.perf_counter()
                        good = solver.iterate(6) == 0
                        dt = time.perf_counter() - t0
                        hists[name] = solver.history()[:8]
                    except Exception as e:
                        sys.stderr.write("rank %d: exchange by %s failed (%s)\n" % (rank, name, e))
                        good = False
                    tt = torch.tensor([dt if good else float("inf")], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    trial[name] = float(tt[0]) / 6 * 1e3
                ref = hists.get("transport")
                best = "transport"
                for name in ("mailbox", "mapped memory"):
                    hh = hists.get(name)
                    same = (ref is not None and hh is not None and len(hh) == len(ref) == 8 and
                            bool(np.all(np.abs(hh[:3] - ref[:3]) <= 1e-9 * np.abs(ref[:3]))))
                    if all_ok(same) and (trial[name] < trial[best] or want == name):
                        best = name
                solver.mail_enable(modes[best])
                reduction = best
        barrier()

    t_stage = time.perf_counter()
    load()
    t_stage = time.perf_counter() - t_stage

    st = solver.iterate(args.warmup) if args.warmup > 0 else 0
    s0 = solver.stats()
    k_ms0 = list(s0.kernel_ms)
    k_n0 = list(s0.kernel_calls)

    barrier()
    t0 = time.perf_counter()
    st = solver.iterate(args.steps)          # returns after a stream synchronise
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    s1 = solver.stats()
    hist = solver.history()
    ok = (st == 0 and s1.iterations == args.warmup + args.steps and np.all(np.isfinite(hist)))

    # Several GPUs: the timed region runs without per-kernel events (rank 0 would lag behind the
    # others by their cost); the kernel table and the roofline come from a few extra passes.
    probe = 0
    if world > 1 and all_ok(ok):          # (a collective decision: every rank runs the extra passes or none)
        try:
            solver.set_profile(True)
            k_ms0, k_n0 = list(s1.kernel_ms), list(s1.kernel_calls)
            probe = 6
            solver.iterate(probe)
            s1 = solver.stats()
        except Exception as e:
            sys.stderr.write("rank %d: no per-kernel times (%s)\n" % (rank, e))
            probe = 0

    if rank == 0:
        npts = float(n + 1) ** 2 * (nz + 1)
        words = kernel_words(L, var, False)
        kern = {}
        for k, w in words.items():
            calls = s1.kernel_calls[k] - k_n0[k]
            ms = s1.kernel_ms[k] - k_ms0[k]
            if calls > 0:
                kern[k] = dict(name=KIND_NAMES[k], calls=calls, avg_ms=ms / calls, words=w,
                               gbs=8.0 * w * npts / world / (ms / calls * 1e-3) / 1e9)
        dom = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["calls"]) if kern else None
        roof = None
        if dom is not None:
            roof = {"bound": "hbm", "kernel": kern[dom]["name"], "achieved": round(kern[dom]["gbs"], 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kern[dom]["gbs"] / HBM_PEAK_GBS, 4),
                    "traffic": None, "avg_launch_ms": round(kern[dom]["avg_ms"], 4),
                    "algorithmic_bytes_per_launch": 8.0 * kern[dom]["words"] * npts / world}
            if probe:
                roof["measured_on"] = "%d passes after the timed region, rank 0; launches include waits for ghost planes" % probe
            tr, src = measured_traffic(kern[dom]["name"], n, L, args.workload, world)
            if tr is not None:
                roof["traffic"] = tr
                roof["traffic_source"] = "profiles/" + src
        ms_step = elapsed / args.steps * 1e3
        W = {1: (26 if var else 20), 2: (62 if var else 50)}[L]
        whole = 8.0 * W * npts / (ms_step * 1e-3) / 1e9
        out = {
            "metric": "BiCGStab iters/sec + achieved HBM GB/s (fp64 512^3) at 1/2/4/8 MI355X",
            "value": round(args.steps / elapsed, 3), "unit": "iter/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s fp64 %s-coefficient 3D stencil, BiCGStab(l=%d), %s faces, "
                                   "%d z-slab(s)" % ("%d^3" % n if nz == n else "%dx%dx%d" % (n, n, nz),
                                                     "variable" if var else "constant", L,
                                                     "mixed Dirichlet/extrapolating" if args.workload == "mixed"
                                                     else "Dirichlet", world),
                       "grid": n, "l": L, "coefficients": args.workload, "parallelism": "z-slabs x%d" % world,
                       "transport": transport, "exchange": reduction,
                       "exchange_trial_ms_per_pass": trial},
            "valid": bool(ok), "solver_path": s1.fused,
            "iteration_GBps": round(whole, 1),
            "iteration_frac_of_hbm_peak": round(whole / HBM_PEAK_GBS / world, 4),
            "iteration_words_per_point": W,
            "stage_s": round(t_stage, 3),
            "residual_after": float(hist[-1]) if len(hist) else None,
            "kernels": {v["name"]: {"avg_ms": round(v["avg_ms"], 4), "GBps": round(v["gbs"], 1),
                                    "words": v["words"], "calls": v["calls"]} for v in kern.values()},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu and not args.weak:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, L, args.workload, n, args.cpu_iters)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
