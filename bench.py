#!/usr/bin/env python3
"""Throughput of the BiCGStab(l) stencil solve on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" is one pass of the outer do{}while loop of the reference's
solve_int (BICGStab3D/solveLin_BICGStab.C:327-496): 2*l operator
applications, the vector updates and all dot products.  The workload is the
512^3 fp64 variable-coefficient stencil (Dxy = 0) with Dirichlet faces; with N
GPUs the grid is split into N z-slabs (strong scaling: total work fixed), one
process per GPU.  Called with --gpus N > 1 outside torch.distributed.run, the
script starts the N rank processes itself (before anything in it has touched a
GPU) and relays rank 0's line.  Fields are generated on the host, staged to HBM
and the solve's prologue is run before the timed region; the timed region is K
passes on resident data.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (HIP events
recorded on the solver's own stream around every launch of the timed region);
`cpu_baseline` is the CPU oracle (oracle/, a line-faithful restatement of the
reference's CPU path) timed on the host cores on a bounded sample.  On one GPU
the line also carries, under `also`, a few passes each of l = 1 on the same grid
and of the 256^3 constant-coefficient problem (l = 1 and 2).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

KIND_NAMES = {0: "fused_A0", 1: "fused_A1", 2: "fused_B0", 3: "fused_B0L", 4: "fused_B1L",
              5: "fused_residual", 6: "mgs2", 7: "final"}
METRIC = "BiCGStab iters/sec + achieved HBM GB/s (fp64 512^3) at 1/2/4/8 MI355X"


def kernel_words(L, var, offd):
    """Doubles read + written per grid point by each kernel of one pass (every
    array touched once per kernel, stencil neighbours served on chip)."""
    d = (4 if offd else 3) if var else 0
    if L == 1:
        return {0: 7 + d,          # r0,u0,u1(old),lam,rt -> u0,u1
                3: 5 + d,          # r0,u1,lam -> r0,r1
                7: 7}              # phi,u0,r0,r1,rt -> phi,r0   (the u0 update rides in the next pass's first kernel)
    if L > 2:
        return {}                  # the timing slots of l > 2 hold several kernels each: no per-kernel table
    return {0: 6 + d,              # r0,u0,lam,rt -> u0,u1
            2: 6 + d,              # r0,u1,lam,rt -> r0,r1
            1: 6 + d,              # r1,u1,lam,rt -> u1,u2
            4: 5 + d,              # r1,u2,lam -> r1,r2
            6: 4,                  # r2,r1,r0,u1 -> three dots (r2' and r0'' are recomputed by the last kernel)
            7: 11}                 # phi,r0..r2,u0,u1,u2,rt -> phi,r0,u0


def contract_words(L, var):
    """Doubles per point and pass of the fully fused schedule the metric's byte model is defined on."""
    if L > 2:
        return moved_words(L, var)  # no contract figure exists for l > 2: the words the kernels move
    return {1: (26 if var else 20), 2: (62 if var else 50)}[L]


def moved_words(L, var):
    """Doubles per point and pass the kernels move."""
    if L > 2:
        # 2l stencil kernels, the pointwise sweeps (2 l^2 + l - 3), the Gram-Schmidt kernels (2 l^2), the end of the pass (2l + 7)
        return 2 * L * (9 if var else 6) + 4 * L * L + 3 * L + 4
    return sum(kernel_words(L, var, False).values())


def measured_traffic(kernel_name, n, L, workload, world):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes
    (profiles/*_hbm_traffic.json), if they were taken on this very workload."""
    import glob
    want = "%d^3 fp64 %s-coefficient, l=%d, %d GPU" % (n, {"var": "variable", "const": "constant",
                                                            "mixed": "mixed-face variable"}[workload], L, world)
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
    elif workload == "poisson":
        # the 64^3 plumbing case: D = (1,1,1), lambda = 0, rhs = -3 pi^2 sin sin sin
        f["lam"] = np.zeros(shape)
        f["rhs"] = full(-3.0 * (pi * pi) * (sinx * siny * sinz))
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


# --------------------------------------------------------------------------- CPU leg

def _cpu_start(n, L, workload, max_outer, eps, threads, split=None):
    """Starts one timed oracle solve in a child process (thread counts are fixed at start-up).
    split: None = the serial restatement (or its OpenMP build when threads > 1);
    (px,py,pz) = the restatement of the reference's `-D parallel` path with one thread per
    sub-domain of that Cartesian split.  The child times the solve itself (not its set-up)."""
    code = (
        "import sys, json, time, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "from oracle import oracle as orc\n"
        "orc.build()\n"
        "%s\n"
        "f = bench.slab_problem(%d, 1, 0, %r)\n"
        "var = %r in ('var', 'mixed')\n"
        "cfg = orc.make_cfg(f['mx'], f['delx'], f['bc'], L=%d, eps=%r, max_outer=%d, var_coef=var, "
        "use_offdiag=False, Dc=f['Dc'])\n"
        "phi = f['phi']\n"
        "z = np.zeros_like(phi) if var else None\n"
        "split = %r\n"
        "t0 = time.perf_counter()\n"
        "if split: r = orc.solve_ranks(cfg, split, phi, f['rhs'], f['lam'], f.get('Dxx'), f.get('Dyy'), f.get('Dzz'), z)\n"
        "else: r = orc.solve(cfg, phi, f['rhs'], f['lam'], f.get('Dxx'), f.get('Dyy'), f.get('Dzz'), z)\n"
        "print(json.dumps({'iters': r['iters'], 'seconds': time.perf_counter() - t0, 'status': r['status'],"
        " 'final': r['norm_final'] / r['norm_rhs']}))\n"
    ) % (ROOT, "orc.use_openmp_build()" if (threads > 1 and not split) else "", n, workload, workload, L, eps,
         max_outer, split)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
    return subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _cpu_finish(proc, timeout=900):
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        raise RuntimeError("oracle run exceeded %d s" % timeout)
    if proc.returncode != 0:
        raise RuntimeError(err.strip().splitlines()[-1] if err.strip() else "oracle run failed")
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])   # (the oracle prints the reference's NaN notes)
    d["iters"] = max(d["iters"], 1)
    d["rate"] = d["iters"] / d["seconds"]
    return d


def _cpu_run(n, L, workload, max_outer, eps, threads, split=None):
    return _cpu_finish(_cpu_start(n, L, workload, max_outer, eps, threads, split))


def cpu_baseline(L, workload, n_full, n_sample, iters, cores_wanted):
    """The CPU oracle next to the GPU line (kind "port": oracle/krylov_oracle.c restates the
    reference's serial path with its un-fused pass structure, oracle/krylov_oracle_ranks.c its
    `-D parallel` path with the driver's Cartesian split, one thread per MPI rank), timed on the
    headline workload at its full size n_full^3:
      * `value`: `iters` passes on P = 8 ranks (2 x 2 x 2, BICGStab3D/main.C:146-150);
      * `single_core`: 3 passes of the serial restatement on one core;
        (the two legs run side by side on 9 of the host's cores, after all GPU work)
      * `sample_small`: the 8-rank leg once more at n_sample^3 (what earlier lines extrapolated from);
      * `c1`: the 64^3 plumbing case solved to 1e-6 in full on one core: the reference driver's
        test 1 (constant coefficients, l = 2) and the pure Poisson problem with l = 1 (its
        right-hand side is an eigenvector of the operator: one pass; with l = 2 the reference
        breaks down on it with "alpha is nan")."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    desc = "%s-coefficient l=%d" % ("variable" if workload != "const" else "constant", L)
    out = {"unit": "iter/s", "kind": "port"}
    many_ok = avail >= 8 and cores_wanted >= 8
    side_by_side = many_ok and avail >= 9
    p_many = _cpu_start(n_full, L, workload, iters, 0.0, 8, split=(2, 2, 2)) if many_ok else None
    p_one = _cpu_start(n_full, L, workload, 3, 0.0, 1) if (side_by_side or not many_ok) else None
    many = None
    if p_many is not None:
        try:
            many = _cpu_finish(p_many, 600)
        except Exception as e:
            out["many_core_error"] = str(e)
    if p_one is None:
        p_one = _cpu_start(n_full, L, workload, 3, 0.0, 1)
    one = _cpu_finish(p_one, 600)
    out["single_core"] = {"value": one["rate"], "cores": 1,
                          "sample": "%d passes (plus prologue/epilogue residuals) of the %s solve at %d^3, serial "
                                    "restatement on 1 core, %.1f s" % (one["iters"], desc, n_full, one["seconds"])}
    out.update(value=one["rate"], cores=1, sample=out["single_core"]["sample"])
    if many is not None:
        out.update(value=many["rate"], cores=8,
                   sample="%d passes (plus prologue/epilogue residuals) of the %s solve at %d^3 on 8 host cores: "
                          "restatement of the reference's MPI path, 2x2x2 sub-domains, one thread per rank, %.1f s%s"
                          % (many["iters"], desc, n_full, many["seconds"],
                             "; the single-core leg ran beside it on a ninth core" if side_by_side else ""))
        if n_sample and n_sample != n_full:
            try:
                small = _cpu_run(n_sample, L, workload, iters, 0.0, 8, split=(2, 2, 2))
                out["sample_small"] = {"grid": n_sample, "value": small["rate"], "cores": 8, "passes": small["iters"],
                                       "scaled_by_points_to_full_size": small["rate"] * ((n_sample + 1) ** 3)
                                       / float((n_full + 1) ** 3)}
            except Exception as e:
                out["sample_small"] = {"grid": n_sample, "error": str(e)}
    out["c1"] = []
    for wl, cl, name in (("const", 2, "reference driver test 1 (D = 1,1,10; lambda = 0.2 x y^2 z)"),
                         ("poisson", 1, "Poisson (D = 1,1,1; lambda = 0)")):
        try:
            c1 = _cpu_run(64, cl, wl, 0, 1e-6, 1)
            out["c1"].append({"workload": "64^3 %s, Dirichlet, l=%d, eps 1e-6, serial restatement, 1 core, full solve"
                                          % (name, cl), "iterations": c1["iters"], "status": c1["status"],
                              "seconds": round(c1["seconds"], 3), "iter_per_s": round(c1["rate"], 2),
                              "relative_residual": c1["final"]})
        except Exception as e:
            out["c1"].append({"workload": name, "error": str(e)})
    return out


# ------------------------------------------------------------------------- launcher

def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(argv, n, timeout=None):
    """Start n rank processes of this script (fresh interpreters: nothing here has touched a
    GPU), rank 0's stdout is relayed, everything else goes to stderr.  Returns the exit code."""
    port = free_port()
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                KRY_BENCH_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    deadline = time.time() + (timeout or float(os.environ.get("KRY_BENCH_TIMEOUT", "1500")))
    line, rc = b"", 0
    try:
        line = procs[0].communicate(timeout=max(1.0, deadline - time.time()))[0]
        for q in procs[1:]:
            q.wait(timeout=max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    for q in procs:
        if q.returncode not in (0, None):
            rc = rc or q.returncode
    sys.stdout.write(line.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def claim_stdout():
    """Routes file descriptor 1 to stderr for the rest of the run and returns a
    descriptor of the real stdout: libraries underneath (gloo, RCCL) print start-up
    notes on fd 1, and stdout must carry the one JSON line only."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


# ---------------------------------------------------------------------- measurement

def choose_headline(runs, transport, rccl_error, chosen_by_trial, trial_note):
    """Which of the fully timed routes the line's `value` reports, and why.
    runs: {route: {"valid": bool, "ms_per_step": float}} with "transport" always present.
    The RCCL run whenever RCCL carried the transport and its run is valid; otherwise (RCCL could not
    start, so the transport is the host-staged stand-in, or its run failed) the fastest valid run;
    with no valid run at all the transport's (the caller reports the failure)."""
    valid = {k: r for k, r in runs.items() if r.get("valid")}
    if transport == "rccl" and "transport" in valid:
        note = ("the trial's fastest route (%s) was timed in full as well, see `routes`" % chosen_by_trial
                if chosen_by_trial != "transport" else "no faster valid route was found (%s)" % trial_note)
        return "transport", "RCCL send/receive and all-reduce, the route the solver is specified on; " + note
    if valid:
        name = min(valid, key=lambda k: valid[k]["ms_per_step"])
        return name, ("RCCL could not be used (%s): the transport here is the host-staged stand-in; headline = fastest "
                      "valid full run of the routes timed (`routes`)" % (rccl_error or "its timed run failed"))
    return "transport", "no route produced a valid run"


def kernel_table(s0, s1, L, var, npts_local):
    """Per-kernel average launch time and algorithmic rate between two kry_stats snapshots."""
    kern = {}
    for k, w in kernel_words(L, var, False).items():
        calls = s1.kernel_calls[k] - s0.kernel_calls[k]
        ms = s1.kernel_ms[k] - s0.kernel_ms[k]
        if calls > 0:
            kern[k] = dict(name=KIND_NAMES[k], calls=calls, avg_ms=ms / calls, words=w,
                           gbs=8.0 * w * npts_local / (ms / calls * 1e-3) / 1e9)
    return kern


def single_gpu_passes(capi, device, n, L, workload, steps, warmup, fields=None):
    """A few passes of another configuration on this GPU (resident fields, prologue outside
    the timed region), for the `also` entries of the line."""
    f = fields or slab_problem(n, 1, 0, workload)
    var = workload != "const"
    with capi.Solver(f["mx"], f["delx"], f["bc"], L=L, eps=0.0, max_outer=0, devices=(device,)) as s:
        def load():
            if var:
                s.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
            else:
                s.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])
        load()
        # the timed passes run without the per-kernel events (two event records per launch cost a
        # few microseconds of queue time each, 3-4 % of a 0.5 ms pass); the kernel table comes from
        # the same number of passes repeated with the events on
        s.set_profile(False)
        s.iterate(warmup)
        it0 = s.stats().iterations
        t0 = time.perf_counter()
        st = s.iterate(steps)
        dt = time.perf_counter() - t0
        ran = s.stats().iterations - it0
        hist = s.history()
        s.set_profile(True)
        load()                               # the same passes of the same solve once more
        s.iterate(warmup)
        a = s.stats()
        t0 = time.perf_counter()
        s.iterate(steps)
        dt_events = time.perf_counter() - t0
        b = s.stats()
    npts = float(n + 1) ** 3
    kern = kernel_table(a, b, L, var, npts)
    moved = moved_words(L, var)
    ms = dt / steps * 1e3
    return {"workload": "%d^3 fp64 %s-coefficient, l=%d, Dirichlet faces" % (n, "variable" if var else "constant", L),
            "value": round(steps / dt, 2), "unit": "iter/s", "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
            "ms_per_step_with_kernel_events": round(dt_events / steps * 1e3, 4),
            "valid": bool(st == 0 and ran == steps and np.all(np.isfinite(hist))),
            "frac_of_hbm_peak_contract_words": round(8.0 * contract_words(L, var) * npts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_of_hbm_peak_moved_words": round(8.0 * moved * npts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernels_measured_on": "the same %d passes of the same solve run again with HIP events around every launch" % steps,
            "kernels": {v["name"]: {"avg_ms": round(v["avg_ms"], 4), "GBps": round(v["gbs"], 1), "words": v["words"]}
                        for v in kern.values()}}


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
    ap.add_argument("--no-also", action="store_true", help="skip the extra single-GPU configurations")
    ap.add_argument("--settle", type=float, default=2.0,
                    help="seconds of idle GPU before each single-GPU configuration (outside every timed region)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the ranks rendezvous, and rank 0 prints who took part (launcher check)")
    ap.add_argument("--cpu-grid", dest="cpu_n", type=int, default=256,
                    help="grid of the extra, small CPU sample (the CPU legs proper run at the GPU line's own size)")
    ap.add_argument("--cpu-iters", type=int, default=10)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # called directly: become the launcher.  Nothing above loads the solver library or torch.
        return launch_ranks(sys.argv[1:], args.gpus)

    real_stdout = claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.dry_run:
        who = [(rank, local_rank, os.getpid())]
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
            got = [None] * world
            dist.all_gather_object(got, who[0])
            who = got
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            os.write(real_stdout, (json.dumps({"dry_run": True, "n_gpus": world, "ranks": who, "launched_by": "bench.py"
                                               if os.environ.get("KRY_BENCH_LAUNCHED") else "other"}) + "\n").encode())
        return 0
    # per-kernel HIP events only where the roofline is reported (one GPU)
    os.environ.setdefault("KRY_PROFILE", "1" if world == 1 else "0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: the control plane stays on loopback

    # the solver library first (it binds the ROCm installation's HIP runtime),
    # torch.distributed afterwards and for the control plane only
    from krylov_amd import capi
    capi.lib()
    ndev = capi.device_count()
    if ndev < 1:
        sys.exit("bench.py: no HIP device")
    # one GPU per rank; with fewer GPUs than ranks (a rehearsal on a smaller box) ranks share them
    device = int(os.environ.get("KRY_BENCH_DEVICE", local_rank % ndev))
    capi.device_check(device)

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)

    n, L, var = args.n, args.L, args.workload != "const"
    nz = 128 * world if args.weak else n
    if nz % world != 0:
        sys.exit("grid size %d is not divisible by %d slabs" % (nz, world))
    f = slab_problem(n, world, rank, args.workload, nz)

    def new_solver():
        return capi.Solver(f["mx"], f["delx"], f["bc"], L=L, eps=0.0, max_outer=0, nslabs=world,
                           slab_begin=rank, slab_count=1, devices=(device,))

    def all_ok(mine):
        flag = torch.tensor([1 if mine else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag[0]) == 1

    solver = new_solver()
    transport, rccl = "none", {"ranks": 0, "error": None}
    if world > 1:
        transport = "rccl"
        err = None
        try:
            uid = [capi.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            solver.comm_init(uid[0], rank, world)
        except Exception as e:           # no usable RCCL: keep the run alive on a host-staged path
            err = str(e)
            sys.stderr.write("rank %d: RCCL communicator unavailable (%s)\n" % (rank, e))
        errs = [None] * world
        dist.all_gather_object(errs, err)
        if any(errs):
            from krylov_amd import gloo_transport
            rccl["error"] = next(e for e in errs if e)
            transport = "gloo-staged (RCCL refused)"
            solver.close()
            solver = new_solver()
            solver.comm_init_callbacks(rank, world, *gloo_transport.make(rank, world))
        else:
            rccl["ranks"] = int(solver.stats().rccl_ranks)

    def barrier():
        if dist is not None:
            dist.barrier()

    def load():
        if var:
            solver.load_var(f["phi"], f["rhs"], f["lam"], f["Dxx"], f["Dyy"], f["Dzz"], None, False)
        else:
            solver.load_const(f["phi"], f["rhs"], f["lam"], f["Dc"])

    # Exchange between the ranks: through the transport (per dot product one RCCL all-reduce and
    # two small kernels, per operator output one grouped send/receive), with the dot products
    # all-reduced by the reduction kernels themselves through device mailboxes ("mailbox"), or
    # with the ghost planes travelling through mapped device memory as well ("mapped memory").
    # All are tried on a few passes of this very problem; the fastest one that reproduces the
    # transport's residuals is timed in full next to the transport (see run_route below).
    # KRY_BENCH_REDUCE=transport skips the trial, =mailbox / ="mapped memory" prefers that route
    # when it is valid.
    exchange, trial, why = "none", None, "one GPU: nothing to exchange"
    mail_connected = False
    if world > 1:
        exchange, why = "transport", "KRY_BENCH_REDUCE=transport"
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
            connected, cerr = False, None
            if all(h is not None for h in handles):
                try:
                    solver.mail_connect(rank, world, handles)
                    connected = True
                except Exception as e:
                    cerr = str(e)
                    sys.stderr.write("rank %d: mailboxes not connected (%s)\n" % (rank, e))
            else:
                cerr = "a rank could not export its mailbox"
            if not all_ok(connected):
                if connected:
                    solver.mail_enable(0)
                cerrs = [None] * world
                dist.all_gather_object(cerrs, cerr)
                why = "the ranks' GPUs cannot map each other's memory (%s)" % next((e for e in cerrs if e), "?")
            else:
                mail_connected = True
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
                exchange = best
                why = ("the trial (6 passes of this problem per route) found none faster that reproduces the transport's residuals"
                       if want == "auto" else "KRY_BENCH_REDUCE=%s" % want)
                trial = {k: (round(v, 4) if np.isfinite(v) else None) for k, v in trial.items()}
        barrier()

    restarts = [0]

    def run_passes(n):
        """n passes.  With eps = 0 the solver never converges, but on a hard problem the recurrences
        can break down once the residual sits on the rounding floor (alpha or the norm is no number:
        the reference exits there, solveLin_BICGStab.C:406-409, 491-494); the passes queued behind
        that point do nothing.  They must not count: the solve is then restarted from the current
        iterate (residual and recurrences anew, inside the timed region) for the passes still owed."""
        owed, st = n, 0
        while owed > 0:
            before = solver.stats().iterations
            st = solver.iterate(owed)             # returns after a stream synchronise
            owed -= solver.stats().iterations - before
            if owed > 0:
                if restarts[0] >= 8:
                    return st if st != 0 else -1
                restarts[0] += 1
                solver.reset()
        return 0

    def measure():
        """Stage, warm up, time K passes.  Returns None when this rank's run failed."""
        try:
            restarts[0] = 0
            ts = time.perf_counter()
            load()
            ts = time.perf_counter() - ts
            if args.warmup > 0:
                run_passes(args.warmup)
            a = solver.stats()
            barrier()
            t0 = time.perf_counter()
            stt = run_passes(args.steps)
            t1 = time.perf_counter()
            barrier()
            return ts, a, solver.stats(), stt, t1 - t0, solver.history()
        except capi.KryError as e:
            sys.stderr.write("rank %d: %s\n" % (rank, e))
            try:
                barrier()
                barrier()
            except Exception:
                pass
            return None

    def ranks_agree(m):
        """Every rank ran, finished, and holds the same residual history bit for bit (the sums are
        formed once and shared, or added in the same order everywhere): stale or lost exchange data
        on any rank would break that."""
        mine = None if m is None else (int(m[3]), m[5].tobytes())
        got = [None] * world
        dist.all_gather_object(got, mine)
        return all(g is not None and g[0] == 0 and g[1] == got[0][1] for g in got)

    def probe_passes():
        """Several GPUs: the timed region runs without per-kernel events (rank 0 would lag behind the
        others by their cost); the kernel table, the roofline and the transport's share of a pass
        come from a few extra passes with the events on.  Collective: every rank runs them or none."""
        try:
            solver.set_profile(True)
            pa = solver.stats()
            barrier()
            tp = time.perf_counter()
            solver.iterate(6)
            ms = (time.perf_counter() - tp) / 6 * 1e3
            pb = solver.stats()
            solver.set_profile(False)
            return 6, ms, pa, pb
        except Exception as e:
            sys.stderr.write("rank %d: no per-kernel times (%s)\n" % (rank, e))
            return 0, None, None, None

    def wait_shares(route, s0, s1, elapsed, probe):
        """Share of a pass spent exchanging (rank 0): mapped-memory routes over the timed region
        (the kernels count their own waiting), the transport over the probe passes (HIP events)."""
        n_probe, probe_ms, pa, pb = probe
        if n_probe and (pb.comm_wait_calls[0] or pb.comm_wait_calls[1]) and route != "mapped memory":
            per = n_probe * probe_ms
            ar = (pb.comm_wait_ms[0] - pa.comm_wait_ms[0]) / per
            hl = (pb.comm_wait_ms[1] - pa.comm_wait_ms[1]) / per
            basis = "HIP events around the transport's calls and the main stream's waits, %d probe passes" % n_probe
            if route == "mailbox":
                basis = "all-reduce: time counted by the reducing kernels; ghost planes: " + basis
        else:
            per = elapsed * 1e3
            ar = (s1.comm_wait_ms[0] - s0.comm_wait_ms[0]) / per
            hl = (s1.comm_wait_ms[1] - s0.comm_wait_ms[1]) / per
            basis = "time counted by the waiting kernels themselves over the timed region"
        return {"allreduce": round(ar, 4), "halo": round(hl, 4), "basis": basis}

    def run_route(route):
        """The whole timed run (stage, warm up, K passes between barriers, max over ranks) on one
        exchange route, then the probe passes.  Returns what the line reports about it."""
        if world > 1:
            if mail_connected:
                solver.mail_enable(modes[route])
            barrier()
        m = measure()
        agree = ranks_agree(m) if world > 1 else None
        if m is None or agree is False:
            return {"route": route, "valid": False, "ranks_hold_identical_history": agree, "m": None}
        t_stage, s0, s1, st, elapsed, hist = m
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt[0])
        ok = (st == 0 and (restarts[0] > 0 or s1.iterations == args.warmup + args.steps) and np.all(np.isfinite(hist))
              and agree is not False)
        r = {"route": route, "valid": bool(ok), "ms_per_step": round(elapsed / args.steps * 1e3, 4),
             "value": round(args.steps / elapsed, 3), "restarts": restarts[0], "ranks_hold_identical_history": agree,
             "m": (t_stage, s0, s1, st, elapsed, hist), "probe": (0, None, None, None)}
        if world > 1 and all_ok(ok):
            r["probe"] = probe_passes()
            r["wait_share_of_pass"] = wait_shares(route, s0, s1, elapsed, r["probe"])
        return r

    # One GPU: one run.  Several: the transport (RCCL send/receive and all-reduce -- the route the
    # solver is specified on) is always timed for the full K passes; when the trial found a valid
    # faster route through mapped device memory, that one is timed for the full K passes as well.
    # The headline is the RCCL run; when RCCL itself could not start (the transport is then the
    # host-staged stand-in, e.g. several ranks rehearsing on one GPU) it is the fastest valid run.
    modes = {"mailbox": capi.KRY_MAIL_SUMS, "mapped memory": capi.KRY_MAIL_SUMS | capi.KRY_MAIL_PLANES, "transport": 0}
    runs = {}
    def run_also():
        res = []
        for (an, aL, awl, asteps, awarm) in ((256, 1, "const", 400, 100), (256, 2, "const", 200, 50),
                                             (n, 1 if L == 2 else 2, args.workload, 20, 3)):
            time.sleep(args.settle)
            try:
                res.append(single_gpu_passes(capi, device, an, aL, awl, asteps, awarm,
                                             fields=f if (an == n and awl == args.workload) else None))
            except Exception as e:
                res.append({"workload": "%d^3 %s l=%d" % (an, awl, aL), "error": str(e)})
        return [res[2], res[0], res[1]]       # reported in the order the line has always had
    # The same driver run also times l = 1 on this grid and the 256^3 constant-coefficient problem.  Every
    # configuration starts after --settle seconds of idle GPU, the extra ones before the headline: the chip settles
    # about 1 % lower after a few seconds of sustained 512^3 passes, and a configuration timed right behind another
    # inherits that state (256^3 l = 1: 2060-2070 passes/s behind the 512^3 run, 2085-2090 from idle); the timed
    # region of each keeps its own warm-up passes.
    also = None
    if world == 1 and not args.weak and not args.no_also:
        also = run_also()
    if world == 1:
        time.sleep(args.settle)
    if world == 1:
        head = run_route("none")
    else:
        chosen_by_trial = exchange
        runs["transport"] = run_route("transport")
        if chosen_by_trial != "transport":
            runs[chosen_by_trial] = run_route(chosen_by_trial)
        name, why = choose_headline(runs, transport, rccl["error"], chosen_by_trial, why)
        head = runs[name]
        exchange = head["route"]
    if head.get("m") is None:
        # (a collective decision: the flags are the same on every rank)
        if rank == 0:
            os.write(real_stdout, (json.dumps({"metric": METRIC, "value": None, "unit": "iter/s", "n_gpus": world,
                                               "valid": False, "error": "the timed run failed on a rank, or the ranks' residual histories differ",
                                               "config": {"transport": transport, "exchange": exchange}}) + "\n").encode())
        return 1
    t_stage, s0, s1, st, elapsed, hist = head["m"]
    ok = head["valid"]
    agree = head["ranks_hold_identical_history"]
    restarts[0] = head["restarts"]
    probe, probe_ms, pa_, pb_ = head["probe"]
    sa, sb = (pa_, pb_) if probe else (s0, s1)

    if rank == 0:
        npts = float(n + 1) ** 2 * (nz + 1)
        kern = kernel_table(sa, sb, L, var, npts / world)
        dom = max(kern, key=lambda k: kern[k]["avg_ms"] * kern[k]["calls"]) if kern else None
        roof = None
        if dom is not None:
            roof = {"bound": "hbm", "kernel": kern[dom]["name"], "achieved": round(kern[dom]["gbs"], 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kern[dom]["gbs"] / HBM_PEAK_GBS, 4),
                    "traffic": None, "avg_launch_ms": round(kern[dom]["avg_ms"], 4),
                    "algorithmic_bytes_per_launch": 8.0 * kern[dom]["words"] * npts / world}
            if probe:
                roof["measured_on"] = "%d passes after the timed region, rank 0" % probe
            tr, src = measured_traffic(kern[dom]["name"], n, L, args.workload, world)
            if tr is not None:
                roof["traffic"] = tr
                roof["traffic_source"] = "profiles/" + src
        ms_step = elapsed / args.steps * 1e3
        W = contract_words(L, var)
        moved = moved_words(L, var)
        whole = 8.0 * W * npts / (ms_step * 1e-3) / 1e9
        whole_moved = 8.0 * moved * npts / (ms_step * 1e-3) / 1e9
        wait = head.get("wait_share_of_pass")
        routes = None
        if world > 1:
            routes = {name: {k: r.get(k) for k in ("valid", "ms_per_step", "value", "restarts", "wait_share_of_pass",
                                                   "ranks_hold_identical_history")} for name, r in runs.items()}
            routes["transport"]["kind"] = transport
        out = {
            "metric": METRIC,
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
                       "launched_by": "bench.py" if os.environ.get("KRY_BENCH_LAUNCHED") else
                                      ("torch.distributed.run" if world > 1 else "direct"),
                       "devices_visible": ndev, "ranks_per_device": -(-world // ndev),
                       "transport": transport, "rccl_ranks": rccl["ranks"], "rccl_error": rccl["error"],
                       "exchange": exchange, "exchange_reason": why,
                       "exchange_trial_ms_per_pass": trial, "wait_share_of_pass": wait,
                       "ranks_hold_identical_history": agree,
                       # every route's full timed run (K passes each); "rccl" repeats the transport's entry when
                       # the transport is RCCL
                       "routes": routes, "rccl": (routes["transport"] if routes and transport == "rccl" else None)},
            "valid": bool(ok), "solver_path": s1.fused, "restarts": restarts[0],
            "iteration_GBps": round(whole, 1),
            "iteration_frac_of_hbm_peak": round(whole / HBM_PEAK_GBS / world, 4),
            "iteration_words_per_point": W,
            "iteration_words_moved": moved,
            "iteration_GBps_moved": round(whole_moved, 1),
            "iteration_frac_of_hbm_peak_moved": round(whole_moved / HBM_PEAK_GBS / world, 4),
            "stage_s": round(t_stage, 3),
            "residual_after": float(hist[-1]) if len(hist) else None,
            "kernels": {v["name"]: {"avg_ms": round(v["avg_ms"], 4), "GBps": round(v["gbs"], 1),
                                    "words": v["words"], "calls": v["calls"]} for v in kern.values()},
            "roofline": roof,
        }
    solver.close()

    if rank == 0:
        if also is not None:
            out["also"] = also
        if world == 1 and not args.no_cpu and not args.weak:
            out["cpu_baseline"] = cpu_baseline(L, args.workload, n, args.cpu_n, args.cpu_iters, 8)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
