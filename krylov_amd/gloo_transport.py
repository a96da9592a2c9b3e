"""Host-staged inter-process transport for the C ABI's callback hooks, built on
torch.distributed (gloo).  Used by the multi-process tests, and by bench.py as
a fallback when an RCCL communicator cannot be created; production runs use
RCCL (kry_comm_init).  Import the solver library before torch so that the ROCm
installation's HIP runtime is the one mapped."""
import ctypes as C

_D2H, _H2D = 2, 1


def make(rank, world):
    """Returns (allreduce, halo) callables for Solver.comm_init_callbacks."""
    import torch
    import torch.distributed as dist
    # hipMemcpy of the very HIP runtime the solver library is linked to (a Python
    # package may have mapped a second copy of the runtime under another name):
    # symbols looked up on the library's handle resolve through its dependencies
    from krylov_amd import capi
    hip = capi.lib()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int

    def to_host(dev, n):
        t = torch.empty(n, dtype=torch.float64)
        if hip.hipMemcpy(t.data_ptr(), dev, n * 8, _D2H) != 0:
            raise RuntimeError("hipMemcpy device->host failed")
        return t

    def to_dev(dev, t):
        if hip.hipMemcpy(dev, t.data_ptr(), t.numel() * 8, _H2D) != 0:
            raise RuntimeError("hipMemcpy host->device failed")

    def allreduce(buf, n):
        t = to_host(buf, n)
        parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, t)
        acc = parts[0].clone()          # added in rank order on every rank
        for q in parts[1:]:
            acc += q
        to_dev(buf, acc)

    def halo(plo, slo, rlo, phi, shi, rhi, count):
        reqs, got = [], []
        if plo >= 0:
            s, r = to_host(slo, count), torch.empty(count, dtype=torch.float64)
            reqs += [dist.isend(s, plo), dist.irecv(r, plo)]
            got.append((rlo, r))
        if phi >= 0:
            s, r = to_host(shi, count), torch.empty(count, dtype=torch.float64)
            reqs += [dist.isend(s, phi), dist.irecv(r, phi)]
            got.append((rhi, r))
        for q in reqs:
            q.wait()
        for dev, r in got:
            to_dev(dev, r)

    return allreduce, halo
