"""Host logic of bench.py: slab fields agree with the undivided fields, the
per-kernel byte model adds up to no more than the contract figure."""
import numpy as np
import pytest

import bench


@pytest.mark.parametrize("workload", ["var", "const"])
@pytest.mark.parametrize("nslabs", [2, 4])
def test_slab_fields_tile_the_global_fields(workload, nslabs):
    n = 16
    whole = bench.slab_problem(n, 1, 0, workload)
    mzl = n // nslabs
    for s in range(nslabs):
        part = bench.slab_problem(n, nslabs, s, workload)
        assert part["mx"] == (n, n, mzl)
        for key in ("rhs", "lam") + (("Dxx", "Dyy", "Dzz") if workload == "var" else ()):
            # local plane k (array index k+1) is global plane s*mzl + k, ghosts included
            assert np.array_equal(part[key], whole[key][s * mzl: s * mzl + mzl + 3]), key


def test_direct_call_with_several_gpus_launches_its_own_ranks():
    """`python bench.py --gpus N` outside torch.distributed.run starts N fresh rank processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) and relays rank 0's
    one line; --dry-run keeps the ranks off the GPU."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, bench.__file__, "--gpus", "3", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["launched_by"] == "bench.py"
    assert sorted(r[0] for r in d["ranks"]) == [0, 1, 2] and len({r[2] for r in d["ranks"]}) == 3


def test_launcher_reports_a_failing_rank():
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["KRY_LIB_PATH"] = "/nonexistent/libkrylov_amd.so"      # the ranks cannot load the solver library
    out = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""


def test_contract_and_moved_words():
    for L in (1, 2):
        for var in (False, True):
            assert sum(bench.kernel_words(L, var, False).values()) <= bench.contract_words(L, var)


def test_kernel_byte_model_totals():
    # doubles per point and pass under the fully fused schedule: 20/26 (l=1), 50/62 (l=2)
    assert sum(bench.kernel_words(1, False, False).values()) == 19      # contract: 20
    assert sum(bench.kernel_words(1, True, False).values()) == 25       # contract: 26
    assert sum(bench.kernel_words(2, False, False).values()) <= 50
    assert sum(bench.kernel_words(2, True, False).values()) <= 62


def test_headline_is_the_rccl_run_unless_rccl_is_out():
    """Several GPUs: every route is timed in full; `value` reports the RCCL run whenever RCCL carried the
    transport, also when a mapped-memory route was faster; without RCCL the fastest valid run."""
    fast, slow = {"valid": True, "ms_per_step": 1.0}, {"valid": True, "ms_per_step": 2.0}
    name, why = bench.choose_headline({"transport": slow, "mapped memory": fast}, "rccl", None, "mapped memory", "")
    assert name == "transport" and "mapped memory" in why and "RCCL" in why
    name, why = bench.choose_headline({"transport": slow}, "rccl", None, "transport", "cannot map")
    assert name == "transport" and "cannot map" in why
    # RCCL refused (ranks sharing a GPU): host-staged stand-in, the fastest valid run leads and the line says why
    name, why = bench.choose_headline({"transport": slow, "mapped memory": fast}, "gloo-staged (RCCL refused)",
                                      "2 ranks on one device", "mapped memory", "")
    assert name == "mapped memory" and "2 ranks on one device" in why
    # RCCL's own run failed: the other route, with the reason
    name, why = bench.choose_headline({"transport": {"valid": False}, "mailbox": slow}, "rccl", None, "mailbox", "")
    assert name == "mailbox" and "failed" in why
    name, _ = bench.choose_headline({"transport": {"valid": False}}, "rccl", None, "transport", "")
    assert name == "transport"
