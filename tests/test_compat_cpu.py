"""Host-side drop-in classes (krylov_amd/compat): array layout, grid geometry
and BoundaryHandler3D::do_BCs on host arrays, checked without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from krylov_amd import problems as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "krylov_amd", "libkrylov_compat_test.so")
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(LIB):
        pytest.fail("libkrylov_compat_test.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB)
    i3, i6 = C.c_int * 3, C.c_int * 6
    L.compat_fill_ijk.argtypes = [i3, _dp]
    L.compat_do_bcs.argtypes = [i3, i6, _dp]
    L.compat_grid.argtypes = [C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), _dp]
    return L


def test_nummatrix_layout_is_x_fastest_ghost_inclusive(shim):
    mx = (5, 4, 3)
    a = np.zeros((mx[2] + 3, mx[1] + 3, mx[0] + 3))
    shim.compat_fill_ijk((C.c_int * 3)(*mx), a)
    k, j, i = np.meshgrid(np.arange(-1, mx[2] + 2), np.arange(-1, mx[1] + 2), np.arange(-1, mx[0] + 2), indexing="ij")
    assert np.array_equal(a, i + 100.0 * j + 10000.0 * k)


@pytest.mark.parametrize("bc", [(0,) * 6, (1,) * 6, (0, 0, 1, 1, 0, 1), (1, 0, 0, 1, 1, 0), (2, 0, 1, 2, 0, 1)])
@pytest.mark.parametrize("mx", [(6, 5, 4), (9, 9, 9)])
def test_host_do_bcs_equals_oracle(shim, orc, mx, bc):
    p = P.random_fields(mx, seed=31, bc=bc)
    want = p.phi.copy()
    orc.do_bcs(orc.make_cfg(p.mx, p.delx, bc), want)
    got = p.phi.copy()
    shim.compat_do_bcs((C.c_int * 3)(*mx), (C.c_int * 6)(*bc), got)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("N", [17, 33, 65, 101])
def test_grid_numbers(shim, N):
    """delx = Len/mx (grid_manager.C:372-373); grid_1D::get_del() is the width of
    cell 0 and equals Len/(Nx-1) exactly when xb = 0 (grid_manager.C:665-667,
    715-718, 758-761); positions are xb + delx*i."""
    d1, d2 = C.c_double(0), C.c_double(0)
    pos = np.zeros(N + 2)
    shim.compat_grid(N, 0.0, 1.0, C.byref(d1), C.byref(d2), pos)
    assert d1.value == 1.0 / (N - 1)
    assert d2.value == (0.5 * d1.value) - (-0.5 * d1.value)
    if (N - 1) & (N - 2) == 0:
        assert d2.value == d1.value
    assert np.array_equal(pos, 0.0 + d1.value * np.arange(-1, N + 1))


MPIEXEC = "/opt/conda/bin/mpiexec"
MPI_SELFTEST = os.path.join(ROOT, "krylov_amd", "driver", "compat_mpi_selftest")


@pytest.mark.parametrize("nproc", [(1, 1, 2), (1, 1, 4), (2, 2, 2)])
def test_parallel_flavour_topology_and_plane_exchange(nproc):
    """The `-D parallel` classes (mpi_manager_3D, BoundaryHandler3D with a process
    topology) under a real MPI: rank layout, local grids, outer-face flags and the
    ghost-plane exchange of do_BCs.  Host only."""
    import subprocess
    if not (os.path.exists(MPIEXEC) and os.path.exists(MPI_SELFTEST)):
        pytest.skip("no MPI installation / parallel flavour not built")
    n = nproc[0] * nproc[1] * nproc[2]
    out = subprocess.run([MPIEXEC, "-n", str(n), MPI_SELFTEST] + [str(v) for v in nproc], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "selftest ok" in out.stdout
