"""Synthetic stencil problems: grids, right-hand sides, coefficient fields.

Every field is a ghost-inclusive box with index range [-1 .. m+1] per
direction, stored as a C-ordered numpy array of shape (mz+3, my+3, mx+3)
(x fastest), which is the host layout the C ABI takes.

The manufactured problems follow the reference driver (BICGStab3D/main.C:
236-430): phi_ana = sin(pi x) sin(pi y) sin(pi z) on [0,1]^3 with
x_i = xb + delx*i (BICGStab3D/grid_manager.C:306-311), evaluated on the grid
points 0..m.  The benchmark configurations additionally define the ghost
cells from the same closed forms, because Neumann faces read them.
"""
from dataclasses import dataclass, field

import numpy as np

PI = np.pi


@dataclass
class Problem:
    name: str
    mx: tuple                      # (mx, my, mz): last grid-point index
    delx: tuple
    bc: tuple                      # xlo,xhi,ylo,yhi,zlo,zhi: 0 Dirichlet, 1 Neumann
    var_coef: bool
    use_offdiag: bool
    Dc: tuple
    phi: np.ndarray
    rhs: np.ndarray
    lam: np.ndarray
    Dxx: np.ndarray = None
    Dyy: np.ndarray = None
    Dzz: np.ndarray = None
    Dxy: np.ndarray = None
    phi_ana: np.ndarray = None
    meta: dict = field(default_factory=dict)

    @property
    def n_pts(self):
        return (self.mx[0] + 1) * (self.mx[1] + 1) * (self.mx[2] + 1)


def coords(mx, delx, xb=(0., 0., 0.)):
    """Positions of indices -1..m+1 per direction: xb + delx*i."""
    ax = [xb[d] + delx[d] * np.arange(-1, mx[d] + 2, dtype=np.float64) for d in range(3)]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    return x, y, z


def box(mx):
    return np.zeros((mx[2] + 3, mx[1] + 3, mx[0] + 3), dtype=np.float64)


def interior_only(a):
    """Keep values on points 0..m, zero the ghost layer (the reference driver
    fills 0..m only, BICGStab3D/main.C:251-253)."""
    out = np.zeros_like(a)
    out[1:-1, 1:-1, 1:-1] = a[1:-1, 1:-1, 1:-1]
    return out


def _grid(n):
    n3 = (n, n, n) if np.isscalar(n) else tuple(n)
    delx = tuple(1.0 / m for m in n3)          # grid_manager.C:372-373 with Len = 1
    return n3, delx


def poisson(n, name="poisson"):
    """Config 1/2: constant D=(1,1,1), lambda = 0, Dirichlet,
    rhs = -3 pi^2 sin sin sin."""
    mx, delx = _grid(n)
    x, y, z = coords(mx, delx)
    ana = np.sin(PI * x) * np.sin(PI * y) * np.sin(PI * z)
    rhs = interior_only(-3.0 * PI * PI * ana)
    return Problem(name, mx, delx, (0,) * 6, False, False, (1., 1., 1.),
                   box(mx), rhs, box(mx), phi_ana=interior_only(ana))


def reference_test1(n, lambda_factor=0.2, Dc=(1., 1., 10.)):
    """The reference driver's test 1 (BICGStab3D/main.C:223-229,277,282-284)."""
    mx, delx = _grid(n)
    x, y, z = coords(mx, delx)
    ana = np.sin(PI * x) * np.sin(PI * y) * np.sin(PI * z)
    lam = lambda_factor * x * (y * y) * z
    rhs = -((PI * PI) * (Dc[0] + Dc[1] + Dc[2]) + lam) * ana
    return Problem("ref_test1", mx, delx, (0,) * 6, False, False, tuple(Dc),
                   box(mx), interior_only(rhs), interior_only(lam),
                   phi_ana=interior_only(ana))


def reference_test2(n, lambda_factor=0.2):
    """The reference driver's test 2: Dxx=y, Dyy=x, Dzz=z (main.C:299-311).
    Ghost cells of the D fields keep their initial 1 (main.C:231-234)."""
    mx, delx = _grid(n)
    x, y, z = coords(mx, delx)
    ana = np.sin(PI * x) * np.sin(PI * y) * np.sin(PI * z)
    lam = lambda_factor * x * (y * y) * z
    rhs = (-((PI * PI) * (x + y + z) + lam) * ana
           + PI * np.sin(PI * x) * np.sin(PI * y) * np.cos(PI * z))

    def dfield(v):
        out = np.ones_like(v)
        out[1:-1, 1:-1, 1:-1] = v[1:-1, 1:-1, 1:-1]
        return out
    return Problem("ref_test2", mx, delx, (0,) * 6, True, False, (1., 1., 1.),
                   box(mx), interior_only(rhs), interior_only(lam),
                   dfield(y), dfield(x), dfield(z), box(mx), phi_ana=interior_only(ana))


def reference_test3(n, lambda_factor=0.2):
    """The reference driver's test 3: as test 2 plus Dxy = A x^2 y z,
    off-diagonal diffusion on (main.C:344-357,479-480)."""
    p = reference_test2(n, lambda_factor)
    x, y, z = coords(p.mx, p.delx)
    A = 0.1
    dxy = A * (x * x) * y * z
    lam = lambda_factor * x * (y * y) * z
    sx, sy, sz = np.sin(PI * x), np.sin(PI * y), np.sin(PI * z)
    cx, cy, cz = np.cos(PI * x), np.cos(PI * y), np.cos(PI * z)
    rhs = (-((PI * PI) * (x + y + z) + lam) * (sx * sy * sz)
           + PI * sx * sy * cz
           + 2. * dxy * (PI * PI) * cx * cy * sz
           + 2. * A * x * y * z * PI * sx * cy * sz
           + A * (x * x) * z * PI * cx * sy * sz)
    p.name = "ref_test3"
    p.use_offdiag = True
    p.Dxy = interior_only(dxy)
    p.rhs = interior_only(rhs)
    return p


def smooth_var(n, name="smooth_var", bc=(0,) * 6):
    """Config 3: smooth variable coefficients, Dxy = 0, lambda = 0.2 x y^2 z,
    rhs = sin sin sin; every field defined on ghosts too."""
    mx, delx = _grid(n)
    x, y, z = coords(mx, delx)
    Dxx = 1.0 + 0.5 * np.sin(2 * PI * x) * np.cos(2 * PI * y)
    Dyy = 1.0 + 0.5 * np.sin(2 * PI * y) * np.cos(2 * PI * z)
    Dzz = 1.0 + 0.5 * np.sin(2 * PI * z) * np.cos(2 * PI * x)
    lam = 0.2 * x * (y * y) * z
    rhs = np.sin(PI * x) * np.sin(PI * y) * np.sin(PI * z)
    return Problem(name, mx, delx, tuple(bc), True, False, (1., 1., 1.),
                   box(mx), rhs, lam, Dxx, Dyy, Dzz, box(mx))


def high_contrast(n, name="high_contrast", bc=(0, 0, 1, 1, 0, 1)):
    """Config 5: contrast 1e4, anisotropy 100, mixed faces (x Dirichlet,
    y Neumann, z-low Dirichlet, z-high Neumann), lambda = 1e-3."""
    mx, delx = _grid(n)
    x, y, z = coords(mx, delx)
    s = np.sin(2 * PI * x) * np.sin(2 * PI * y) * np.sin(2 * PI * z)
    base = np.power(10.0, 2.0 * s)
    lam = np.full_like(x, 1e-3)
    rhs = np.sin(PI * x) * np.sin(PI * y) * np.sin(PI * z)
    return Problem(name, mx, delx, tuple(bc), True, False, (1., 1., 1.),
                   box(mx), rhs, lam, base, 0.1 * base, 10.0 * base, box(mx))


def offdiag_var(n, name="offdiag_var", bc=(0,) * 6):
    """Variable coefficients with a non-zero Dxy (11-point stencil path),
    modelled on the reference driver's test 3 coefficient shapes."""
    mx, delx = _grid(n)
    x, y, z = coords(mx, delx)
    Dxx = 1.0 + y
    Dyy = 1.0 + x
    Dzz = 1.0 + z
    Dxy = 0.1 * (x * x) * y * z
    lam = 0.2 * x * (y * y) * z
    rhs = np.sin(PI * x) * np.sin(PI * y) * np.sin(PI * z)
    return Problem(name, mx, delx, tuple(bc), True, True, (1., 1., 1.),
                   box(mx), rhs, lam, Dxx, Dyy, Dzz, Dxy)


def random_fields(mx, seed, bc=(0,) * 6, var_coef=True, use_offdiag=True, Dc=(1., 2., 3.),
                  delx=None):
    """Randomised fields for differential tests (the pattern of the reference's
    BiCGStab3D_CL/CL_Matrix.cpp:474-651 device-vs-host tests)."""
    rng = np.random.default_rng(seed)
    mx = tuple(mx)
    if delx is None:
        delx = tuple(1.0 / m for m in mx)
    shp = (mx[2] + 3, mx[1] + 3, mx[0] + 3)
    u = lambda lo, hi: rng.uniform(lo, hi, size=shp)
    return Problem("random", mx, tuple(delx), tuple(bc), var_coef, use_offdiag, tuple(Dc),
                   u(-1, 1), u(-1, 1), u(0, 2), u(0.5, 2), u(0.5, 2), u(0.5, 2),
                   u(-0.3, 0.3) if use_offdiag else box(mx))
