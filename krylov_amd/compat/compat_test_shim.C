// C entry points over the host-side drop-in classes so that the Python tests can
// exercise them (NumMatrix layout, grid geometry, BoundaryHandler3D::do_BCs on
// host arrays, the BICGStab adaptor).
#include <cstring>

#include "BiCGStabCL.hpp"
#include "BoundaryHandler.H"
#include "grid_manager.H"
#include "solveLin_BICGStab.H"

namespace {
void wrap(NumMatrix<double, 3> &m, const int mx[3], const double *src)
{
	m.resize(Index::set(-1, -1, -1), Index::set(mx[0] + 1, mx[1] + 1, mx[2] + 1));
	if (src) memcpy(m.data(), src, m.size() * sizeof(double));
}
} // namespace

extern "C" {

// fills `a` (ghost-inclusive box, x fastest) through NumMatrix::operator()
void compat_fill_ijk(const int mx[3], double *a)
{
	NumMatrix<double, 3> m;
	wrap(m, mx, NULL);
	for (int k = -1; k <= mx[2] + 1; ++k)
		for (int j = -1; j <= mx[1] + 1; ++j)
			for (int i = -1; i <= mx[0] + 1; ++i) m(i, j, k) = i + 100. * j + 10000. * k;
	memcpy(a, m.data(), m.size() * sizeof(double));
}

void compat_do_bcs(const int mx[3], const int bc[6], double *a)
{
	NumMatrix<double, 3> m;
	wrap(m, mx, a);
	BoundaryHandler3D b;
	for (int f = 0; f < 6; ++f) b.set_bcType(f, bc[f]);
	b.do_BCs(m, 1);
	memcpy(a, m.data(), m.size() * sizeof(double));
}

// geometry numbers that enter the operator
void compat_grid(int N, double xb, double xe, double *delx_manager, double *del_1d, double *pos /* N+2 */)
{
	grid_manager g(xb, xb, xb, xe, xe, xe, N, N, N, 1);
	grid_1D g1(xb, xe, N, 2, 0, true);
	*delx_manager = g.get_delx(0);
	*del_1d = g1.get_del();
	for (int i = -1; i <= N; ++i) pos[i + 1] = g.get_Pos(0, i);
}

// full solve through the Linsolver3D interface; returns the pass count
int compat_solve(const int mx[3], const int bc[6], int L, double eps, int max_iter, int var, int offd, const double Dc[3],
                 double *phi, const double *rhs, const double *lam, const double *Dxx, const double *Dyy, const double *Dzz,
                 const double *Dxy, double *hist, int hist_cap, int *fused)
{
	grid_1D gx(0, 1, mx[0] + 1, 2, 0, true), gy(0, 1, mx[1] + 1, 2, 0, true), gz(0, 1, mx[2] + 1, 2, 0, true);
	BoundaryHandler3D b;
	for (int f = 0; f < 6; ++f) b.set_bcType(f, bc[f]);
	NumMatrix<double, 3> mphi, mrhs, mlam, mDxx, mDyy, mDzz, mDxy;
	wrap(mphi, mx, phi); wrap(mrhs, mx, rhs); wrap(mlam, mx, lam);
	BICGStab s;
	Linsolver3D *ls = &s;
	NumArray<int> pars(1);
	pars(0) = L;
	ls->setup(gx, gy, gz, eps, pars, false, false, max_iter);
	if (var) {
		wrap(mDxx, mx, Dxx); wrap(mDyy, mx, Dyy); wrap(mDzz, mx, Dzz); wrap(mDxy, mx, Dxy);
		ls->solve(b, mphi, mrhs, mlam, mDxx, mDyy, mDzz, mDxy, 0, offd != 0);
	} else {
		ls->solve(b, mphi, mrhs, mlam, Dc[0], Dc[1], Dc[2], 0);
	}
	memcpy(phi, mphi.data(), mphi.size() * sizeof(double));
	std::vector<double> h = s.residualHistory();
	for (int i = 0; i < (int)h.size() && i < hist_cap; ++i) hist[i] = h[i];
	*fused = s.usedFusedKernels() ? 1 : 0;
	return s.iterations();
}

// the three-field overload (BICGStab3D/solveLin_BICGStab.C:238-248); Dzz is copied back so
// that the caller sees what the solver did to it; quirk != 0: the reference's behaviour
int compat_solve3(const int mx[3], int L, double eps, double *phi, const double *rhs, const double *lam, const double *Dxx,
                  const double *Dyy, double *Dzz, int quirk)
{
	grid_1D gx(0, 1, mx[0] + 1, 2, 0, true), gy(0, 1, mx[1] + 1, 2, 0, true), gz(0, 1, mx[2] + 1, 2, 0, true);
	BoundaryHandler3D b;
	b.set_bcType(0);
	NumMatrix<double, 3> mphi, mrhs, mlam, mDxx, mDyy, mDzz;
	wrap(mphi, mx, phi); wrap(mrhs, mx, rhs); wrap(mlam, mx, lam);
	wrap(mDxx, mx, Dxx); wrap(mDyy, mx, Dyy); wrap(mDzz, mx, Dzz);
	BICGStab s;
	s.set_ReferenceThreeFieldQuirk(quirk != 0);
	Linsolver3D *ls = &s;
	NumArray<int> pars(1);
	pars(0) = L;
	ls->setup(gx, gy, gz, eps, pars, true, false, 0);
	ls->solve(b, mphi, mrhs, mlam, mDxx, mDyy, mDzz, 0);
	memcpy(phi, mphi.data(), mphi.size() * sizeof(double));
	memcpy(Dzz, mDzz.data(), mDzz.size() * sizeof(double));
	return s.iterations();
}

// `steps` solves in a row through one BICGStab object, the right-hand side scaled by (1 + step) each
// time and phi restarted from `phi0` -- the shape of a time-stepping caller.  keep != 0: coefficients
// stay on the device between the solves.  out_iters / out_staged: per step.
int compat_time_steps(const int mx[3], int L, double eps, int steps, int keep, const double *phi0, double *phi,
                      const double *rhs0, const double *lam, const double *Dxx, const double *Dyy, const double *Dzz,
                      int *out_iters, int *out_staged)
{
	grid_1D gx(0, 1, mx[0] + 1, 2, 0, true), gy(0, 1, mx[1] + 1, 2, 0, true), gz(0, 1, mx[2] + 1, 2, 0, true);
	BoundaryHandler3D b;
	b.set_bcType(0);
	NumMatrix<double, 3> mphi, mrhs, mlam, mDxx, mDyy, mDzz, mDxy;
	wrap(mphi, mx, phi0); wrap(mrhs, mx, rhs0); wrap(mlam, mx, lam);
	wrap(mDxx, mx, Dxx); wrap(mDyy, mx, Dyy); wrap(mDzz, mx, Dzz); wrap(mDxy, mx, NULL);
	BICGStab s;
	s.set_KeepCoefficients(keep != 0);
	Linsolver3D *ls = &s;
	NumArray<int> pars(1);
	pars(0) = L;
	ls->setup(gx, gy, gz, eps, pars, true, false, 0);
	for (int t = 0; t < steps; ++t) {
		memcpy(mphi.data(), phi0, mphi.size() * sizeof(double));
		for (size_t i = 0; i < mrhs.size(); ++i) mrhs.data()[i] = rhs0[i] * (1. + t);
		ls->solve(b, mphi, mrhs, mlam, mDxx, mDyy, mDzz, mDxy, 0, false, 0.1, true);
		out_iters[t] = s.iterations();
		out_staged[t] = s.stagedFields();
		memcpy(phi + (size_t)t * mphi.size(), mphi.data(), mphi.size() * sizeof(double));
	}
	return 0;
}

// Two solves through one BICGStab object with the coefficients kept on the device, the face types
// changing in between (bc_a, then bc_b) and every array the same: the second solve runs on a new device
// handle and must stage everything again.  phi_b / staged[1]: what the second solve returned / staged.
int compat_keep_across_face_change(const int mx[3], int L, double eps, const int bc_a[6], const int bc_b[6],
                                   const double *phi0, double *phi_b, const double *rhs, const double *lam,
                                   const double *Dxx, const double *Dyy, const double *Dzz, int staged[2])
{
	grid_1D gx(0, 1, mx[0] + 1, 2, 0, true), gy(0, 1, mx[1] + 1, 2, 0, true), gz(0, 1, mx[2] + 1, 2, 0, true);
	NumMatrix<double, 3> mphi, mrhs, mlam, mDxx, mDyy, mDzz, mDxy;
	wrap(mphi, mx, phi0); wrap(mrhs, mx, rhs); wrap(mlam, mx, lam);
	wrap(mDxx, mx, Dxx); wrap(mDyy, mx, Dyy); wrap(mDzz, mx, Dzz); wrap(mDxy, mx, NULL);
	BICGStab s;
	s.set_KeepCoefficients(true);
	Linsolver3D *ls = &s;
	NumArray<int> pars(1);
	pars(0) = L;
	ls->setup(gx, gy, gz, eps, pars, true, false, 0);
	for (int t = 0; t < 2; ++t) {
		BoundaryHandler3D b;
		for (int f = 0; f < 6; ++f) b.set_bcType(f, t == 0 ? bc_a[f] : bc_b[f]);
		memcpy(mphi.data(), phi0, mphi.size() * sizeof(double));
		ls->solve(b, mphi, mrhs, mlam, mDxx, mDyy, mDzz, mDxy, 0, false);
		staged[t] = s.stagedFields();
	}
	memcpy(phi_b, mphi.data(), mphi.size() * sizeof(double));
	return s.iterations();
}

// the construction style of the reference's OpenCL solver (BiCGStab3D_CL/main.cpp:410-412, 532-540):
// constant-coefficient solve on Dirichlet faces; out = {iterations, steptimeMin, steptimeMax, number of step times,
// sum of the unrounded step times in ms}
int compat_solve_cl_style(int m, int L, double eps, const double Dc[3], double *phi, const double *rhs, const double *lam,
                          double out[5])
{
	const int mx[3] = {m, m, m};
	grid_manager grid(0., 0., 0., 1., 1., 1., m + 1, m + 1, m + 1, 1);
	BoundaryHandler3D b;
	b.set_bcType(0);
	NumMatrix<double, 3> mphi, mrhs, mlam;
	wrap(mphi, mx, phi); wrap(mrhs, mx, rhs); wrap(mlam, mx, lam);
	BiCGStabSolver solver(grid, eps, L, 0);
	solver.setVerbose(false);
	if (solver.isInitialized()) return -1;
	solver.setupContext();
	if (!solver.isInitialized()) return -2;
	solver.solve(b, mphi, mrhs, mlam, Dc[0], Dc[1], Dc[2], 0);
	memcpy(phi, mphi.data(), mphi.size() * sizeof(double));
	std::vector<long> t = solver.stepRuntimes();
	std::vector<double> tm = solver.stepRuntimesMs();
	out[0] = (double)solver.iterations();
	out[1] = (double)solver.steptimeMin();
	out[2] = (double)solver.steptimeMax();
	out[3] = (double)t.size();
	out[4] = 0.;
	for (size_t i = 0; i < tm.size(); ++i) out[4] += tm[i];
	solver.cleanupContext();
	return solver.isInitialized() ? -3 : 0;
}

}
