/*
 * krylov_oracle_ranks.c -- TEST INFRASTRUCTURE ONLY (see krylov_oracle.h).
 *
 * CPU restatement of the reference's `-D parallel` build of the BiCGStab(l)
 * path: the domain is split into px*py*pz sub-domains that share their
 * interface planes (mpi_manager_3D::make_LocalGrid, BICGStab3D/mpi_manager.C:
 * 342-408; the driver hard-codes 2 x 2 x 2, BICGStab3D/main.C:146-150), every
 * sub-domain runs solve_int on its local arrays, ghost planes travel through
 * BoundaryHandler3D::do_bc_MPI (BICGStab3D/BoundaryHandler.C:579-1045) and every
 * dot product / norm ends in an MPI_Allreduce (BICGStab3D/Krylov.C:61-62,149-150,
 * 239-240).  MPI is not used here: each rank is a thread, a message is a copy
 * through a per-rank buffer between two barriers, an all-reduce is a sum over a
 * shared array in rank order.  It is the multi-core CPU baseline timed by
 * bench.py next to the GPU, and is checked against the serial restatement in
 * tests/test_oracle_ranks.py.
 *
 * What the parallel build does differently from the serial one, all kept:
 *   - solve_int first exchanges the ghosts of the coefficient fields and applies
 *     do_BCs to rhs (solveLin_BICGStab.C:259-267);
 *   - do_BCs treats only faces of the global domain and runs the exchange of a
 *     direction right after that direction's faces (BoundaryHandler.C:463-465,
 *     520-522, 569-571);
 *   - the weights 1/2, 1/4, 1/8 (1/4 at corners in the norm) apply at the faces of
 *     every sub-domain, interfaces included (Krylov.C:81-143, 180-233);
 *   - as a consequence of the order "x faces, x exchange, y faces, ..." and of the y-face
 *     loops stopping at ix = mx (BoundaryHandler.C:479-480), the edge ghosts (mx+1, 0, k) and
 *     (mx+1, my, k) next to an x interface hold untreated neighbour values; only the
 *     off-diagonal term reads them, so with Dxy != 0 an x split changes the operator.
 *
 * PARITY UNPINNED, like the serial restatement.
 */
#include "krylov_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	int mx, my, mz;
	long sx, sy, n;
} lbox;
#define AT(b, a, i, j, k) ((a)[((long)(k) + 1) * (b).sy + ((long)(j) + 1) * (b).sx + ((i) + 1)])

struct shared;
typedef struct {
	struct shared *sh;
	int rank, co[3];
	int nbr[6];                 /* rank behind face xlo,xhi,ylo,yhi,zlo,zhi, -1: face of the global domain */
	orc_cfg c;                  /* local geometry; bc[f] = -1 on interfaces */
	lbox b;
	double *phi, *rhs, *lam, *Dxx, *Dyy, *Dzz, *Dxy;
	double *buf;                /* outgoing face */
} rank_t;

struct shared {
	int P, np[3];
	orc_cfg g;                  /* global problem */
	pthread_barrier_t bar;
	double *red;                /* P values of an all-reduce */
	rank_t *rk;
	double *hist;
	int hist_cap;
	orc_out out;
};

static double sqr(double v) { return v * v; }

/* ---- MPI stand-ins ------------------------------------------------------- */
static void barrier(rank_t *r) { pthread_barrier_wait(&r->sh->bar); }

static double allreduce(rank_t *r, double v, int take_max)
{
	struct shared *s = r->sh;
	s->red[r->rank] = v;
	barrier(r);
	double t = s->red[0];
	for (int p = 1; p < s->P; ++p) t = take_max ? (s->red[p] > t ? s->red[p] : t) : t + s->red[p];
	barrier(r);
	return t;
}

/* one plane of the box (index `idx` along `dir`) over the full tangential extent, ghosts included */
static void plane_copy(const lbox *b, double *d, int dir, int idx, double *buf, int to_buf)
{
	const int m[3] = {b->mx, b->my, b->mz};
	const int a = (dir == 0) ? 1 : 0, c = (dir == 2) ? 1 : 2;   /* the two tangential directions, a faster */
	long t = 0;
	for (int ic = -1; ic <= m[c] + 1; ++ic)
		for (int ia = -1; ia <= m[a] + 1; ++ia, ++t) {
			int ijk[3];
			ijk[dir] = idx; ijk[a] = ia; ijk[c] = ic;
			double *p = &AT(*b, d, ijk[0], ijk[1], ijk[2]);
			if (to_buf) buf[t] = *p; else *p = buf[t];
		}
}

/* BoundaryHandler3D::do_bc_MPI for one direction, rim = 1 (BoundaryHandler.C:579-1045):
 * plane 1 goes to the lower neighbour's ghost m+1, then plane m-1 to the upper neighbour's ghost -1 */
static void exchange(rank_t *r, double *d, int dir)
{
	const int m = (dir == 0) ? r->b.mx : (dir == 1) ? r->b.my : r->b.mz;
	const int lo = r->nbr[2 * dir], hi = r->nbr[2 * dir + 1];
	if (lo >= 0) plane_copy(&r->b, d, dir, 1, r->buf, 1);
	barrier(r);
	if (hi >= 0) plane_copy(&r->b, d, dir, m + 1, r->sh->rk[hi].buf, 0);
	barrier(r);
	if (hi >= 0) plane_copy(&r->b, d, dir, m - 1, r->buf, 1);
	barrier(r);
	if (lo >= 0) plane_copy(&r->b, d, dir, -1, r->sh->rk[lo].buf, 0);
	barrier(r);
}

/* BoundaryHandler3D::do_BCs, parallel build (BoundaryHandler.C:370-575) */
static void do_bcs(rank_t *r, double *d, int keep)
{
	const lbox b = r->b;
	const int mx = b.mx, my = b.my, mz = b.mz;
	const int *bc = r->c.bc;     /* -1 on interfaces: nothing happens there */
	if (!keep) {
		if (bc[0] == 0) { for (int iy = -1; iy <= my + 1; ++iy) for (int iz = -1; iz <= mz + 1; ++iz) for (int ip = -1; ip <= 0; ++ip) AT(b, d, ip, iy, iz) = 0.; }
		else if (bc[0] == 1) { for (int iy = -1; iy <= my + 1; ++iy) for (int iz = -1; iz <= mz + 1; ++iz) AT(b, d, -1, iy, iz) = 2. * AT(b, d, 0, iy, iz) - AT(b, d, 1, iy, iz); }
		if (bc[1] == 0) { for (int iy = -1; iy <= my + 1; ++iy) for (int iz = -1; iz <= mz + 1; ++iz) for (int ip = mx; ip <= mx + 1; ++ip) AT(b, d, ip, iy, iz) = 0.; }
		else if (bc[1] == 1) { for (int iy = -1; iy <= my + 1; ++iy) for (int iz = -1; iz <= mz + 1; ++iz) for (int ip = mx; ip <= mx + 1; ++ip) AT(b, d, ip, iy, iz) = 2. * AT(b, d, ip - 1, iy, iz) - AT(b, d, ip - 2, iy, iz); }
	}
	exchange(r, d, 0);
	if (!keep) {
		if (bc[2] == 0) { for (int ix = 0; ix <= mx; ++ix) for (int iz = -1; iz <= mz + 1; ++iz) for (int ip = -1; ip <= 0; ++ip) AT(b, d, ix, ip, iz) = 0.; }
		else if (bc[2] == 1) { for (int ix = 0; ix <= mx; ++ix) for (int iz = -1; iz <= mz + 1; ++iz) AT(b, d, ix, -1, iz) = 2. * AT(b, d, ix, 0, iz) - AT(b, d, ix, 1, iz); }
		if (bc[3] == 0) { for (int ix = 0; ix <= mx; ++ix) for (int iz = -1; iz <= mz + 1; ++iz) for (int ip = my; ip <= my + 1; ++ip) AT(b, d, ix, ip, iz) = 0.; }
		else if (bc[3] == 1) { for (int ix = 0; ix <= mx; ++ix) for (int iz = -1; iz <= mz + 1; ++iz) for (int ip = my; ip <= my + 1; ++ip) AT(b, d, ix, ip, iz) = 2. * AT(b, d, ix, ip - 1, iz) - AT(b, d, ix, ip - 2, iz); }
	}
	exchange(r, d, 1);
	if (!keep) {
		if (bc[4] == 0) { for (int iy = 0; iy <= my; ++iy) for (int ix = 0; ix <= mx; ++ix) for (int ip = -1; ip <= 0; ++ip) AT(b, d, ix, iy, ip) = 0.; }
		else if (bc[4] == 1) { for (int iy = 0; iy <= my; ++iy) for (int ix = 0; ix <= mx; ++ix) for (int ip = -1; ip <= 0; ++ip) AT(b, d, ix, iy, ip) = 2. * AT(b, d, ix, iy, ip + 1) - AT(b, d, ix, iy, ip + 2); }
		if (bc[5] == 0) { for (int iy = 0; iy <= my; ++iy) for (int ix = 0; ix <= mx; ++ix) for (int ip = mz; ip <= mz + 1; ++ip) AT(b, d, ix, iy, ip) = 0.; }
		else if (bc[5] == 1) { for (int iy = 0; iy <= my; ++iy) for (int ix = 0; ix <= mx; ++ix) for (int ip = mz; ip <= mz + 1; ++ip) AT(b, d, ix, iy, ip) = 2. * AT(b, d, ix, iy, ip - 1) - AT(b, d, ix, iy, ip - 2); }
	}
	exchange(r, d, 2);
}

/* Krylov::dot_product (Krylov.C:161-244): the local weighted sum is the serial one on the
 * local box, then the all-reduce */
static double dot(rank_t *r, const double *a, const double *b)
{
	return allreduce(r, orc_dot(&r->c, a, b), 0);
}

/* Krylov::get_l2Norm (Krylov.C:36-157) with the two all-reduces */
static double l2norm(rank_t *r, const double *v)
{
	const lbox b = r->b;
	const int mx = b.mx, my = b.my, mz = b.mz;
	volatile double sink = 0.;
	{
		double m = 0.;                                     /* vec.max_norm(), value discarded (:47, 57) */
		for (long i = 0; i < b.n; ++i) { double a = fabs(v[i]); if (a > m) m = a; }
		sink = m;
	}
	(void)sink;
	double mymax = -10;
	for (int iz = 0; iz <= mz; iz++)
		for (int iy = 0; iy <= my; iy++)
			for (int ix = 0; ix <= mx; ix++) { double a = fabs(AT(b, v, ix, iy, iz)); mymax = (mymax < a) ? a : mymax; }
	const double valMax = allreduce(r, mymax, 1);
	if (valMax == 0.) return 0;
	const double sc = 1. / valMax;
	double sum = 0.;
	for (int iz = 1; iz < mz; iz++) for (int iy = 1; iy < my; iy++) for (int ix = 1; ix < mx; ix++) sum += sqr(sc * AT(b, v, ix, iy, iz));
	for (int iz = 1; iz < mz; iz++) for (int iy = 1; iy < my; iy++) { sum += 0.5 * sqr(sc * AT(b, v, 0, iy, iz)); sum += 0.5 * sqr(sc * AT(b, v, mx, iy, iz)); }
	for (int iz = 1; iz < mz; iz++) for (int ix = 1; ix < mx; ix++) { sum += 0.5 * sqr(sc * AT(b, v, ix, 0, iz)); sum += 0.5 * sqr(sc * AT(b, v, ix, my, iz)); }
	for (int iy = 1; iy < my; iy++) for (int ix = 1; ix < mx; ix++) { sum += 0.5 * sqr(sc * AT(b, v, ix, iy, 0)); sum += 0.5 * sqr(sc * AT(b, v, ix, iy, mz)); }
	for (int iz = 1; iz < mz; ++iz) {
		sum += 0.25 * sqr(sc * AT(b, v, 0, 0, iz)); sum += 0.25 * sqr(sc * AT(b, v, 0, my, iz));
		sum += 0.25 * sqr(sc * AT(b, v, mx, 0, iz)); sum += 0.25 * sqr(sc * AT(b, v, mx, my, iz));
	}
	for (int iy = 1; iy < my; ++iy) {
		sum += 0.25 * sqr(sc * AT(b, v, 0, iy, 0)); sum += 0.25 * sqr(sc * AT(b, v, 0, iy, mz));
		sum += 0.25 * sqr(sc * AT(b, v, mx, iy, 0)); sum += 0.25 * sqr(sc * AT(b, v, mx, iy, mz));
	}
	for (int ix = 1; ix < mx; ++ix) {
		sum += 0.25 * sqr(sc * AT(b, v, ix, 0, 0)); sum += 0.25 * sqr(sc * AT(b, v, ix, 0, mz));
		sum += 0.25 * sqr(sc * AT(b, v, ix, my, 0)); sum += 0.25 * sqr(sc * AT(b, v, ix, my, mz));
	}
	sum += 0.25 * sqr(sc * AT(b, v, 0, 0, 0)); sum += 0.25 * sqr(sc * AT(b, v, 0, my, 0));
	sum += 0.25 * sqr(sc * AT(b, v, 0, 0, mz)); sum += 0.25 * sqr(sc * AT(b, v, 0, my, mz));
	sum += 0.25 * sqr(sc * AT(b, v, mx, 0, 0)); sum += 0.25 * sqr(sc * AT(b, v, mx, my, 0));
	sum += 0.25 * sqr(sc * AT(b, v, mx, 0, mz)); sum += 0.25 * sqr(sc * AT(b, v, mx, my, mz));
	return valMax * sqrt(allreduce(r, sum, 0));
}

/* Krylov::multiply_withMat, both forms (Krylov.C:407-541) */
static void spmv(rank_t *r, double *psi, double *out, int apply_bcs)
{
	const lbox b = r->b;
	if (r->c.var_coef) {
		orc_spmv_var(&r->c, psi, r->lam, r->Dxx, r->Dyy, r->Dzz, r->Dxy, out, 0);
	} else {
		double coeff[3];
		for (int d = 0; d < 3; ++d) coeff[d] = r->c.Dc[d] / sqr(r->c.delx[d]);
		do_bcs(r, psi, 0);                                 /* :516 */
		for (int iz = 0; iz <= b.mz; iz++)
			for (int iy = 0; iy <= b.my; iy++)
				for (int ix = 0; ix <= b.mx; ix++)
					AT(b, out, ix, iy, iz) =
					    (coeff[0] * (AT(b, psi, ix + 1, iy, iz) + AT(b, psi, ix - 1, iy, iz)) +
					     coeff[1] * (AT(b, psi, ix, iy + 1, iz) + AT(b, psi, ix, iy - 1, iz)) +
					     coeff[2] * (AT(b, psi, ix, iy, iz + 1) + AT(b, psi, ix, iy, iz - 1)) -
					     (2. * (coeff[0] + coeff[1] + coeff[2]) + AT(b, r->lam, ix, iy, iz)) * AT(b, psi, ix, iy, iz));
	}
	if (apply_bcs) do_bcs(r, out, 0);
}

/* whole-array operators of NumMatrix, un-fused as in the serial restatement */
static void arr_scale(long n, double *a, double s) { for (long i = 0; i < n; ++i) a[i] *= s; }
static void arr_add(long n, double *a, const double *b) { for (long i = 0; i < n; ++i) a[i] += b[i]; }
static void arr_axpy(long n, double *a, const double *b, double s, int sign)
{
	double *t = (double *)malloc((size_t)n * sizeof(double));
	for (long i = 0; i < n; ++i) t[i] = b[i] * s;
	if (sign > 0) for (long i = 0; i < n; ++i) a[i] += t[i];
	else for (long i = 0; i < n; ++i) a[i] -= t[i];
	free(t);
}

static void residual(rank_t *r, double *res)
{
	const long n = r->b.n;
	spmv(r, r->phi, res, 0);
	double *t = (double *)malloc((size_t)n * sizeof(double));
	for (long i = 0; i < n; ++i) t[i] = r->rhs[i] - res[i];
	memcpy(res, t, (size_t)n * sizeof(double));
	free(t);
	do_bcs(r, res, 0);
}

/* BICGStab::solve_int, parallel build (solveLin_BICGStab.C:250-527), on one rank */
static void *rank_main(void *arg)
{
	rank_t *rk = (rank_t *)arg;
	struct shared *sh = rk->sh;
	const long n = rk->b.n;
	const int L = rk->c.L;
	const double eps = rk->c.eps;
	orc_out o;
	memset(&o, 0, sizeof o);

	/* :259-267 (the constant-coefficient overload passes rhs for all four fields, :214) */
	if (rk->c.var_coef) { do_bcs(rk, rk->Dxx, 1); do_bcs(rk, rk->Dyy, 1); do_bcs(rk, rk->Dzz, 1); do_bcs(rk, rk->Dxy, 1); }
	else for (int k = 0; k < 4; ++k) do_bcs(rk, rk->rhs, 1);
	do_bcs(rk, rk->rhs, 0);
	double normRHS = l2norm(rk, rk->rhs);
	if (normRHS == 0.) normRHS = 1.;
	o.norm_rhs = normRHS;

	double **r = (double **)malloc((size_t)(L + 1) * sizeof(double *));
	double **u = (double **)malloc((size_t)(L + 1) * sizeof(double *));
	for (int k = 0; k <= L; ++k) { r[k] = (double *)calloc((size_t)n, sizeof(double)); u[k] = (double *)calloc((size_t)n, sizeof(double)); }
	double *rt = (double *)calloc((size_t)n, sizeof(double));
	int iter_steps = 0;
	double norm = 1.e99;

	residual(rk, r[0]);
	const double norm_init = l2norm(rk, r[0]);
	o.norm_init = norm_init;
	if (norm_init < eps * normRHS) { o.norm_final = norm_init; o.norm_true = norm_init; goto done; }
	memcpy(rt, r[0], (size_t)n * sizeof(double));
	{
		double rho0 = 1., rho1, alpha = 0., omega = 1.;
		double tau[L + 1][L + 1], sigma[L + 1], gammap[L + 1], gammapp[L + 1], gamma[L + 1];
		do {
			iter_steps++;
			rho0 *= -omega;
			for (int jj = 0; jj < L; ++jj) {
				rho1 = dot(rk, r[jj], rt);
				double beta = alpha * rho1 / rho0;
				rho0 = rho1;
				for (int ii = 0; ii <= jj; ++ii) { arr_scale(n, u[ii], -beta); arr_add(n, u[ii], r[ii]); }
				spmv(rk, u[jj], u[jj + 1], 1);
				alpha = rho0 / dot(rk, u[jj + 1], rt);
				for (int ii = 0; ii <= jj; ++ii) arr_axpy(n, r[ii], u[ii + 1], alpha, -1);
				spmv(rk, r[jj], r[jj + 1], 1);
				arr_axpy(n, rk->phi, u[0], alpha, +1);
				if (orc_really_is_nan(alpha)) { o.status = 1; goto stop; }
			}
			for (int jj = 1; jj <= L; ++jj) {
				for (int ii = 1; ii < jj; ++ii) {
					tau[ii][jj] = dot(rk, r[jj], r[ii]) / sigma[ii];
					arr_axpy(n, r[jj], r[ii], tau[ii][jj], -1);
				}
				sigma[jj] = dot(rk, r[jj], r[jj]);
				gammap[jj] = dot(rk, r[0], r[jj]) / sigma[jj];
			}
			omega = gamma[L] = gammap[L];
			for (int jj = L - 1; jj >= 1; --jj) {
				gamma[jj] = gammap[jj];
				for (int ii = jj + 1; ii <= L; ++ii) gamma[jj] -= tau[jj][ii] * gamma[ii];
			}
			for (int jj = 1; jj < L; ++jj) {
				gammapp[jj] = gamma[jj + 1];
				for (int ii = jj + 1; ii < L; ++ii) gammapp[jj] += tau[jj][ii] * gamma[ii + 1];
			}
			arr_axpy(n, rk->phi, r[0], gamma[1], +1);
			arr_axpy(n, r[0], r[L], gammap[L], -1);
			arr_axpy(n, u[0], u[L], gamma[L], -1);
			for (int jj = 1; jj < L; ++jj) {
				arr_axpy(n, u[0], u[jj], gamma[jj], -1);
				arr_axpy(n, rk->phi, r[jj], gammapp[jj], +1);
				arr_axpy(n, r[0], r[jj], gammap[jj], -1);
			}
			norm = l2norm(rk, r[0]);
			if (rk->rank == 0 && sh->hist && iter_steps <= sh->hist_cap) sh->hist[iter_steps - 1] = norm;
			if (orc_really_is_nan(norm)) { o.status = 3; goto stop; }
			if (rk->c.max_outer > 0 && iter_steps >= rk->c.max_outer && norm > eps * normRHS) { o.status = 4; break; }
		} while (norm > eps * normRHS);
	}
	residual(rk, r[0]);                                    /* :512-518 */
	o.norm_true = l2norm(rk, r[0]);
stop:
	o.norm_final = norm;
done:
	o.iters = iter_steps;
	for (int k = 0; k <= L; ++k) { free(r[k]); free(u[k]); }
	free(r); free(u); free(rt);
	if (rk->rank == 0) sh->out = o;
	return NULL;
}

/* local <-> global boxes */
static void scatter(const orc_cfg *g, const rank_t *rk, const double *G, double *loc)
{
	lbox gb;
	gb.mx = g->mx[0]; gb.my = g->mx[1]; gb.mz = g->mx[2];
	gb.sx = gb.mx + 3; gb.sy = gb.sx * (long)(gb.my + 3);
	const lbox b = rk->b;
	const int ox = rk->co[0] * b.mx, oy = rk->co[1] * b.my, oz = rk->co[2] * b.mz;
	for (int k = -1; k <= b.mz + 1; ++k)
		for (int j = -1; j <= b.my + 1; ++j)
			for (int i = -1; i <= b.mx + 1; ++i) AT(b, loc, i, j, k) = AT(gb, G, ox + i, oy + j, oz + k);
}

static void gather(const orc_cfg *g, const rank_t *rk, const double *loc, double *G)
{
	lbox gb;
	gb.mx = g->mx[0]; gb.my = g->mx[1]; gb.mz = g->mx[2];
	gb.sx = gb.mx + 3; gb.sy = gb.sx * (long)(gb.my + 3);
	const lbox b = rk->b;
	const int ox = rk->co[0] * b.mx, oy = rk->co[1] * b.my, oz = rk->co[2] * b.mz;
	/* points 0..m of the sub-domain, plus the ghost layer where it is the global one */
	const int i0 = rk->nbr[0] < 0 ? -1 : 0, i1 = rk->nbr[1] < 0 ? b.mx + 1 : b.mx;
	const int j0 = rk->nbr[2] < 0 ? -1 : 0, j1 = rk->nbr[3] < 0 ? b.my + 1 : b.my;
	const int k0 = rk->nbr[4] < 0 ? -1 : 0, k1 = rk->nbr[5] < 0 ? b.mz + 1 : b.mz;
	for (int k = k0; k <= k1; ++k)
		for (int j = j0; j <= j1; ++j)
			for (int i = i0; i <= i1; ++i) AT(gb, G, ox + i, oy + j, oz + k) = AT(b, loc, i, j, k);
}

/* The whole solve on np[0]*np[1]*np[2] ranks.  Fields are the GLOBAL ghost-inclusive arrays;
 * phi returns the assembled solution.  Rank of sub-domain (cx,cy,cz): (cx*np[1] + cy)*np[2] + cz
 * (MPI_Cart_create without reordering, mpi_manager.C:112-151). */
int orc_solve_ranks(const orc_cfg *g, const int np[3], double *phi, const double *rhs, const double *lambda,
                    const double *Dxx, const double *Dyy, const double *Dzz, const double *Dxy,
                    double *hist, int hist_cap, orc_out *out)
{
	for (int d = 0; d < 3; ++d)
		if (np[d] < 1 || g->mx[d] % np[d] != 0 || g->mx[d] / np[d] < 2) return -1;
	struct shared sh;
	memset(&sh, 0, sizeof sh);
	sh.P = np[0] * np[1] * np[2];
	for (int d = 0; d < 3; ++d) sh.np[d] = np[d];
	sh.g = *g;
	sh.hist = hist;
	sh.hist_cap = hist_cap;
	sh.red = (double *)calloc((size_t)sh.P, sizeof(double));
	sh.rk = (rank_t *)calloc((size_t)sh.P, sizeof(rank_t));
	pthread_barrier_init(&sh.bar, NULL, (unsigned)sh.P);
	for (int p = 0; p < sh.P; ++p) {
		rank_t *r = &sh.rk[p];
		r->sh = &sh;
		r->rank = p;
		r->co[0] = p / (np[1] * np[2]); r->co[1] = (p / np[2]) % np[1]; r->co[2] = p % np[2];
		r->c = *g;
		for (int d = 0; d < 3; ++d) {
			r->c.mx[d] = g->mx[d] / np[d];
			int cc[3] = {r->co[0], r->co[1], r->co[2]};
			cc[d] = r->co[d] - 1;
			r->nbr[2 * d] = r->co[d] > 0 ? (cc[0] * np[1] + cc[1]) * np[2] + cc[2] : -1;
			cc[d] = r->co[d] + 1;
			r->nbr[2 * d + 1] = r->co[d] < np[d] - 1 ? (cc[0] * np[1] + cc[1]) * np[2] + cc[2] : -1;
			if (r->nbr[2 * d] >= 0) r->c.bc[2 * d] = -1;
			if (r->nbr[2 * d + 1] >= 0) r->c.bc[2 * d + 1] = -1;
		}
		lbox *b = &r->b;
		b->mx = r->c.mx[0]; b->my = r->c.mx[1]; b->mz = r->c.mx[2];
		b->sx = b->mx + 3; b->sy = b->sx * (long)(b->my + 3); b->n = b->sy * (long)(b->mz + 3);
		double **f[7] = {&r->phi, &r->rhs, &r->lam, &r->Dxx, &r->Dyy, &r->Dzz, &r->Dxy};
		const double *src[7] = {phi, rhs, lambda, Dxx, Dyy, Dzz, Dxy};
		for (int k = 0; k < 7; ++k) {
			if (k >= 3 && !g->var_coef) { *f[k] = NULL; continue; }
			*f[k] = (double *)calloc((size_t)b->n, sizeof(double));
			/* the field overload clears Dxy unless it is used (solveLin_BICGStab.C:231-233) */
			if (src[k] && !(k == 6 && !g->use_offdiag)) scatter(g, r, src[k], *f[k]);
		}
		long face = (long)(b->mx + 3) * (b->my + 3);
		if ((long)(b->mx + 3) * (b->mz + 3) > face) face = (long)(b->mx + 3) * (b->mz + 3);
		if ((long)(b->my + 3) * (b->mz + 3) > face) face = (long)(b->my + 3) * (b->mz + 3);
		r->buf = (double *)malloc((size_t)face * sizeof(double));
	}
	pthread_t *th = (pthread_t *)malloc((size_t)sh.P * sizeof(pthread_t));
	for (int p = 0; p < sh.P; ++p) pthread_create(&th[p], NULL, rank_main, &sh.rk[p]);
	for (int p = 0; p < sh.P; ++p) pthread_join(th[p], NULL);
	for (int p = 0; p < sh.P; ++p) {
		rank_t *r = &sh.rk[p];
		gather(g, r, r->phi, phi);
		free(r->phi); free(r->rhs); free(r->lam); free(r->Dxx); free(r->Dyy); free(r->Dzz); free(r->Dxy); free(r->buf);
	}
	pthread_barrier_destroy(&sh.bar);
	if (out) *out = sh.out;
	const int st = sh.out.status;
	free(th); free(sh.red); free(sh.rk);
	return st;
}
