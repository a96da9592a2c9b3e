/* Plain C99 caller of the C ABI: test 1 of the reference's driver (BICGStab3D/main.C:223-229,
 * 243, 276-284) -- constant diffusion D = (1,1,10), lambda = 0.2 x y^2 z, Dirichlet faces on
 * [0,1]^3, rhs = -(12 pi^2 + lambda) sin(pi x) sin(pi y) sin(pi z) -- through the constant-
 * coefficient overload of the solver (BICGStab3D/solveLin_BICGStab.C:200-215), solved to 1e-6
 * and compared with the analytic solution sin(pi x) sin(pi y) sin(pi z).
 * (With lambda = 0 and D = (1,1,1) the right-hand side is an eigenvector of the discrete
 * operator: the first BiCG step is exact and BiCGStab(2) then breaks down with alpha = 0/0,
 * in the reference as here.)
 *
 *   gcc -std=c99 -O2 -Iinclude examples/solve_poisson.c -Lkrylov_amd -lkrylov_amd -lm \
 *       -Wl,-rpath,$PWD/krylov_amd -o examples/solve_poisson && examples/solve_poisson [N] [L]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "krylov_amd.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 64;      /* points 0..n per direction */
	const int L = argc > 2 ? atoi(argv[2]) : 2;
	const size_t w = (size_t)n + 3, total = w * w * w; /* ghost-inclusive box -1..n+1 */
	double *phi = calloc(total, sizeof *phi), *rhs = calloc(total, sizeof *rhs), *lam = calloc(total, sizeof *lam);
	kry_desc d;
	kry_stats st;
	kry_handle h;
	int status = 0, i, j, k;
	double err = 0.;

	if (!phi || !rhs || !lam || n < 8) return 2;
	for (k = 0; k <= n; ++k)
		for (j = 0; j <= n; ++j)
			for (i = 0; i <= n; ++i) {
				const double x = (double)i / n, y = (double)j / n, z = (double)k / n;
				const size_t p = ((size_t)(k + 1) * w + (j + 1)) * w + (i + 1);
				lam[p] = 0.2 * x * y * y * z;
				rhs[p] = -(12. * M_PI * M_PI + lam[p]) * sin(M_PI * x) * sin(M_PI * y) * sin(M_PI * z);
			}

	memset(&d, 0, sizeof d);
	for (i = 0; i < 3; ++i) { d.mx[i] = n; d.delx[i] = 1. / n; }
	d.L = L;                       /* all six bc_type entries 0: Dirichlet */
	d.eps = 1e-6;
	d.nslabs = d.slab_count = 1;   /* one GPU, device 0 */
	h = kry_create(&d, &status);
	if (!h) {
		fprintf(stderr, "kry_create: %s (%s)\n", kry_status_string(status), kry_last_error());
		return 1;
	}
	status = kry_solve_const(h, phi, rhs, lam, 1., 1., 10., 0);
	kry_get_stats(h, &st);
	kry_destroy(h);
	if (status != KRY_OK) {
		fprintf(stderr, "kry_solve_const: %s (%s)\n", kry_status_string(status), kry_last_error());
		return 1;
	}
	for (k = 0; k <= n; ++k)
		for (j = 0; j <= n; ++j)
			for (i = 0; i <= n; ++i) {
				const double x = (double)i / n, y = (double)j / n, z = (double)k / n;
				const double e = phi[((size_t)(k + 1) * w + (j + 1)) * w + (i + 1)] -
				                 sin(M_PI * x) * sin(M_PI * y) * sin(M_PI * z);
				err += e * e;
			}
	err = sqrt(err / ((double)(n + 1) * (n + 1) * (n + 1)));
	printf("n %d  l %d  passes %d  residual %.3e of |rhs| %.3e  l2 error %.3e  kernels %s\n", n, L, st.iterations,
	       st.norm_final, st.norm_rhs, err, st.fused ? "fused" : "generic");
	free(phi); free(rhs); free(lam);
	/* second-order discretisation: the error is about (pi h)^2 / 12 of the amplitude */
	return err < 2. / ((double)n * n) ? 0 : 1;
}
