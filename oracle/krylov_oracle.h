/*
 * krylov_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, single thread) of the reference's BiCGStab(l)
 * stencil solve:  BICGStab3D/solveLin_BICGStab.C, BICGStab3D/Krylov.C and
 * BoundaryHandler3D::do_BCs of BICGStab3D/BoundaryHandler.C (serial build,
 * i.e. compiled without -D parallel).
 *
 * It exists to CHECK the HIP path.  Only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py may load it.  Nothing in krylov_amd/ links,
 * imports or calls it.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and
 * its sources cannot be built here (the array library matrix.H they depend on
 * is not part of the reference tree).  The restatement follows the cited
 * lines operation by operation and is compiled with -ffp-contract=off so that
 * no fused multiply-add changes a rounding; it is sanity-anchored by the
 * analytic solutions of the reference driver (BICGStab3D/main.C:276-396).
 *
 * Array layout: every field is a ghost-inclusive box [-1..mx+1]^3, x fastest:
 *   index(i,j,k) = ((k+1)*(my+3) + (j+1))*(mx+3) + (i+1).
 */
#ifndef KRYLOV_ORACLE_H
#define KRYLOV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int mx[3];        /* last grid-point index per direction (points 0..mx) */
	double delx[3];   /* grid spacing */
	int bc[6];        /* xlo,xhi,ylo,yhi,zlo,zhi: 0 Dirichlet, 1 "Neumann"
	                     (linear extrapolation), anything else: untouched */
	int L;            /* BiCGStab(l) polynomial degree */
	double eps;       /* relative tolerance */
	int max_outer;    /* 0 = unlimited (reference behaviour) */
	int var_coef;     /* 0: constant-coefficient overload; 1: field overload */
	int use_offdiag;  /* field overload: 0 clears Dxy first (as the reference) */
	double Dc[3];     /* constant D_xx, D_yy, D_zz */
	int debug;        /* >2 prints the reference's per-iteration lines */
} orc_cfg;

typedef struct {
	int iters;            /* outer iterations executed */
	int status;           /* 0 ok, 1 alpha nan (ref exit(1)), 3 norm nan (ref exit(3)),
	                         4 stopped by max_outer */
	double norm_rhs;      /* normRHS after the zero -> one substitution */
	double norm_init;     /* l2 of the first residual */
	double norm_final;    /* last value of `norm` in the loop */
	double norm_true;     /* l2 of rhs - A phi recomputed after the loop */
} orc_out;

/* Whole solve.  phi is in/out; rhs/lambda/D* are inputs but may be modified
 * exactly where the reference modifies them (Dxy cleared, do_BCs on phi via the
 * constant-coefficient operator).  hist receives `norm` of each outer
 * iteration (up to hist_cap entries).  Returns out->status. */
int orc_solve(const orc_cfg *cfg, double *phi, double *rhs, double *lambda,
              double *Dxx, double *Dyy, double *Dzz, double *Dxy,
              double *hist, int hist_cap, orc_out *out);

/* krylov_oracle_ranks.c: the same solve as the reference's `-D parallel` build runs it, on
 * np[0]*np[1]*np[2] sub-domains (one thread per MPI rank; BICGStab3D/mpi_manager.C:56-408,
 * BICGStab3D/BoundaryHandler.C:579-1045).  Fields are the GLOBAL arrays, phi returns the
 * assembled solution; every mx[d] must be a multiple of np[d].  Returns out->status, -1 for a
 * split that does not divide the grid.  No reduction hook: sums run in the reference's order
 * per rank and in rank order across ranks. */
int orc_solve_ranks(const orc_cfg *cfg, const int np[3], double *phi, const double *rhs,
                    const double *lambda, const double *Dxx, const double *Dyy, const double *Dzz,
                    const double *Dxy, double *hist, int hist_cap, orc_out *out);

/* Reduction call sites of orc_solve, for orc_set_reduce_hook. */
enum {
	ORC_SITE_NORM_RHS = 0,  /* l2(rhs)                     solveLin_BICGStab.C:268 */
	ORC_SITE_NORM_INIT,     /* l2(first residual)          :301 */
	ORC_SITE_RHO,           /* (r[jj], r~)                 :349 */
	ORC_SITE_ALPHA,         /* (u[jj+1], r~)               :390 */
	ORC_SITE_TAU,           /* (r[jj], r[ii])              :422 */
	ORC_SITE_SIGMA,         /* (r[jj], r[jj])              :427 */
	ORC_SITE_GAMMAP,        /* (r[0], r[jj])               :428 */
	ORC_SITE_NORM           /* l2(r[0]) at the end of a pass :464 */
};

/* A hook replaces the summation of every reduction in orc_solve (norm sites must
 * return the norm itself, not its square).  iter counts passes from 1; jj is the
 * loop index of the site.  NULL restores the reference's summation order. */
typedef double (*orc_reduce_hook)(void *ctx, int site, int iter, int jj, const orc_cfg *cfg,
                                  const double *a, const double *b);
void orc_set_reduce_hook(orc_reduce_hook hook, void *ctx);

/* The reference's cross-check handle (solveLin_BICGStab.C:13-47): XOR of the raw bits of every
 * stored cell of a field, started from 797003437.  With a trace buffer set, orc_solve records
 * hash(phi), hash(rhs), hash(lambda) on entry (the reference's disabled prints at :284-287) and
 * hash(phi), hash(residuals[0]) at the end of every pass (:473-477): what a build of the reference
 * with those prints enabled must show.  orc_hash_trace_count: values recorded so far (may exceed cap). */
unsigned long long orc_hash(const orc_cfg *cfg, const double *a);
void orc_set_hash_trace(unsigned long long *buf, int cap);
int  orc_hash_trace_count(void);

/* gpu_order.c: repeat the sums in the order of the HIP kernels (fused tile rows
 * `ty`, planes per chunk `zc`, streaming-kernel grid target `stream_blocks`);
 * enable = 0 switches back to the reference order. */
void orc_use_gpu_order(int enable, int ty, int zc, int stream_blocks);
/* The device run used `nslabs` z slabs (per-slab sums added in slab order);
 * split_launch: its stencil kernels ran in two parts, edge chunks first (fused
 * Dirichlet path; the general-boundary path never splits).  Defaults: 1, 0. */
void orc_gpu_order_slabs(int nslabs, int split_launch);

/* Building blocks, exported for kernel-level parity tests. */
void   orc_do_bcs(const orc_cfg *cfg, double *a);
void   orc_spmv_const(const orc_cfg *cfg, double *psi, const double *lambda,
                      double *out, int apply_bcs);
void   orc_spmv_var(const orc_cfg *cfg, const double *psi, const double *lambda,
                    const double *Dxx, const double *Dyy, const double *Dzz,
                    const double *Dxy, double *out, int apply_bcs);
double orc_dot(const orc_cfg *cfg, const double *a, const double *b);
double orc_l2norm(const orc_cfg *cfg, const double *v);
int    orc_really_is_nan(double x);
long   orc_store_size(const orc_cfg *cfg);

#ifdef __cplusplus
}
#endif
#endif
