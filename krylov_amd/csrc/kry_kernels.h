// Kernel argument blocks and launch wrappers (implemented in kry_kernels.hip).
#pragma once

#include "kry_device.h"

#define KRY_TXW 128   // fused tile width in grid points
#define KRY_WDOT_BLOCKS 4096   // largest grid of the generic weighted dot
#ifndef KRY_SEG
#define KRY_SEG 2048  // doubles of a plane handled per block and plane by the streaming kernels
#endif
#ifndef KRY_SURF_BLOCKS
#define KRY_SURF_BLOCKS 512    // largest grid of the boundary-plane dot kernel
#endif

// fused stencil kernel kinds
enum KryFusedKind {
	KF_A0 = 0,   // [l = 1: u0 -= gamma_1*u_1, deferred from the last pass] ; u0 = r0 - beta*u0 ; u1 = A u0 ; (u1, r~)
	KF_A1,       // l=2, jj=1: u1 = r1 - beta*u1 ; u2 = A u1 ; (u2, r~)   (phi and u0 of this step: k_final)
	KF_B0,       // l=2, jj=0: r0 -= alpha*u1 ; r1 = A r0 ; (r1, r~)
	KF_B0L,      // l=1:       r0 -= alpha*u1 ; r1 = A r0 ; (r0,r1), (r1,r1)
	KF_B1L,      // l=2, jj=1: r1 -= alpha*u2 ; r2 = A r1 ; (r1,r1), (r2,r1)
	KF_RES,      // r0 = r~ = rhs - A phi ; |r0|^2
	KF_SPMV,     // out = A psi
	KF_NKINDS
};

// grid reduction bookkeeping
struct KryRed {
	double *partial;     // nblk * ND doubles
	unsigned *counter;   // ticket counter, zero between launches
	KryState *S;
	int nblk;
	int phase;           // KryPhase run by the last block when S->local_scalars
	double *out;         // S == nullptr: the sums go here instead of S->sums (no recurrences, no mailboxes)
	int out_sqrt;        //   ... as their square roots
};

struct KryFused {
	KryGeo g;
	KryOp op;
	KryState *S;
	const double *X, *Y;     // psi = X + Y*c  (c = -beta for A kinds, -alpha for B kinds)
	const double *Z1, *Z2;   // KF_A0: Y is first updated, Y -= gamma_L*Z1 [; Y -= gamma_1*Z2]
	int nz;                  // how many of Z1, Z2 are in use (1 for l = 1, 0 for l = 2)
	double *Xout;            // where psi is stored (ping-pong partner of X's role)
	double *O, *O2;          // stencil output (O2: second copy, KF_RES only)
	const double *lam, *Dxx, *Dyy, *Dzz, *Dxy;
	const double *RT, *RHS;
	KryRed red;
	int zlo, zhi;            // planes on which the stencil is evaluated
	int zc;                  // planes per chunk
	int ntx, nty, nchunk;
	int nblk;                // blocks of THIS launch
	int ch_sel;              // 0: all chunks, 1: chunk 0 and the last n_edge-1 chunks, 2: the chunks in between,
	                         // 3: the tiles inside the box ib (they read no cell an exchange fills), 4: all others
	int ib[6];               // ch_sel 3, 4: tx0, tx1, ty0, ty1, ch0, ch1 (inclusive) of the inner tiles
	int n_edge;              // 2 or 3
	int blk_off;             // offset of this launch's blocks in the partial-sum array
	int rev;                 // 1: the blocks take the tiles in reverse order (every kernel starts where the one
	                         // before it ended, on data the memory-side cache still holds)
	int y_off;               // general boundaries: tile row 0 starts at grid row -y_off (1: the tiles cover the rows
	                         // -1 .. my+1; 0: they cover 0 .. 8*nty-1 and kry_launch_bcs_rows updates the rest)
};

// pointwise statements of BiCG step jj > 0 of the general-l pass: the current copies of r_0..r_jj-1 and u_0..u_jj
struct KryPointN {
	int jj;
	double *r[KRY_LMAX];
	double *u[KRY_LMAX + 1];
};

struct KryStream {
	KryGeo g;
	KryState *S;
	const double *v[8];      // read-only operands
	double *w[4];            // in/out operands
	KryRed red;
	int zfirst;              // first plane processed
	int nplanes;             // planes processed, strided over blockIdx.y
	int cnt_lo, cnt_hi;      // planes that count in the dot products
	int gen;                 // 1: general boundaries (weighted dots over the points 0..m)
	int rev;                 // 1: the blocks take the plane segments in reverse order
};

hipError_t kry_launch_fused(int kind, int ty, const KryFused &a, bool var, bool offd, bool exact,
                            hipStream_t st);
// inbc: constant coefficients with a face that is not Dirichlet -- the operator's input passes through do_BCs
hipError_t kry_launch_fused_gen(int kind, const KryFused &a, bool var, bool offd, bool exact, bool inbc, hipStream_t st);
// ... and then this goes first: the treated input on and behind the faces, into a.Xout (the operands' ghost cells
// must be current; the kernel launched with inbc reads them back)
hipError_t kry_launch_psi_shell(int kind, const KryFused &a, hipStream_t st);
int kry_fused_blocks_per_cu(int ty, bool var, bool offd, bool exact, bool gen);
hipError_t kry_launch_surf(int kind, const KryFused &a, int nblk, hipStream_t st);
hipError_t kry_launch_mgs2(const KryStream &a, int nbx, int nby, hipStream_t st);
hipError_t kry_launch_final(int L, const KryStream &a, int nbx, int nby, hipStream_t st);
// general l (l > 2): pointwise updates, one Gram-Schmidt step, end of pass
//   w[0] = phi
hipError_t kry_launch_bicgn(const KryStream &a, const KryPointN &f, int nbx, int nby, hipStream_t st);
//   v[0] = w[0] = r_jj, v[1] = r_ii' of the pending update (iip > 0), v[2] = r_ii of the dot (last == 0) or r_0 (last)
hipError_t kry_launch_mgsn(const KryStream &a, int jj, int iip, int last, int nbx, int nby, hipStream_t st);
//   r = r_0..r_L, u = u_0..u_L, v[0] = r~, w[0] = phi; r_0 and u_0 are updated in place
struct KryFinalN {
	const double *r[KRY_LMAX + 1], *u[KRY_LMAX + 1];
	double *r0, *u0;
	int L;
};
hipError_t kry_launch_finaln(const KryStream &a, const KryFinalN &f, int nbx, int nby, hipStream_t st);
hipError_t kry_launch_scalar(KryState *S, int phase, hipStream_t st);
hipError_t kry_launch_sum_slabs(const double *gather, int nslabs, double *out, hipStream_t st);
hipError_t kry_launch_axpy(double *dst, const double *x, const double *y, double c, long n, hipStream_t st);
hipError_t kry_launch_repack(double *dev, double *packed, int w, long rows, int px, bool to_device, hipStream_t st);
hipError_t kry_launch_face(double *field, double *buf, const KryGeo &g, int dir, int idx, bool to_buf, hipStream_t st);
hipError_t kry_launch_halo_push(const double *send_lo, double *slot_lo, unsigned long long *flag_lo, const double *send_hi,
                                double *slot_hi, unsigned long long *flag_hi, size_t count, unsigned long long seq,
                                unsigned *ticket, const KryState *S, hipStream_t st);
hipError_t kry_launch_halo_pull(double *recv_lo, const double *slot_lo, unsigned long long *flag_lo, double *recv_hi,
                                const double *slot_hi, unsigned long long *flag_hi, size_t count, unsigned long long seq,
                                long long timeout, KryState *S, hipStream_t st);
// a one-block grid reduction of `v` with r.nblk == 1: S->sums[0] receives v, or the sum of the
// ranks' / slabs' values when S->mail is set
hipError_t kry_launch_link_probe(const KryRed &r, double v, hipStream_t st);
hipError_t kry_launch_sub(double *dst, const double *x, const double *y, long n, hipStream_t st);
hipError_t kry_launch_bcs(double *d, const KryGeo &g, hipStream_t st, int ndirs = 3);   // x, y[, z] faces
// x and y faces of the output a.O of a general-boundary stencil kernel launched with y_off == 0, after the
// vector update of the rows its tiles leave out (-1 and row_hi .. my+1)
// (where an x face of the domain extrapolates, also the operator on the four points of row my that its treatment reads)
hipError_t kry_launch_bcs_rows(int kind, const KryFused &a, int row_hi, bool inbc, bool var, bool offd, bool exact,
                               hipStream_t st);
hipError_t kry_launch_gspmv(const KryGeo &g, const KryOp &op, bool var, bool offd, const double *psi,
                            const double *lam, const double *Dxx, const double *Dyy, const double *Dzz,
                            const double *Dxy, double *out, hipStream_t st);
hipError_t kry_launch_wdot(const KryGeo &g, const double *x, const double *y, int norm_mode, KryRed red,
                           hipStream_t st);
// The weighted dot with its partial sums and ticket in a file-scope device array instead of
// allocated memory (see kry_op_dot_static in include/krylov_amd.h).
hipError_t kry_static_scratch_ready(int device, double **out);
hipError_t kry_launch_wdot_static(int device, const KryGeo &g, const double *x, const double *y, int norm_mode,
                                  double *result, hipStream_t st);
