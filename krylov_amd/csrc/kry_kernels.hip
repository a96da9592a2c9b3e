// gfx950 kernels of the BiCGStab(l) stencil solver.
//
// Two families:
//   * fused kernels (k_fused, k_surf, k_mgs2, k_final; k_bicgn, k_mgsn, k_finaln for
//     l > 2): LDS-tiled, z-marching stencil fused with the vector update that produces
//     its input and with the dot products that consume its output; wave64 shuffles + a
//     single-pass grid reduction whose last block also advances the scalar recurrences.
//     Dirichlet faces, or (GEN instantiations) any mix of Dirichlet and extrapolating faces.
//   * generic kernels (g_*): one kernel per primitive, any face type: the prologue of
//     the general-face path, face types the fused kernels do not know, and an
//     independent cross-check of every configuration.
//
// The translation unit is built with -ffp-contract=off: every product and sum
// below rounds separately, in the order the reference evaluates them
// (BICGStab3D/Krylov.C:434-456, 524-530; BICGStab3D/solveLin_BICGStab.C:356-462),
// so results differ from a CPU evaluation only through summation order in dots.
#include "kry_kernels.h"

#include <math.h>

#include <algorithm>
#include <cstdlib>

#define NT 256                 // threads per block in every kernel here
#define NW (NT / 64)           // wavefronts per block
#define TXW KRY_TXW            // tile width in grid points = one 1 KiB wave load of double2
#define QOFF 2                  // streaming kernels start a plane at column 2: 128-byte aligned wave loads
#define SEG2 (KRY_SEG / 2)      // double2 elements of a plane segment handled per block and plane
#define LW 132                 // LDS row pitch: col 1 = left halo, 2..129 = tile, 130 = right halo

// ---------------------------------------------------------------------------
// scalar recurrences (solveLin_BICGStab.C:341-352, 390, 406-409, 419-446, 464-496)
// ---------------------------------------------------------------------------
__device__ inline bool really_is_nan(double xd)
{
	// Krylov::reallyIsNan(float), Krylov.C:553-559: narrows to float first.
	float x = (float)xd;
	return (__float_as_uint(x) & 0x7F800000u) == 0x7F800000u;
}

__device__ void kry_scalar_step(KryState *S, int phase_word)
{
	const double *s = S->sums;
	const int phase = phase_word & 0xff, jj = (phase_word >> 8) & 0xff, ii = (phase_word >> 16) & 0xff;
	switch (phase) {
	case PH_INIT: {
		double n = sqrt(s[0]);
		S->norm_init = n;
		S->norm = 1.e99;
		S->iter = 0;
		S->status = 0;
		S->done = 0;
		if (n < S->eps * S->norm_rhs) {           // :303-311, no iterations
			S->norm = n;
			S->done = 1;
			break;
		}
		// :317-320 then the head of the first pass (:341, :349-352, jj = 0)
		double rho0 = 1., alpha = 0., omega = 1.;
		rho0 = rho0 * (-omega);
		double rho1 = s[0];                       // (r0, r~) with r~ = r0
		S->beta = alpha * rho1 / rho0;
		S->rho0 = rho1;
		S->rho1 = rho1;
		S->alpha = alpha;
		S->omega = omega;
		break;
	}
	case PH_A:
		S->alpha = S->rho0 / s[0];                // :390
		break;
	case PH_B: {
		S->alpha0 = S->alpha;
		if (really_is_nan(S->alpha)) { S->status = 1; S->done = 1; break; }   // :406-409
		double rho1 = s[0];                       // :349 of the next jj
		S->beta = S->alpha * rho1 / S->rho0;      // :351
		S->rho0 = rho1;
		S->rho1 = rho1;
		break;
	}
	case PH_BL1:
		if (really_is_nan(S->alpha)) { S->status = 1; S->done = 1; break; }
		S->sigma[1] = s[1];                       // :427
		S->gammap[1] = s[0] / S->sigma[1];        // :428
		S->omega = S->gamma[1] = S->gammap[1];    // :431
		break;
	case PH_BL2:
		if (really_is_nan(S->alpha)) { S->status = 1; S->done = 1; break; }
		S->sigma[1] = s[0];                       // :427, jj = 1
		S->tau[1][2] = s[1] / S->sigma[1];        // :422
		break;
	case PH_M2:
		S->gammap[1] = s[2] / S->sigma[1];        // :428, jj = 1 (its dot rides in k_mgs2)
		S->sigma[2] = s[0];
		S->gammap[2] = s[1] / S->sigma[2];
		S->omega = S->gamma[2] = S->gammap[2];
		S->gamma[1] = S->gammap[1] - S->tau[1][2] * S->gamma[2];   // :434-439
		S->gammapp[1] = S->gamma[2];                                // :441-446
		break;
	case PH_F: {
		double norm = sqrt(s[0]);                 // :464 (unscaled single pass)
		S->norm = norm;
		if (S->iter < KRY_HIST_CAP) S->hist[S->iter] = norm;
		S->iter += 1;
		if (really_is_nan(norm)) { S->status = 3; S->done = 1; break; }      // :491-494
		if (!(norm > S->eps * S->norm_rhs)) { S->done = 1; break; }           // :496
		if (S->max_outer > 0 && S->iter >= S->max_outer) { S->status = 4; S->done = 1; break; }
		double rho0 = S->rho0 * (-S->omega);      // :341
		double rho1 = s[1];                       // :349, jj = 0
		S->beta = S->alpha * rho1 / rho0;
		S->rho0 = rho1;
		S->rho1 = rho1;
		break;
	}
	case PH_BN:
		if (really_is_nan(S->alpha)) { S->status = 1; S->done = 1; }          // :406-409
		break;
	case PH_TAU:
		S->tau[ii][jj] = s[0] / S->sigma[ii];                                 // :422
		break;
	case PH_SIG: {
		const int L = S->L;
		S->sigma[jj] = s[0];                                                  // :427
		S->gammap[jj] = s[1] / S->sigma[jj];                                  // :428
		if (jj < L) break;
		S->omega = S->gamma[L] = S->gammap[L];                                // :431
		for (int j = L - 1; j >= 1; --j) {                                    // :434-439
			S->gamma[j] = S->gammap[j];
			for (int i = j + 1; i <= L; ++i) S->gamma[j] -= S->tau[j][i] * S->gamma[i];
		}
		for (int j = 1; j < L; ++j) {                                         // :441-446
			S->gammapp[j] = S->gamma[j + 1];
			for (int i = j + 1; i < L; ++i) S->gammapp[j] += S->tau[j][i] * S->gamma[i + 1];
		}
		break;
	}
	default: break;
	}
}

__global__ void k_scalar(KryState *S, int phase)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		if (S->done && (phase & 0xff) != PH_INIT) return;
		kry_scalar_step(S, phase);
	}
}

// total of the per-slab sums, added in slab order
__global__ void k_sum_slabs(const double *gather, int nslabs, double *out)
{
	const int k = threadIdx.x;
	if (k >= KRY_NSUMS) return;
	double v = gather[k];
	for (int s = 1; s < nslabs; ++s) v += gather[s * KRY_NSUMS + k];
	out[k] = v;
}

// ---------------------------------------------------------------------------
// block reduction + single-pass grid reduction
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}

// Every block deposits ND partial sums; the block that draws the last ticket
// adds all partials in a fixed order (so the result does not depend on which
// block that is), publishes them in S->sums and, when the solver runs on one
// slab, advances the scalar recurrences.  The partials travel between blocks
// as agent-scope (write-through / cache-bypassing) stores and loads, so no
// cache write-back is needed around the ticket: stores, vmcnt drain, barrier,
// relaxed agent-scope ticket; the last block issues one agent-scope acquire and
// reads the partials with ordinary loads.
__device__ inline void pub_store(double *p, double v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double pub_load(const double *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Tail of a grid reduction when the domain is split over processes: this rank's
// sums (the rows of `red`, still to be added) go to every rank's mailbox, the
// ranks' contributions are added in rank order.  Run by the whole last block;
// lane r of the block talks to rank r.
template <int ND>
__device__ inline void mail_allreduce(double (*red)[ND], const KryRed &r)
{
	static_assert(ND <= KRY_NSUMS, "message slot too small");
	const KryMail *m = r.S->mail;
	const int tid = threadIdx.x, n = m->nranks, me = m->rank;
	__shared__ double mine[KRY_NSUMS];
	__shared__ int failed;
	unsigned long long *seqp = reinterpret_cast<unsigned long long *>(m->box);
	if (tid < ND) {
		double v = red[0][tid];
#pragma unroll
		for (int i = 1; i < NW; ++i) v += red[i][tid];
		mine[tid] = v;
	}
	if (tid == 0) failed = 0;
	__syncthreads();
	const long long t_begin = wall_clock64();
	const unsigned long long seq = *seqp + 1ull;
	const long slot0 = KRY_MAIL_SLOT + (long)(seq & 1ull) * n * KRY_MAIL_SLOT;
	for (int p = tid; p < n; p += NT) {
		double *dst = m->peer[p] + slot0 + (long)me * KRY_MAIL_SLOT;
#pragma unroll
		for (int k = 0; k < ND; ++k) __hip_atomic_store(dst + k, mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + KRY_NSUMS), seq, __ATOMIC_RELEASE,
		                   __HIP_MEMORY_SCOPE_SYSTEM);
		const unsigned long long *flag =
		    reinterpret_cast<const unsigned long long *>(m->box + slot0 + (long)p * KRY_MAIL_SLOT + KRY_NSUMS);
		const long long t0 = wall_clock64();
		unsigned spins = 0;
		while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
			__builtin_amdgcn_s_sleep(4);
			if ((++spins & 255u) == 0u && wall_clock64() - t0 > m->timeout) {
				failed = 1;          // the peer never arrived: give up, every later kernel exits at once
				break;
			}
		}
	}
	__syncthreads();
	if (tid == 0) {
		const double *src = m->box + slot0;
#pragma unroll
		for (int k = 0; k < ND; ++k) {
			double v = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			for (int p = 1; p < n; ++p)
				v += __hip_atomic_load(src + (long)p * KRY_MAIL_SLOT + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			r.S->sums[k] = failed ? __builtin_nan("") : v;
		}
		*seqp = seq;
		r.S->wait_sum_ticks += wall_clock64() - t_begin;
		__hip_atomic_store(r.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (failed) {
			r.S->status = -3;    // KRY_ERR_COMM
			r.S->done = 1;
		} else if (r.S->local_scalars && r.phase != PH_NONE) {
			kry_scalar_step(r.S, r.phase);
		}
	}
}

// slot: where this block's partial sums go.  It names the block's WORK (tile, plane segment), not its
// place in the launch, so the order in which the partials are added does not depend on the order in
// which the blocks take the work (-1: the block's own index).
template <int ND>
__device__ inline void reduce_and_finalize(double (&acc)[ND], const KryRed &r, long slot = -1)
{
	const long bid = slot >= 0 ? slot : blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
	__shared__ double red[NW][ND];
	__shared__ int is_last;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

#pragma unroll
	for (int k = 0; k < ND; ++k) {
		double v = wave_sum(acc[k]);
		if (lane == 0) red[w][k] = v;
	}
	__syncthreads();
	if (tid < ND) {
		double v = red[0][tid];
#pragma unroll
		for (int i = 1; i < NW; ++i) v += red[i][tid];
		pub_store(&r.partial[bid * ND + tid], v);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (tid == 0) {
		unsigned t = __hip_atomic_fetch_add(r.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		is_last = (t == (unsigned)r.nblk - 1u);
	}
	__syncthreads();
	if (!is_last) return;

	// one agent-scope acquire for the block, then ordinary (pipelined) loads
	if (tid == 0) {
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	__syncthreads();
	double tot[ND];
#pragma unroll
	for (int k = 0; k < ND; ++k) tot[k] = 0.;
	const double *part = r.partial;
#pragma unroll 4
	for (int b = tid; b < r.nblk; b += NT) {
#pragma unroll
		for (int k = 0; k < ND; ++k) tot[k] += part[(long)b * ND + k];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < ND; ++k) {
		double v = wave_sum(tot[k]);
		if (lane == 0) red[w][k] = v;
	}
	__syncthreads();
	if (r.S && r.S->mail) {
		mail_allreduce<ND>(red, r);
		return;
	}
	if (tid == 0) {
		double *dst = r.S ? r.S->sums : r.out;
#pragma unroll
		for (int k = 0; k < ND; ++k) {
			double v = red[0][k];
#pragma unroll
			for (int i = 1; i < NW; ++i) v += red[i][k];
			dst[k] = (!r.S && r.out_sqrt) ? sqrt(v) : v;
		}
		__hip_atomic_store(r.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (r.S && r.S->local_scalars && r.phase != PH_NONE) kry_scalar_step(r.S, r.phase);
	}
}

// ---------------------------------------------------------------------------
// fused stencil kernel
// ---------------------------------------------------------------------------
__device__ inline double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ inline void st2(double *p, double2 v) { *reinterpret_cast<double2 *>(p) = v; }
// Streaming (non-temporal) forms for data that is touched once per kernel: they
// keep the one-pass streams from evicting the lines neighbouring tiles re-read.
typedef double kry_d2 __attribute__((ext_vector_type(2)));
__device__ inline double2 ld2s(const double *p)
{
	kry_d2 v = __builtin_nontemporal_load(reinterpret_cast<const kry_d2 *>(p));
	double2 r;
	r.x = v.x;
	r.y = v.y;
	return r;
}
__device__ inline void st2s(double *p, double2 v)
{
	kry_d2 w;
	w.x = v.x;
	w.y = v.y;
	__builtin_nontemporal_store(w, reinterpret_cast<kry_d2 *>(p));
}
__device__ inline void st1s(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ inline double2 axpy2(double2 x, double2 y, double c)
{
	// x + (y*c): product rounded, then sum rounded (solveLin_BICGStab.C:358-360, 393)
	double2 t;
	t.x = y.x * c;
	t.y = y.y * c;
	t.x = x.x + t.x;
	t.y = x.y + t.y;
	return t;
}
__device__ inline double axpy1(double x, double y, double c)
{
	double t = y * c;
	return x + t;
}

template <bool EXACT>
__device__ inline double div2d(double v, double two_d, double inv_two_d)
{
	if (EXACT) return v * inv_two_d;   // two_d is a power of two: identical rounding
	return v / two_d;
}

// One output value of the operator at a point.
//   p*: psi at the point and its neighbours; VAR adds coefficient fields.
struct Sten {
	double pc, xm, xp, ym, yp, zm, zp;
	double pmm, pmp, ppm, ppp;              // psi(x-1,y-1), (x-1,y+1), (x+1,y-1), (x+1,y+1)
	double lam;
	double dxx, dyy, dzz, dxy;
	double dxx_m, dxx_p, dyy_m, dyy_p, dzz_m, dzz_p;
	double dxy_xm, dxy_xp, dxy_ym, dxy_yp;
};

template <bool VAR, bool OFFD, bool EXACT>
__device__ inline double apply_op(const Sten &s, const KryOp &op)
{
	if (!VAR) {
		// Krylov.C:524-530
		double a = op.c[0] * (s.xp + s.xm);
		double b = op.c[1] * (s.yp + s.ym);
		double c = op.c[2] * (s.zp + s.zm);
		double d = (op.cd2 + s.lam) * s.pc;
		return ((a + b) + c) - d;
	} else {
		// Krylov.C:434-456, summed left to right
		double cx = op.c[0] * s.dxx, cy = op.c[1] * s.dyy, cz = op.c[2] * s.dzz;
		double t1 = cx * (s.xp + s.xm);
		double t2 = cy * (s.yp + s.ym);
		double t3 = cz * (s.zp + s.zm);
		double t4 = (2. * ((cx + cy) + cz) + s.lam) * s.pc;
		double acc = ((t1 + t2) + t3) - t4;
		double gx = div2d<EXACT>(s.dxx_p - s.dxx_m, op.two_d[0], op.inv_two_d[0]);
		double gy = div2d<EXACT>(s.dyy_p - s.dyy_m, op.two_d[1], op.inv_two_d[1]);
		if (OFFD) {
			double t5 = (op.cxy * s.dxy) * (((s.ppp - s.ppm) - s.pmp) + s.pmm);
			acc = acc + t5;
			gx = gx + div2d<EXACT>(s.dxy_yp - s.dxy_ym, op.two_d[1], op.inv_two_d[1]);
			gy = div2d<EXACT>(s.dxy_xp - s.dxy_xm, op.two_d[0], op.inv_two_d[0]) + gy;
		}
		double t6 = div2d<EXACT>(gx * (s.xp - s.xm), op.two_d[0], op.inv_two_d[0]);
		double t7 = div2d<EXACT>(gy * (s.yp - s.ym), op.two_d[1], op.inv_two_d[1]);
		double gz = div2d<EXACT>(s.dzz_p - s.dzz_m, op.two_d[2], op.inv_two_d[2]);
		double t8 = div2d<EXACT>(gz * (s.zp - s.zm), op.two_d[2], op.inv_two_d[2]);
		return ((acc + t6) + t7) + t8;
	}
}

// ---------------------------------------------------------------------------
// do_BCs on the INPUT of the constant-coefficient operator (Krylov.C:516), for the
// fused general-boundary kernels.  The stencil input psi = X + c*Y is produced in the
// kernel; its boundary treatment (BoundaryHandler.C:416-568: x faces, then y, then z,
// each over the tangential range the reference gives it) makes the value of a cell in
// the two outermost layers a fixed expression in untreated values further inside.
// bc_psi() evaluates that expression for one cell from memory -- the same products and
// differences, in the same order, as the sequential treatment -- so no thread needs
// another thread's result.  Defined for the cells the 7-point operator reads (grid
// points and the ghost cells straight behind a face); edge and corner ghosts, which
// nothing reads, get some value of the same recursion.
// ---------------------------------------------------------------------------
struct PsiSrc {
	const double *X, *Y, *Z1, *Z2;
	double c, dz1, dz2;
	int nz;
};

__device__ __forceinline__ double psi_raw(const PsiSrc &s, const KryGeo &g, int i, int j, int k)
{
	const long p = kry_idx(g, i, j, k);
	double y = s.Y[p];
	if (s.nz >= 1) y = axpy1(y, s.Z1[p], s.dz1);
	if (s.nz >= 2) y = axpy1(y, s.Z2[p], s.dz2);
	return axpy1(s.X[p], y, s.c);
}

// How do_BCs forms a cell's value along one axis: from the cell itself (P_ID), zero (P_ZERO),
// 2 f(a) - f(b) (P_LIN), or the second extrapolated cell 2 (2 f(a) - f(b)) - f(a) (P_LIN2).
enum { P_ID = 0, P_ZERO, P_LIN, P_LIN2 };
struct PsiAxis {
	int mode, a, b;
};

// index c of an axis with last point m: lower face type tlo (treated unless `nlo`: a neighbour lies
// behind it), upper face type thi.  zlow: the lower z face also rewrites the boundary plane, from
// planes 1 and 2 (BoundaryHandler.C:536-544); the lower x and y faces fill the ghost cell only.
__device__ __forceinline__ PsiAxis psi_axis(int c, int m, int nlo, int nhi, int tlo, int thi, bool zlow)
{
	PsiAxis r = {P_ID, c, c};
	if (!nlo && c <= 0) {
		if (tlo == 0) r.mode = P_ZERO;
		else if (tlo == 1 && c < 0) { r.mode = P_LIN; r.a = 0; r.b = 1; }
		else if (tlo == 1 && zlow) { r.mode = P_LIN; r.a = 1; r.b = 2; }
	}
	if (!nhi && c >= m) {
		if (thi == 0) r.mode = P_ZERO;
		else if (thi == 1) { r.mode = (c == m) ? P_LIN : P_LIN2; r.a = m - 1; r.b = m - 2; }
	}
	return r;
}

__device__ __forceinline__ double psi_comb(int mode, double fa, double fb)
{
	double t = 2. * fa - fb;
	if (mode == P_LIN2) t = 2. * t - fa;
	return t;
}

// value of cell (i,j,k) after do_BCs; `v` is its untreated value (returned where no face reaches)
__device__ __forceinline__ double bc_psi(const PsiSrc &s, const KryGeo &g, double v, int i, int j, int k)
{
	if (i > 0 && i < g.mx && j > 0 && j < g.my && k > 0 && k < g.mz) return v;
	if (i < -1 || i > g.mx + 1 || j < -1 || j > g.my + 1 || k < -1 || k > g.mz + 1) return v;   // tile overshoot
	// x faces reach over the whole tangential box, y faces over i = 0..mx, z faces over i = 0..mx, j = 0..my
	const PsiAxis ax = psi_axis(i, g.mx, g.nb[0], g.nb[1], g.bc[0], g.bc[1], false);
	PsiAxis ay = {P_ID, j, j}, az = {P_ID, k, k};
	const bool in_x = (i >= 0 && i <= g.mx);
	if (in_x) ay = psi_axis(j, g.my, g.nb[2], g.nb[3], g.bc[2], g.bc[3], false);
	if (in_x && j >= 0 && j <= g.my) az = psi_axis(k, g.mz, g.has_lo, g.has_hi, g.bc[4], g.bc[5], true);
	if (ax.mode == P_ZERO || ay.mode == P_ZERO || az.mode == P_ZERO) return 0.;
	if (ax.mode == P_ID && ay.mode == P_ID && az.mode == P_ID) return v;
	// x treatment of the untreated values, y treatment of those, z treatment of those
	auto ex = [&](int jj, int kk) __attribute__((always_inline)) -> double {
		const double fa = psi_raw(s, g, ax.a, jj, kk);
		if (ax.mode == P_ID) return fa;
		return psi_comb(ax.mode, fa, psi_raw(s, g, ax.b, jj, kk));
	};
	auto ey = [&](int kk) __attribute__((always_inline)) -> double {
		const double fa = ex(ay.a, kk);
		if (ay.mode == P_ID) return fa;
		return psi_comb(ay.mode, fa, ex(ay.b, kk));
	};
	const double fa = ey(az.a);
	if (az.mode == P_ID) return fa;
	return psi_comb(az.mode, fa, ey(az.b));
}

// bc_psi with the memory accesses of all the untreated values it needs issued before any of them is
// used (the nested form above waits for one cell's operands before it asks for the next: up to eight
// memory latencies in a row).  Same values, same products and differences in the same order.
__device__ __forceinline__ double bc_psi_batched(const PsiSrc &s, const KryGeo &g, int i, int j, int k)
{
	PsiAxis ax = {P_ID, i, i}, ay = {P_ID, j, j}, az = {P_ID, k, k};
	const bool touched = !(i > 0 && i < g.mx && j > 0 && j < g.my && k > 0 && k < g.mz) &&
	                     !(i < -1 || i > g.mx + 1 || j < -1 || j > g.my + 1 || k < -1 || k > g.mz + 1);
	if (touched) {
		ax = psi_axis(i, g.mx, g.nb[0], g.nb[1], g.bc[0], g.bc[1], false);
		const bool in_x = (i >= 0 && i <= g.mx);
		if (in_x) ay = psi_axis(j, g.my, g.nb[2], g.nb[3], g.bc[2], g.bc[3], false);
		if (in_x && j >= 0 && j <= g.my) az = psi_axis(k, g.mz, g.has_lo, g.has_hi, g.bc[4], g.bc[5], true);
	}
	if (ax.mode == P_ZERO || ay.mode == P_ZERO || az.mode == P_ZERO) return 0.;
	// corner m of the 2 x 2 x 2 set of untreated cells: bit 0 picks ax.b, bit 1 ay.b, bit 2 az.b
	double x[8], y[8], z1[8], z2[8];
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		x[m] = y[m] = z1[m] = z2[m] = 0.;
		const bool need = (!(m & 1) || ax.mode != P_ID) && (!(m & 2) || ay.mode != P_ID) && (!(m & 4) || az.mode != P_ID);
		if (need) {
			const long p = kry_idx(g, (m & 1) ? ax.b : ax.a, (m & 2) ? ay.b : ay.a, (m & 4) ? az.b : az.a);
			x[m] = s.X[p];
			y[m] = s.Y[p];
			if (s.nz >= 1) z1[m] = s.Z1[p];
			if (s.nz >= 2) z2[m] = s.Z2[p];
		}
	}
	double f[8];
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		double yy = y[m];
		if (s.nz >= 1) yy = axpy1(yy, z1[m], s.dz1);
		if (s.nz >= 2) yy = axpy1(yy, z2[m], s.dz2);
		f[m] = axpy1(x[m], yy, s.c);
	}
#pragma unroll
	for (int m = 0; m < 8; m += 2) f[m] = (ax.mode == P_ID) ? f[m] : psi_comb(ax.mode, f[m], f[m + 1]);
#pragma unroll
	for (int m = 0; m < 8; m += 4) f[m] = (ay.mode == P_ID) ? f[m] : psi_comb(ay.mode, f[m], f[m + 2]);
	return (az.mode == P_ID) ? f[0] : psi_comb(az.mode, f[0], f[4]);
}

template <int KIND> struct KindTraits;
template <> struct KindTraits<KF_A0>   { static constexpr int ND = 1; static constexpr bool UPD = true;  };
template <> struct KindTraits<KF_A1>   { static constexpr int ND = 1; static constexpr bool UPD = true;  };
template <> struct KindTraits<KF_B0>   { static constexpr int ND = 1; static constexpr bool UPD = true;  };
template <> struct KindTraits<KF_B0L>  { static constexpr int ND = 2; static constexpr bool UPD = true;  };
template <> struct KindTraits<KF_B1L>  { static constexpr int ND = 2; static constexpr bool UPD = true;  };
template <> struct KindTraits<KF_RES>  { static constexpr int ND = 1; static constexpr bool UPD = false; };
template <> struct KindTraits<KF_SPMV> { static constexpr int ND = 1; static constexpr bool UPD = false; };

// Tile: TXW points in x (one double2 per lane), TY rows, marched over a chunk of
// planes.  psi of plane z lives in LDS buffer z&1 (tile + one halo ring), psi of
// planes z-1, z, z+1 at the thread's own points lives in registers.
//
// GEN (general boundaries, every face of type 0 or 1): the vector updates also
// cover the ghost cells, because the reference's whole-array updates
// (solveLin_BICGStab.C:356-362, 392-394) act on them and a later stencil reads
// them.  Rows j = -1 .. my+1 and planes -1 .. mz+1 are tiled like the others; in
// x the tiles stay on the aligned columns i = 0 .. and the lanes that fetch the
// halo columns also update i = -1 (first tile) and i = mx, mx+1 (last tile).  The
// operator is evaluated on the points 0 .. mx-1 (Krylov.C:429-459; its value at
// i = mx is overwritten by either face type before anything reads it,
// BoundaryHandler.C:444-460); its boundary treatment and the boundary-plane terms
// of the dot products follow in g_bc and k_surf.
// (Compiled for the register budget the code wants -- 2 blocks per CU for the field-coefficient
// tiles.  Forcing 3 or 4 blocks per CU through __launch_bounds__ ran 3-25 % slower at 512^3.)
// INBC (constant coefficients on faces that are not all Dirichlet): psi is passed through
// do_BCs before the operator reads it and before it is stored, as Krylov.C:516 does.  That
// changes cells on and behind the faces only; g_psi_shell evaluates them ahead of this
// kernel, which reads them back where a tile or its halo ring holds such a cell.
template <int KIND, int TY, bool VAR, bool OFFD, bool EXACT, bool GEN, bool INBC = false>
__global__ __launch_bounds__(NT) void k_fused(const KryFused a)
{
	static_assert(!INBC || (GEN && !VAR && KindTraits<KIND>::UPD), "input treatment: constant-coefficient updates only");
	constexpr int VD = TY / NW;                 // rows per thread
	constexpr int ND = KindTraits<KIND>::ND;
	constexpr bool UPD = KindTraits<KIND>::UPD; // psi = X + Y*c and is stored to Xout
	constexpr bool IS_A = (KIND == KF_A0 || KIND == KF_A1);
	__shared__ __align__(16) double lds[2][(TY + 2) * LW];

	const KryGeo &g = a.g;
	KryState *S = a.S;
	if (KIND != KF_SPMV && KIND != KF_RES && S->done) return;

	const double cpsi = UPD ? (IS_A ? -S->beta : -S->alpha) : 0.;
	// l = 1: KF_A0 first applies the u0 update that closed the previous pass
	// (solveLin_BICGStab.C:452): u0 -= gamma_1*u_1, so the end-of-pass kernel neither
	// reads u_1 nor writes u0.  (nz = 0 for l = 2, whose end-of-pass kernel has u0 and
	// u_1 in registers anyway and closes u0 itself.)
	constexpr bool DEFER = (KIND == KF_A0);
	const int nz = DEFER ? a.nz : 0;
	const double dz1 = DEFER ? -S->gamma[S->L] : 0., dz2 = DEFER ? -S->gamma[1] : 0.;

	// XCD-aware block -> tile map: blocks that share blockIdx%8 get neighbouring tiles
	int bid;
	{
		const int nb = a.nblk, b = blockIdx.x;
		const int q = nb >> 3, r = nb & 7, x = b & 7;
		bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
		if (a.rev) bid = nb - 1 - bid;
	}
	int tx = bid % a.ntx;
	int ty = (bid / a.ntx) % a.nty;
	int ch = bid / (a.ntx * a.nty);
	// a launch covers all chunks, the outermost ones (chunk 0 and the last n_edge-1
	// chunks: those holding the planes the neighbours need), or the ones in between
	if (a.ch_sel == 1) ch = ch ? a.nchunk - a.n_edge + ch : 0;
	else if (a.ch_sel == 2) ch += 1;
	else if (a.ch_sel >= 3) {
		// the tiles inside / outside a box of tile indices; either way a tile keeps the partial-sum
		// slot it has in the undivided launch
		const int nxi = a.ib[1] - a.ib[0] + 1, nyi = a.ib[3] - a.ib[2] + 1, nci = a.ib[5] - a.ib[4] + 1;
		int r = bid;
		if (a.ch_sel == 3) {
			tx = a.ib[0] + r % nxi;
			ty = a.ib[2] + (r / nxi) % nyi;
			ch = a.ib[4] + r / (nxi * nyi);
		} else {
			const int plane_tiles = a.ntx * a.nty;
			const int n1 = (a.nchunk - nci) * plane_tiles;             // whole chunks outside
			const int row_tiles = (a.nty - nyi) * a.ntx;               // whole tile rows outside, per inner chunk
			const int n2 = nci * row_tiles;
			if (r < n1) {
				const int c = r / plane_tiles;
				ch = c < a.ib[4] ? c : c + nci;
				r -= c * plane_tiles;
				ty = r / a.ntx;
				tx = r % a.ntx;
			} else if (r < n1 + n2) {
				r -= n1;
				ch = a.ib[4] + r / row_tiles;
				r %= row_tiles;
				const int y = r / a.ntx;
				ty = y < a.ib[2] ? y : y + nyi;
				tx = r % a.ntx;
			} else {
				r -= n1 + n2;
				const int col_tiles = a.ntx - nxi;                     // tiles outside per inner row
				ch = a.ib[4] + r / (nyi * col_tiles);
				r %= nyi * col_tiles;
				ty = a.ib[2] + r / col_tiles;
				const int x = r % col_tiles;
				tx = x < a.ib[0] ? x : x + nxi;
			}
		}
		bid = (ch * a.nty + ty) * a.ntx + tx;
	}

	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int z0 = a.zlo + ch * a.zc;
	const int z1 = min(z0 + a.zc, a.zhi + 1);   // exclusive
	// (GEN: tile rows from j = -1, or from j = 0 with the rows outside the tiles left to g_bc_xy_rows: a.y_off)
	const int y0 = ty * TY - (GEN ? a.y_off : 0);     // first row (grid index j) of the tile
	const int c0 = 2 + tx * TXW;                // first column of the tile
	const int col = c0 + 2 * lane;              // column of .x ; grid index i = col - 2
	const int ix = col - 2;
	// strict interior: the points that count in the dot products here
	const int ix_first = (GEN && g.nb[0]) ? 0 : 1;     // an x interface counts on its upper side
	const bool mxx = (ix >= ix_first && ix <= g.mx - 1);
	const bool mxy = (ix + 1 >= ix_first && ix + 1 <= g.mx - 1);
	// cells whose vector updates are stored / points whose operator value is stored
	const bool uxx = GEN ? (ix >= -1 && ix <= g.mx + 1) : mxx;
	const bool uxy = GEN ? (ix + 1 >= -1 && ix + 1 <= g.mx + 1) : mxy;
	const bool oxx = GEN ? (ix >= 0 && ix <= g.mx) : mxx;
	const bool oxy = GEN ? (ix + 1 >= 0 && ix + 1 <= g.mx) : mxy;

	long off[VD];        // in-plane offset of the own double2
	int lidx[VD];        // LDS index of the own double2
	bool mrow[VD], urow[VD], orow[VD];
#pragma unroll
	for (int e = 0; e < VD; ++e) {
		const int r = w * VD + e;
		const int j = y0 + r;
		off[e] = (long)(j + 1) * g.px + col;
		lidx[e] = (r + 1) * LW + 2 + 2 * lane;
		mrow[e] = (j >= 1 && (j <= g.my - 1 || (GEN && j == g.my && g.nb[3])));   // a y interface counts on its lower side
		urow[e] = GEN ? (j <= g.my + 1) : mrow[e];
		orow[e] = GEN ? (j >= 0 && j <= g.my) : mrow[e];
	}
	// halo duties: waves 0,1 load the row above / below the tile, wave 2 the two columns
	const bool hrow = tid < 128;
	const int hr = tid >> 6;                                    // 0: row -1, 1: row TY
	const long hoff = (long)(y0 + (hr ? TY : -1) + 1) * g.px + col;
	const int hl = (hr ? TY + 1 : 0) * LW + 2 + 2 * lane;
	// (GEN: a third column, one further right, which only the last tile stores)
	const int cidx = tid - 128;
	const bool hcol = (cidx >= 0 && cidx < (GEN ? 3 : 2) * (TY + 2));
	const int crr = GEN ? cidx % (TY + 2) : cidx >> 1, cside = GEN ? cidx / (TY + 2) : cidx & 1;
	const int ccol = cside ? c0 + TXW + (cside - 1) : c0 - 1;
	const long coff = (long)(y0 + crr - 1 + 1) * g.px + ccol;
	const int cl = crr * LW + (cside ? 130 : 1);
	// Dirichlet faces: the solver's work vectors are zero on and behind a face, so a halo column on or
	// behind an x face of the domain (which sits in a cache line nothing else reads) is not loaded.
	// (Not so for the residual and plain operator kinds: their input is the caller's array.)
	const bool cload = hcol && (GEN || !UPD || !((cside == 0 && tx == 0) || (cside == 1 && ccol - 2 >= g.mx)));
	// halo-column cells that are ghost / face cells of this tile's rows: updated here
	// (behind a Dirichlet face of the domain these cells are zero and stay zero under every update)
	const bool cstore = GEN && hcol && crr >= 1 && crr <= TY && (y0 + crr - 1) <= g.my + 1 &&
	                    (cside ? (tx == a.ntx - 1 && ccol - 2 <= g.mx + 1 && (g.nb[1] || g.bc[1] != 0))
	                           : (tx == 0 && (g.nb[0] || g.bc[0] != 0)));

	double2 pm[VD], pc[VD], pp[VD];
	double2 dzm[VD], dzc[VD], dzp[VD];
	double2 nx[VD], ny[VD];
	double2 nz1[DEFER ? VD : 1], nz2[DEFER ? VD : 1];
	double2 hx = {0., 0.}, hy = {0., 0.}, hz1 = {0., 0.}, hz2 = {0., 0.};
	double sx = 0., sy = 0., sz1 = 0., sz2 = 0.;
	double acc[ND];
#pragma unroll
	for (int k = 0; k < ND; ++k) acc[k] = 0.;

	auto plane = [&](int z) -> long { return (long)(z + 1) * g.ps; };

	// input treatment of the constant-coefficient operator (INBC): the treated psi of every cell that is not
	// strictly inside the box was written to Xout by g_psi_shell before this launch; those cells take it from
	// there (fetched with the operands of their plane) instead of forming psi from the operands.
	const bool fx = INBC && (ix <= 0 || ix >= g.mx), fy = INBC && (ix + 1 <= 0 || ix + 1 >= g.mx);
	bool frow[VD];
#pragma unroll
	for (int e = 0; e < VD; ++e) frow[e] = INBC && (y0 + w * VD + e <= 0 || y0 + w * VD + e >= g.my);
	const bool fh = INBC && (y0 + (hr ? TY : -1) <= 0 || y0 + (hr ? TY : -1) >= g.my);
	const bool fc = INBC && (ccol - 2 <= 0 || ccol - 2 >= g.mx || y0 + crr - 1 <= 0 || y0 + crr - 1 >= g.my);
	double2 bx[INBC ? VD : 1], hb = {0., 0.};
	double sb = 0.;
	int zi = 0;          // the plane whose operands were issued last
	auto bc2 = [&](double2 v, double2 t, bool frw) __attribute__((always_inline)) -> double2 {
		if (INBC) {
			const bool fz = (zi <= 0 || zi >= g.mz);
			if (fz || frw || fx) v.x = t.x;
			if (fz || frw || fy) v.y = t.y;
		}
		return v;
	};

	// issue the loads that psi of plane z needs (own points and halo ring)
	auto issue = [&](int z) {
		zi = z;
#ifdef KRY_EXP_CLAMPZ
		const long pb = plane(min(max(z, z0), z1 - 1));   // timing experiment only (wrong results): no chunk halo planes from memory
#else
		const long pb = plane(z);
#endif
#pragma unroll
		for (int e = 0; e < VD; ++e) {
			nx[e] = ld2(a.X + pb + off[e]);
			if (UPD) ny[e] = ld2(a.Y + pb + off[e]);
			if (DEFER && nz >= 1) nz1[DEFER ? e : 0] = ld2(a.Z1 + pb + off[e]);
			if (DEFER && nz >= 2) nz2[DEFER ? e : 0] = ld2(a.Z2 + pb + off[e]);
		}
#ifdef KRY_EXP_NOHROW
		if (false) {                             // timing experiment only (wrong results)
#else
		if (hrow) {
#endif
			hx = ld2(a.X + pb + hoff);
			if (UPD) hy = ld2(a.Y + pb + hoff);
			if (DEFER && nz >= 1) hz1 = ld2(a.Z1 + pb + hoff);
			if (DEFER && nz >= 2) hz2 = ld2(a.Z2 + pb + hoff);
		}
#ifdef KRY_EXP_NOHCOL
		if (false) {                             // timing experiment only (wrong results)
#else
		if (cload) {
#endif
			sx = a.X[pb + coff];
			if (UPD) sy = a.Y[pb + coff];
			if (DEFER && nz >= 1) sz1 = a.Z1[pb + coff];
			if (DEFER && nz >= 2) sz2 = a.Z2[pb + coff];
		}
		if (INBC) {
			const bool fz = (z <= 0 || z >= g.mz);
#pragma unroll
			for (int e = 0; e < VD; ++e)
				if (fz || frow[e] || fx || fy) bx[INBC ? e : 0] = ld2(a.Xout + pb + off[e]);
			if (hrow && (fz || fh || fx || fy)) hb = ld2(a.Xout + pb + hoff);
			if (cload && (fz || fc)) sb = a.Xout[pb + coff];
		}
	};
	// psi = X + c*Y, with Y first brought up to date where its last update was deferred
	auto psi2 = [&](double2 x, double2 y, double2 z1, double2 z2) -> double2 {
		if (!UPD) return x;
		if (DEFER && nz >= 1) y = axpy2(y, z1, dz1);
		if (DEFER && nz >= 2) y = axpy2(y, z2, dz2);
		return axpy2(x, y, cpsi);
	};
	auto psi1 = [&](double x, double y, double z1, double z2) -> double {
		if (!UPD) return x;
		if (DEFER && nz >= 1) y = axpy1(y, z1, dz1);
		if (DEFER && nz >= 2) y = axpy1(y, z2, dz2);
		return axpy1(x, y, cpsi);
	};
	// psi of the issued plane at the own points
	auto own_psi = [&](double2 (&dst)[VD]) {
#pragma unroll
		for (int e = 0; e < VD; ++e)
			dst[e] = bc2(psi2(nx[e], ny[e], nz1[DEFER ? e : 0], nz2[DEFER ? e : 0]), bx[INBC ? e : 0], frow[e]);
	};
	// psi of the issued plane at this lane's halo-column cell
	auto col_psi = [&]() __attribute__((always_inline)) -> double {
		const double v = psi1(sx, sy, sz1, sz2);
		return (INBC && (fc || zi <= 0 || zi >= g.mz)) ? sb : v;
	};
	// write psi (own + halo) of the issued plane into an LDS buffer
	auto to_lds = [&](const double2 (&v)[VD], double *buf) {
#pragma unroll
		for (int e = 0; e < VD; ++e) st2(buf + lidx[e], v[e]);
		if (hrow) st2(buf + hl, bc2(psi2(hx, hy, hz1, hz2), hb, fh));
		if (hcol && (!GEN || cside < 2)) buf[cl] = col_psi();
	};
	// store psi to its ping-pong partner at interior points of plane z
	auto store_psi = [&](const double2 (&v)[VD], int z) {
		if (!UPD) return;
		const long pb = plane(z);
#pragma unroll
		for (int e = 0; e < VD; ++e) {
			if (urow[e]) {
				if (uxx && uxy) st2s(a.Xout + pb + off[e], v[e]);
				else if (uxx) a.Xout[pb + off[e]] = v[e].x;
				else if (uxy) a.Xout[pb + off[e] + 1] = v[e].y;
			}
		}
		if (GEN && cstore) a.Xout[pb + coff] = col_psi();   // sx, sy, ... still hold plane z
		if (GEN && DEFER && nz >= 1) {
			// l = 1: the reference writes u1 = A u0 over the old u1 (one array), and the x- and y-face treatment
			// of that output reads the array's ghost rows j = -1, my+1 and ghost planes (BoundaryHandler.C:420-421,
			// 479-480), which still hold the old u1.  Here the output goes to the partner buffer: hand it the old
			// ghost rows and planes (the xy edge ghosts derived from the rows are read by the 11-point operator).
			const bool gplane = (z == -1 && !g.has_lo) || (z == g.mz + 1 && !g.has_hi);
#pragma unroll
			for (int e = 0; e < VD; ++e) {
				const int j = y0 + w * VD + e;
				if (urow[e] && (gplane || j == -1 || j == g.my + 1)) {
					if (uxx && uxy) st2(a.O + pb + off[e], nz1[DEFER ? e : 0]);
					else if (uxx) a.O[pb + off[e]] = nz1[DEFER ? e : 0].x;
					else if (uxy) a.O[pb + off[e] + 1] = nz1[DEFER ? e : 0].y;
				}
			}
		}
	};

	// ---- prologue: planes z0-1, z0 and the loads of z0+1 ----
	issue(z0 - 1);
	own_psi(pm);
	if (VAR) {
#pragma unroll
		for (int e = 0; e < VD; ++e) {
			dzm[e] = ld2(a.Dzz + plane(z0 - 1) + off[e]);
			dzc[e] = ld2(a.Dzz + plane(z0) + off[e]);
		}
	}
	if (z0 == a.zlo && (GEN || g.has_lo) && z0 == 0) {   // plane -1 is a neighbour's plane or a live ghost: keep it current
		store_psi(pm, -1);
	}
	issue(z0);
	own_psi(pc);
	to_lds(pc, lds[z0 & 1]);
	store_psi(pc, z0);
	issue(z0 + 1);
	// one-pass operands of the stencil (lambda, r~) are fetched one plane ahead as well: nothing the
	// stencil of a plane waits for is requested in the same step (256^3 constant coefficients: -2.5 %;
	// general faces with field coefficients, 512^3: -6 %, still two blocks per compute unit at 245-251
	// registers).  Not in the instantiations with Dxy: measured without effect there, and the first
	// kernel of a pass would lose a resident block to it.
	constexpr bool HAS_RT = (KIND == KF_A0 || KIND == KF_A1 || KIND == KF_B0);
	constexpr bool PFC = !OFFD;
	double2 lamn[VD], rtn[HAS_RT ? VD : 1];
	auto issue_coef = [&](int z) {
		const long pbn = plane(z);
#pragma unroll
		for (int e = 0; e < VD; ++e) {
			lamn[e] = ld2s(a.lam + pbn + off[e]);
			if (HAS_RT) rtn[HAS_RT ? e : 0] = ld2s(a.RT + pbn + off[e]);
		}
	};
	if (PFC) issue_coef(z0);
	__syncthreads();

	for (int z = z0; z < z1; ++z) {
		// 1. psi of plane z+1 -> registers, other LDS buffer, memory
		own_psi(pp);
		to_lds(pp, lds[(z + 1) & 1]);
		if (z + 1 < z1 || (z + 1 == a.zhi + 1 && (GEN || g.has_hi) && a.zhi == g.mz)) store_psi(pp, z + 1);
		// 2. prefetch the inputs of plane z+2
		if (z + 1 < z1) issue(z + 2);

		// 3. stencil on plane z
		const long pb = plane(z);
		const double *buf = lds[z & 1];
		// plane 0 is a domain face or counted by the lower slab; with general
		// boundaries the face plane mz is left to k_surf as well
		const bool cnt = (z >= 1) && (!GEN || z <= g.mz - 1 || g.has_hi);
		double2 lamc[VD], rtc[HAS_RT ? VD : 1];
		if (PFC) {
#pragma unroll
			for (int e = 0; e < VD; ++e) {
				lamc[e] = lamn[e];
				if (HAS_RT) rtc[HAS_RT ? e : 0] = rtn[HAS_RT ? e : 0];
			}
			if (z + 1 < z1) issue_coef(z + 1);
		}
#pragma unroll
		for (int e = 0; e < VD; ++e) {
			const long p = pb + off[e];
			const int li = lidx[e];
			Sten s0, s1;
			const double2 lam = PFC ? lamc[e] : ld2s(a.lam + p);
			const double xl = buf[li - 1], xr = buf[li + 2];
			const double2 ym = ld2(buf + li - LW), yp = ld2(buf + li + LW);
			s0.pc = pc[e].x; s1.pc = pc[e].y;
			s0.xm = xl;      s0.xp = pc[e].y;
			s1.xm = pc[e].x; s1.xp = xr;
			s0.ym = ym.x; s1.ym = ym.y; s0.yp = yp.x; s1.yp = yp.y;
			s0.zm = pm[e].x; s1.zm = pm[e].y; s0.zp = pp[e].x; s1.zp = pp[e].y;
			s0.lam = lam.x; s1.lam = lam.y;
			if (VAR) {
				const double2 dxx = ld2(a.Dxx + p), dyy = ld2(a.Dyy + p);
				const double dxl = a.Dxx[p - 1], dxr = a.Dxx[p + 2];
				const double2 dym = ld2(a.Dyy + p - g.px), dyp = ld2(a.Dyy + p + g.px);
				dzp[e] = ld2(a.Dzz + plane(z + 1) + off[e]);
				s0.dxx = dxx.x; s1.dxx = dxx.y; s0.dyy = dyy.x; s1.dyy = dyy.y;
				s0.dzz = dzc[e].x; s1.dzz = dzc[e].y;
				s0.dxx_m = dxl;   s0.dxx_p = dxx.y;
				s1.dxx_m = dxx.x; s1.dxx_p = dxr;
				s0.dyy_m = dym.x; s1.dyy_m = dym.y; s0.dyy_p = dyp.x; s1.dyy_p = dyp.y;
				s0.dzz_m = dzm[e].x; s1.dzz_m = dzm[e].y; s0.dzz_p = dzp[e].x; s1.dzz_p = dzp[e].y;
				if (OFFD) {
					const double2 dxy = ld2(a.Dxy + p);
					const double xyl = a.Dxy[p - 1], xyr = a.Dxy[p + 2];
					const double2 xym = ld2(a.Dxy + p - g.px), xyp = ld2(a.Dxy + p + g.px);
					s0.dxy = dxy.x; s1.dxy = dxy.y;
					s0.dxy_xm = xyl;   s0.dxy_xp = dxy.y;
					s1.dxy_xm = dxy.x; s1.dxy_xp = xyr;
					s0.dxy_ym = xym.x; s1.dxy_ym = xym.y; s0.dxy_yp = xyp.x; s1.dxy_yp = xyp.y;
					// corner psi values from the LDS ring
					const double ml = buf[li - LW - 1], mr = buf[li - LW + 2];
					const double pl = buf[li + LW - 1], pr = buf[li + LW + 2];
					s0.pmm = ml;   s0.ppm = ym.y; s0.pmp = pl;   s0.ppp = yp.y;
					s1.pmm = ym.x; s1.ppm = mr;   s1.pmp = yp.x; s1.ppp = pr;
				}
			}
			double2 out;
			out.x = apply_op<VAR, OFFD, EXACT>(s0, a.op);
			out.y = apply_op<VAR, OFFD, EXACT>(s1, a.op);
			if (KIND == KF_RES) {
				// residual = rhs - A*phi (Krylov.C:259, 325)
				const double2 rh = ld2s(a.RHS + p);
				out.x = rh.x - out.x;
				out.y = rh.y - out.y;
			}
			const bool m0 = mrow[e] && mxx, m1 = mrow[e] && mxy;
			const bool o0 = orow[e] && oxx, o1 = orow[e] && oxy;
			if (o0 && o1) {
				st2s(a.O + p, out);
				if (KIND == KF_RES) st2s(a.O2 + p, out);
			} else if (o0) {
				a.O[p] = out.x;
				if (KIND == KF_RES) a.O2[p] = out.x;
			} else if (o1) {
				a.O[p + 1] = out.y;
				if (KIND == KF_RES) a.O2[p + 1] = out.y;
			}
			// dot products over counted interior points
			const bool c0 = m0 && cnt, c1 = m1 && cnt;
			auto add = [&](double &s, double vx, double vy) {
				s += c0 ? vx : 0.;
				s += c1 ? vy : 0.;
			};
			if (KIND == KF_A0 || KIND == KF_A1 || KIND == KF_B0) {
				const double2 rt = PFC ? rtc[HAS_RT ? e : 0] : ld2s(a.RT + p);
				add(acc[0], out.x * rt.x, out.y * rt.y);
			} else if (KIND == KF_B0L) {
				add(acc[0], pc[e].x * out.x, pc[e].y * out.y);
				add(acc[1], out.x * out.x, out.y * out.y);
			} else if (KIND == KF_B1L) {
				add(acc[0], pc[e].x * pc[e].x, pc[e].y * pc[e].y);
				add(acc[1], out.x * pc[e].x, out.y * pc[e].y);
			} else if (KIND == KF_RES) {
				add(acc[0], out.x * out.x, out.y * out.y);
			}
		}
		// 4. rotate the z pipeline
#pragma unroll
		for (int e = 0; e < VD; ++e) {
			pm[e] = pc[e];
			pc[e] = pp[e];
			if (VAR) { dzm[e] = dzc[e]; dzc[e] = dzp[e]; }
		}
		__syncthreads();
	}
	if (KIND != KF_SPMV) {
		KryRed r = a.red;
		reduce_and_finalize<ND>(acc, r, a.blk_off + bid);
	}
}

// Boundary-plane terms of the dot products of a general-boundary fused kernel,
// taken after the boundary treatment of its output: all points with i in {0,mx},
// j in {0,my} or k on a z face of the domain, weighted 1/2 per boundary direction
// (Krylov.C:180-233).  The threads that own the z-face points first apply the
// z-face treatment to the output (the x and y faces are done by g_bc before).
// Its blocks append their partial sums to those of k_fused and the last one
// finishes the reduction.  Item order: the z-face planes (x
// fastest), then the y-face rows of the other counted planes, then the x-face
// columns of their inner rows.
template <int KIND>
__global__ __launch_bounds__(NT) void k_surf(const KryFused a)
{
	constexpr int ND = KindTraits<KIND>::ND;
	const KryGeo &g = a.g;
	if (a.S->done) return;
	double acc[ND];
#pragma unroll
	for (int k = 0; k < ND; ++k) acc[k] = 0.;
	// Counted points per direction (interfaces count once, see KryGeo::nb) and the
	// faces of the global domain among this sub-domain's faces.  Item counts fit 32 bits.
	const int cx0 = 0, cx1 = g.nb[1] ? g.mx - 1 : g.mx;
	const int cy0 = g.nb[2] ? 1 : 0;
	const unsigned nx1 = cx1 - cx0 + 1;
	// the z-face family walks ALL points of the face (each holder of an interface row
	// applies the z-face treatment to its own copy) and gives the uncounted ones weight 0
	const unsigned nxa = g.mx + 1, nya = g.my + 1;
	const unsigned nxy = nxa * nya;
	const unsigned nzf = (g.has_lo ? 0 : 1) + (g.has_hi ? 0 : 1);
	const unsigned nyf = (g.nb[2] ? 0 : 1) + (g.nb[3] ? 0 : 1);
	const unsigned nxf = (g.nb[0] ? 0 : 1) + (g.nb[1] ? 0 : 1);
	const int k1 = 1, k2 = g.has_hi ? g.mz : g.mz - 1, nk = k2 - k1 + 1;
	const int j1 = 1, j2 = g.nb[3] ? g.my : g.my - 1;
	const unsigned nym = j2 - j1 + 1;
	const unsigned tz = nzf * nxy, ty = nyf * nk * nx1, tx = nxf * nk * nym;
	const unsigned total = tz + ty + tx;
	for (unsigned t = blockIdx.x * NT + threadIdx.x; t < total; t += gridDim.x * NT) {
		int i, j, k;
		bool bz = false;
		if (t < tz) {
			const unsigned f = t / nxy;
			const unsigned rem = t - f * nxy;
			j = (int)(rem / nxa);
			i = (int)(rem - (unsigned)j * nxa);
			k = (f == 0 && !g.has_lo) ? 0 : g.mz;
			bz = true;
		} else if (t < tz + ty) {
			const unsigned rem = t - tz;
			const unsigned r2 = rem / nx1;
			i = cx0 + (int)(rem - r2 * nx1);
			const unsigned side = r2 % nyf;
			j = (side == 0 && !g.nb[2]) ? 0 : g.my;
			k = k1 + (int)(r2 / nyf);
		} else {
			const unsigned rem = t - tz - ty;
			const unsigned r2 = rem / nym;
			j = j1 + (int)(rem - r2 * nym);
			const unsigned side = r2 % nxf;
			i = (side == 0 && !g.nb[0]) ? 0 : g.mx;
			k = k1 + (int)(r2 / nxf);
		}
		// weight 1/2 per face of the global domain the point lies on
		const bool bx = (i == 0 && !g.nb[0]) || (i == g.mx && !g.nb[1]);
		const bool by = (j == 0 && !g.nb[2]) || (j == g.my && !g.nb[3]);
		const bool counted = (i <= cx1 && j >= cy0);
		const double wgt = counted ? (bx ? 0.5 : 1.) * (by ? 0.5 : 1.) * (bz ? 0.5 : 1.) : 0.;
		const long p = kry_idx(g, i, j, k);
		double o;
		if (bz) {
			// the z-face treatment of the output (BoundaryHandler.C:528-566), done here by
			// the one thread that owns the column; the x and y faces were treated before
			const long s = g.ps;
			double *c = a.O + p;
			const int type = (k == 0) ? g.bc[4] : g.bc[5];
			if (type == 0) {
				o = 0.;
				c[0] = 0.;
				c[k == 0 ? -s : s] = 0.;
			} else if (k == 0) {
				c[-s] = 2. * c[0] - c[s];        // ghost first, from the old plane 0
				o = 2. * c[s] - c[2 * s];
				c[0] = o;
			} else {
				o = 2. * c[-s] - c[-2 * s];
				c[0] = o;
				c[s] = 2. * o - c[-s];
			}
		} else {
			o = a.O[p];
		}
		if (KIND == KF_A0 || KIND == KF_A1 || KIND == KF_B0) {
			acc[0] += wgt * (o * a.RT[p]);
		} else if (KIND == KF_B0L) {
			acc[0] += wgt * (a.Xout[p] * o);
			acc[ND > 1 ? 1 : 0] += wgt * (o * o);
		} else if (KIND == KF_B1L) {
			const double r1 = a.Xout[p];
			acc[0] += wgt * (r1 * r1);
			acc[ND > 1 ? 1 : 0] += wgt * (o * r1);
		}
	}
	KryRed r = a.red;
	reduce_and_finalize<ND>(acc, r, a.blk_off + blockIdx.x);
}

// ---------------------------------------------------------------------------
// streaming kernels of the MR part (no stencil): one block per plane segment
// ---------------------------------------------------------------------------

// l = 2 modified Gram-Schmidt step: r2' = r2 - tau12*r1 ; (r2',r2'), (r0'',r2')
// (solveLin_BICGStab.C:420-428, jj = 2) with r0'' = r0' - alpha*u1 (:392-394), and the
// dot (r0'', r1) of jj = 1 (:428), whose operands are all here.
// Neither r2' nor r0'' is written back: k_final recomputes both with the same two
// roundings, and r2 is overwritten by the next pass before any other read.
// v[0] = r0', v[1] = r1, v[2] = r2, v[3] = u1.
// In-plane weights of the two values of a double2 at plane offset q (counted from
// column QOFF of row 0): 0 off the points 0..m, 1/2 per boundary direction.
// corner[d] tells whether the value sits on an x and a y boundary.
__device__ inline void plane_weights(const KryGeo &g, long q, double (&wxy)[2], bool (&corner)[2])
{
#pragma unroll
	for (int d = 0; d < 2; ++d) {
		const long qq = q + d + QOFF;
		const int row = (int)(qq / g.px);
		const int i = (int)(qq - (long)row * g.px) - 2, j = row - 1;
		// counted points: interfaces once (KryGeo::nb); 1/2 per face of the global domain
		const bool in = (i >= 0 && i <= (g.nb[1] ? g.mx - 1 : g.mx) && j >= (g.nb[2] ? 1 : 0) && j <= g.my);
		const bool bx = (i == 0 && !g.nb[0]) || (i == g.mx && !g.nb[1]);
		const bool by = (j == 0 && !g.nb[2]) || (j == g.my && !g.nb[3]);
		wxy[d] = in ? (bx ? 0.5 : 1.) * (by ? 0.5 : 1.) : 0.;
		corner[d] = in && bx && by;
	}
}

// Which of a thread's double2 elements of a plane segment are worth touching.  General
// boundaries: every stored cell of the plane (ghost cells are live data).  Dirichlet faces:
// the vectors are zero on the faces and behind them and stay so under every update, and
// phi + c*0 is phi, so only pairs holding an interior point (rows 1..my-1, columns 0..mx-1)
// are loaded, updated and stored: the skipped pairs contributed +0 to every sum, the sums
// keep their bits, and the row padding, ghost rows and face columns cost no memory traffic.
template <bool GEN>
__device__ inline bool seg_live(const KryGeo &g, long q)
{
	if (GEN) return q < g.ps;
	const long qq = q + QOFF;
	const int row = (int)(qq / g.px);
	const int i = (int)(qq - (long)row * g.px) - 2, j = row - 1;
	return j >= 1 && j <= g.my - 1 && i >= 0 && i <= g.mx - 1;
}

template <bool GEN>
__global__ __launch_bounds__(NT) void k_mgs2(const KryStream a)
{
	KryState *S = a.S;
	if (S->done) return;
	constexpr int UN = SEG2 / NT;
	const double ntau = -S->tau[1][2];
	const double nalpha = -S->alpha;
	double acc[3] = {0., 0., 0.};
	// rev: the blocks take the plane segments in reverse order (the sums do not notice: see reduce_and_finalize)
	const int bx = a.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
	const int by = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
	const long q0 = 2 * ((long)bx * SEG2 + threadIdx.x);
	double wxy[GEN ? UN : 1][2];
	if (GEN) {
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			bool c[2];
			plane_weights(a.g, q0 + 2L * i * NT, wxy[GEN ? i : 0], c);
		}
	}
	bool live[UN];
#pragma unroll
	for (int i = 0; i < UN; ++i) live[i] = seg_live<GEN>(a.g, q0 + 2L * i * NT);
	for (int pl = by; pl < a.nplanes; pl += gridDim.y) {
		const int z = a.zfirst + pl;
		const bool cnt = (z >= a.cnt_lo && z <= a.cnt_hi);
		const long pb = (long)(z + 1) * a.g.ps + QOFF;
		// all loads first
		double2 r2[UN], r1[UN], r0[UN], u1[UN];
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			const long p = pb + q0 + 2L * i * NT;
			if (live[i]) {
				r2[i] = ld2s(a.v[2] + p); r1[i] = ld2s(a.v[1] + p); r0[i] = ld2s(a.v[0] + p); u1[i] = ld2s(a.v[3] + p);
			}
		}
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			if (live[i]) {
				r2[i] = axpy2(r2[i], r1[i], ntau);   // kept in registers only: k_final recomputes it
				r0[i] = axpy2(r0[i], u1[i], nalpha);  // r0'' = r0' - alpha*u1, likewise
				if (GEN) {
					if (cnt) {
						const bool bz = (z == 0 && !a.g.has_lo) || (z == a.g.mz && !a.g.has_hi);
						const double wz = bz ? 0.5 : 1.;
						const double w0 = wxy[GEN ? i : 0][0] * wz, w1 = wxy[GEN ? i : 0][1] * wz;
						acc[0] += w0 * (r2[i].x * r2[i].x); acc[0] += w1 * (r2[i].y * r2[i].y);
						acc[1] += w0 * (r0[i].x * r2[i].x); acc[1] += w1 * (r0[i].y * r2[i].y);
						acc[2] += w0 * (r0[i].x * r1[i].x); acc[2] += w1 * (r0[i].y * r1[i].y);
					}
				} else if (cnt) {
					acc[0] += r2[i].x * r2[i].x; acc[0] += r2[i].y * r2[i].y;
					acc[1] += r0[i].x * r2[i].x; acc[1] += r0[i].y * r2[i].y;
					acc[2] += r0[i].x * r1[i].x; acc[2] += r0[i].y * r1[i].y;
				}
			}
		}
	}
	reduce_and_finalize<3>(acc, a.red, bx + (long)gridDim.x * by);
}

// End of an outer pass (solveLin_BICGStab.C:403, 450-464):
//   phi += alpha*u0 (deferred from the last BiCG step); phi += gamma1*r0;
//   r0 -= gammap_L*r_L; then for jj = 1..L-1: phi += gammapp_jj*r_jj; r0 -= gammap_jj*r_jj;
//   dots |r0|^2 and (r0, r~).
// l = 1: the u0 update of these lines (u0 -= gamma_1*u_1) is left to the next pass's
// first stencil kernel, which reads u0 anyway.
// l = 2 also carries the pointwise statements of the second BiCG step that no
// stencil needs (:403 of jj = 0, :356-362 with ii = 0 and :392-394 with ii = 0 of
// jj = 1): phi += alpha0*u0; u0 = r0 - beta*u0; r0 -= alpha*u1; then :403 of jj = 1;
// having u0 and u1 in registers for that, it also closes u0 (-= gamma_2*u2, -= gamma_1*u1).
// v[0..L] = r0..rL, v[L+1] = u0, v[L+2] = r~, (l = 2) v[L+3] = u1, v[L+4] = u2;
// w[0] = phi, w[1] = r0, (l = 2) w[2] = u0.
template <int L, bool GEN>
__global__ __launch_bounds__(NT) void k_final(const KryStream a)
{
	KryState *S = a.S;
	if (S->done) return;
	const double alpha = S->alpha;
	const double alpha0 = S->alpha0, nbeta = -S->beta;
	const double ntau12 = (L == 2) ? -S->tau[1][2] : 0.;
	double g1[L + 1], gp[L + 1], gpp[L + 1];
#pragma unroll
	for (int j = 1; j <= L; ++j) { g1[j] = S->gamma[j]; gp[j] = S->gammap[j]; gpp[j] = (j < L) ? S->gammapp[j] : 0.; }
	constexpr int UN = SEG2 / NT;
	double acc[2] = {0., 0.};
	// rev: the blocks take the plane segments in reverse order (the sums do not notice: see reduce_and_finalize)
	const int bx = a.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
	const int by = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
	const long q0 = 2 * ((long)bx * SEG2 + threadIdx.x);
	double wxy[GEN ? UN : 1][2];
	bool wcorner[GEN ? UN : 1][2];
	if (GEN) {
#pragma unroll
		for (int i = 0; i < UN; ++i) plane_weights(a.g, q0 + 2L * i * NT, wxy[GEN ? i : 0], wcorner[GEN ? i : 0]);
	}
	bool live[UN];
#pragma unroll
	for (int i = 0; i < UN; ++i) live[i] = seg_live<GEN>(a.g, q0 + 2L * i * NT);
	for (int pl = by; pl < a.nplanes; pl += gridDim.y) {
		const int z = a.zfirst + pl;
		const bool cnt = (z >= a.cnt_lo && z <= a.cnt_hi);
		const long pb = (long)(z + 1) * a.g.ps + QOFF;
		double2 r[UN][L + 1], u0[UN], ph[UN], rt[UN], u1[L == 2 ? UN : 1], u2[L == 2 ? UN : 1];
		// all loads first
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			const long p = pb + q0 + 2L * i * NT;
			if (live[i]) {
#pragma unroll
				for (int j = 0; j <= L; ++j) r[i][j] = ld2s(a.v[j] + p);
				u0[i] = ld2s(a.v[L + 1] + p);
				ph[i] = ld2s(a.w[0] + p);
				rt[i] = ld2s(a.v[L + 2] + p);
				if (L == 2) { u1[L == 2 ? i : 0] = ld2s(a.v[L + 3] + p); u2[L == 2 ? i : 0] = ld2s(a.v[L + 4] + p); }
			}
		}
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			const long q = q0 + 2L * i * NT;
			if (live[i]) {
				const long p = pb + q;
				if (L == 2) r[i][2] = axpy2(r[i][2], r[i][1], ntau12);   // the Gram-Schmidt step of k_mgs2
				double2 phv = ph[i];
				if (L == 2) {
					phv = axpy2(phv, u0[i], alpha0);                       // phi += alpha0*u0
					u0[i] = axpy2(r[i][0], u0[i], nbeta);                  // u0 = r0' - beta*u0
					r[i][0] = axpy2(r[i][0], u1[L == 2 ? i : 0], -alpha);  // r0'' = r0' - alpha*u1
					// u0 -= gamma_2*u2 ; u0 -= gamma_1*u1 (:452, 458): u1 is here anyway, u2 costs one read
					double2 un = axpy2(u0[i], u2[L == 2 ? i : 0], -g1[2 <= L ? 2 : L]);
					un = axpy2(un, u1[L == 2 ? i : 0], -g1[1]);
					st2s(a.w[2] + p, un);
				}
				phv = axpy2(phv, u0[i], alpha);
				phv = axpy2(phv, r[i][0], g1[1]);
				double2 r0 = axpy2(r[i][0], r[i][L], -gp[L]);
#pragma unroll
				for (int j = 1; j < L; ++j) {
					phv = axpy2(phv, r[i][j], gpp[j]);
					r0 = axpy2(r0, r[i][j], -gp[j]);
				}
				st2s(a.w[0] + p, phv);
				st2s(a.w[1] + p, r0);
				if (GEN && L == 2) {
					// r2' = r2 - tau12*r1 lives in registers only, except on the ghost rows j = -1, my+1 and the ghost
					// planes: the face treatment of the next r2 = A r1 reads them from this array
					// (BoundaryHandler.C:420-421, 479-480), and the reference's array holds r2' there
					// (solveLin_BICGStab.C:424 acts on all cells)
					const int row = (int)((q + QOFF) / a.g.px);
					if (row == 0 || row == a.g.my + 2 || z == -1 || z == a.g.mz + 1) st2(a.w[3] + p, r[i][2]);
				}
				if (GEN) {
					if (cnt) {
						// |r0|^2 with the norm's weights (corners 1/4, Krylov.C:136-143),
						// (r0, r~) with the dot product's (corners 1/8, Krylov.C:226-233)
						const bool bz = (z == 0 && !a.g.has_lo) || (z == a.g.mz && !a.g.has_hi);
						const double wz = bz ? 0.5 : 1.;
						const double w0 = wxy[GEN ? i : 0][0] * wz, w1 = wxy[GEN ? i : 0][1] * wz;
						const double n0 = (bz && wcorner[GEN ? i : 0][0]) ? 0.25 : w0;
						const double n1 = (bz && wcorner[GEN ? i : 0][1]) ? 0.25 : w1;
						acc[0] += n0 * (r0.x * r0.x);   acc[0] += n1 * (r0.y * r0.y);
						acc[1] += w0 * (r0.x * rt[i].x); acc[1] += w1 * (r0.y * rt[i].y);
					}
				} else if (cnt) {
					acc[0] += r0.x * r0.x;   acc[0] += r0.y * r0.y;
					acc[1] += r0.x * rt[i].x; acc[1] += r0.y * rt[i].y;
				}
			}
		}
	}
	reduce_and_finalize<2>(acc, a.red, bx + (long)gridDim.x * by);
}

// ---------------------------------------------------------------------------
// general l (l > 2): streaming kernels with the shape of k_mgs2 / k_final (2048-double plane
// segments x plane groups, per-thread sums in the same order), scalars from the device state
// ---------------------------------------------------------------------------

// Pointwise statements of BiCG step jj > 0 that no stencil kernel carries, each array read and written once:
//   phi += alpha_(jj-1)*u_0      (solveLin_BICGStab.C:403 of the previous step; u_0 is still that step's)
//   u_ii = r_ii - beta*u_ii      (:356-362, ii < jj; ii = jj is done by the A kernel)
//   r_ii -= alpha*u_(ii+1)       (:392-394, ii < jj; ii = jj is done by the B kernel)
// The chain runs upwards in ii with the updated u_(ii+1) formed one step ahead.
__global__ __launch_bounds__(NT) void k_bicgn(const KryStream a, const KryPointN f)
{
	const KryState *S = a.S;
	if (S->done) return;
	constexpr int UN = SEG2 / NT;
	const int jj = f.jj;
	const double nbeta = -S->beta, nalpha = -S->alpha, aprev = S->alpha0;
	const int bx = a.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
	const int by = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
	const long q0 = 2 * ((long)bx * SEG2 + threadIdx.x);
	bool live[UN];
#pragma unroll
	for (int i = 0; i < UN; ++i) live[i] = a.gen ? seg_live<true>(a.g, q0 + 2L * i * NT) : seg_live<false>(a.g, q0 + 2L * i * NT);
	for (int pl = by; pl < a.nplanes; pl += gridDim.y) {
		const long pb = (long)(a.zfirst + pl + 1) * a.g.ps + QOFF;
		double2 un[UN], ro[UN];
		{
			double2 ph[UN], uo[UN];
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				const long p = pb + q0 + 2L * i * NT;
				if (live[i]) {
					ph[i] = ld2s(a.w[0] + p);
					uo[i] = ld2s(f.u[0] + p);
					ro[i] = ld2s(f.r[0] + p);
				}
			}
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				const long q = q0 + 2L * i * NT;
				ph[i] = axpy2(ph[i], uo[i], aprev);
				un[i] = axpy2(ro[i], uo[i], nbeta);
				if (live[i]) st2s(a.w[0] + pb + q, ph[i]);
			}
		}
		for (int ii = 0; ii < jj; ++ii) {
			const bool more = ii + 1 < jj;
			double2 nx[UN], rn[UN];
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				const long p = pb + q0 + 2L * i * NT;
				if (live[i]) {
					nx[i] = ld2s(f.u[ii + 1] + p);
					if (more) rn[i] = ld2s(f.r[ii + 1] + p);
				}
			}
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				const long q = q0 + 2L * i * NT;
				if (more) nx[i] = axpy2(rn[i], nx[i], nbeta);          // the updated u_(ii+1)
				const double2 rnew = axpy2(ro[i], nx[i], nalpha);
				if (live[i]) {
					st2s(f.u[ii] + pb + q, un[i]);
					st2s(f.r[ii] + pb + q, rnew);
				}
				un[i] = nx[i];
				if (more) ro[i] = rn[i];
			}
		}
	}
}

// One step of the modified Gram-Schmidt part (solveLin_BICGStab.C:419-428) for r_jj:
// first the update the previous step's dot product made possible, r_jj -= tau[iip][jj]*r_iip
// (iip = 0: none pending), then either the next dot (r_jj, r_ii), or -- last step of this jj --
// sigma_jj = (r_jj, r_jj) and (r_0, r_jj).
template <bool GEN>
__global__ __launch_bounds__(NT) void k_mgsn(const KryStream a, int jj, int iip, int last)
{
	KryState *S = a.S;
	if (S->done) return;
	constexpr int UN = SEG2 / NT;
	const double ntau = iip > 0 ? -S->tau[iip][jj] : 0.;
	double acc[2] = {0., 0.};
	// rev: the blocks take the plane segments in reverse order (the sums do not notice: see reduce_and_finalize)
	const int bx = a.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
	const int by = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
	const long q0 = 2 * ((long)bx * SEG2 + threadIdx.x);
	double wxy[GEN ? UN : 1][2];
	if (GEN) {
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			bool c[2];
			plane_weights(a.g, q0 + 2L * i * NT, wxy[GEN ? i : 0], c);
		}
	}
	bool live[UN];
#pragma unroll
	for (int i = 0; i < UN; ++i) live[i] = seg_live<GEN>(a.g, q0 + 2L * i * NT);
	for (int pl = by; pl < a.nplanes; pl += gridDim.y) {
		const int z = a.zfirst + pl;
		const bool cnt = (z >= a.cnt_lo && z <= a.cnt_hi);
		const long pb = (long)(z + 1) * a.g.ps + QOFF;
		double2 rj[UN], rp[UN], rk[UN];
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			const long p = pb + q0 + 2L * i * NT;
			if (live[i]) {
				rj[i] = ld2s(a.v[0] + p);
				if (iip > 0) rp[i] = ld2s(a.v[1] + p);
				rk[i] = ld2s(a.v[2] + p);
			}
		}
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			const long q = q0 + 2L * i * NT;
			if (live[i]) {
				if (iip > 0) {
					rj[i] = axpy2(rj[i], rp[i], ntau);
					st2s(a.w[0] + pb + q, rj[i]);
				}
				if (cnt) {
					double w0 = 1., w1 = 1.;
					if (GEN) {
						const bool bz = (z == 0 && !a.g.has_lo) || (z == a.g.mz && !a.g.has_hi);
						const double wz = bz ? 0.5 : 1.;
						w0 = wxy[GEN ? i : 0][0] * wz;
						w1 = wxy[GEN ? i : 0][1] * wz;
					}
					if (last) {
						if (GEN) {
							acc[0] += w0 * (rj[i].x * rj[i].x); acc[0] += w1 * (rj[i].y * rj[i].y);
							acc[1] += w0 * (rk[i].x * rj[i].x); acc[1] += w1 * (rk[i].y * rj[i].y);
						} else {
							acc[0] += rj[i].x * rj[i].x; acc[0] += rj[i].y * rj[i].y;
							acc[1] += rk[i].x * rj[i].x; acc[1] += rk[i].y * rj[i].y;
						}
					} else if (GEN) {
						acc[0] += w0 * (rj[i].x * rk[i].x); acc[0] += w1 * (rj[i].y * rk[i].y);
					} else {
						acc[0] += rj[i].x * rk[i].x; acc[0] += rj[i].y * rk[i].y;
					}
				}
			}
		}
	}
	reduce_and_finalize<2>(acc, a.red, bx + (long)gridDim.x * by);
}

// End of an outer pass for any l (solveLin_BICGStab.C:450-464), statement by statement:
//   phi += alpha*u_0 (:403 of the last BiCG step); phi += gamma_1*r_0; r_0 -= gammap_L*r_L; u_0 -= gamma_L*u_L;
//   for jj = 1..L-1: u_0 -= gamma_jj*u_jj; phi += gammapp_jj*r_jj; r_0 -= gammap_jj*r_jj;
//   |r_0|^2 (norm weights) and (r_0, r~).
template <bool GEN>
__global__ __launch_bounds__(NT) void k_finaln(const KryStream a, const KryFinalN f)
{
	KryState *S = a.S;
	if (S->done) return;
	constexpr int UN = SEG2 / NT;
	const int L = f.L;
	double acc[2] = {0., 0.};
	// rev: the blocks take the plane segments in reverse order (the sums do not notice: see reduce_and_finalize)
	const int bx = a.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
	const int by = a.rev ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
	const long q0 = 2 * ((long)bx * SEG2 + threadIdx.x);
	double wxy[GEN ? UN : 1][2];
	bool wcorner[GEN ? UN : 1][2];
	if (GEN) {
#pragma unroll
		for (int i = 0; i < UN; ++i) plane_weights(a.g, q0 + 2L * i * NT, wxy[GEN ? i : 0], wcorner[GEN ? i : 0]);
	}
	bool live[UN];
#pragma unroll
	for (int i = 0; i < UN; ++i) live[i] = seg_live<GEN>(a.g, q0 + 2L * i * NT);
	for (int pl = by; pl < a.nplanes; pl += gridDim.y) {
		const int z = a.zfirst + pl;
		const bool cnt = (z >= a.cnt_lo && z <= a.cnt_hi);
		const long pb = (long)(z + 1) * a.g.ps + QOFF;
		double2 ph[UN], r0[UN], u0[UN], rt[UN];
		{
			const double alpha = S->alpha, g1 = S->gamma[1], gpL = S->gammap[L], gL = S->gamma[L];
			double2 rl[UN], ul[UN];
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				const long p = pb + q0 + 2L * i * NT;
				if (live[i]) {
					ph[i] = ld2s(a.w[0] + p);
					r0[i] = ld2s(f.r[0] + p);
					u0[i] = ld2s(f.u[0] + p);
					rt[i] = ld2s(a.v[0] + p);
					rl[i] = ld2s(f.r[L] + p);
					ul[i] = ld2s(f.u[L] + p);
				}
			}
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				ph[i] = axpy2(ph[i], u0[i], alpha);
				ph[i] = axpy2(ph[i], r0[i], g1);
				r0[i] = axpy2(r0[i], rl[i], -gpL);
				u0[i] = axpy2(u0[i], ul[i], -gL);
			}
		}
		for (int j = 1; j < L; ++j) {
			const double gj = S->gamma[j], gpj = S->gammap[j], gppj = S->gammapp[j];
			double2 rj[UN], uj[UN];
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				const long p = pb + q0 + 2L * i * NT;
				if (live[i]) {
					rj[i] = ld2s(f.r[j] + p);
					uj[i] = ld2s(f.u[j] + p);
				}
			}
#pragma unroll
			for (int i = 0; i < UN; ++i) {
				u0[i] = axpy2(u0[i], uj[i], -gj);
				ph[i] = axpy2(ph[i], rj[i], gppj);
				r0[i] = axpy2(r0[i], rj[i], -gpj);
			}
		}
#pragma unroll
		for (int i = 0; i < UN; ++i) {
			const long q = q0 + 2L * i * NT;
			if (live[i]) {
				const long p = pb + q;
				st2s(a.w[0] + p, ph[i]);
				st2s(f.r0 + p, r0[i]);
				st2s(f.u0 + p, u0[i]);
				if (GEN) {
					if (cnt) {
						const bool bz = (z == 0 && !a.g.has_lo) || (z == a.g.mz && !a.g.has_hi);
						const double wz = bz ? 0.5 : 1.;
						const double w0 = wxy[GEN ? i : 0][0] * wz, w1 = wxy[GEN ? i : 0][1] * wz;
						const double n0 = (bz && wcorner[GEN ? i : 0][0]) ? 0.25 : w0;
						const double n1 = (bz && wcorner[GEN ? i : 0][1]) ? 0.25 : w1;
						acc[0] += n0 * (r0[i].x * r0[i].x);   acc[0] += n1 * (r0[i].y * r0[i].y);
						acc[1] += w0 * (r0[i].x * rt[i].x); acc[1] += w1 * (r0[i].y * rt[i].y);
					}
				} else if (cnt) {
					acc[0] += r0[i].x * r0[i].x;   acc[0] += r0[i].y * r0[i].y;
					acc[1] += r0[i].x * rt[i].x; acc[1] += r0[i].y * rt[i].y;
				}
			}
		}
	}
	reduce_and_finalize<2>(acc, a.red, bx + (long)gridDim.x * by);
}

// ---------------------------------------------------------------------------
// generic kernels: one per primitive, any face type
// ---------------------------------------------------------------------------

// dst = a + b*c over the whole stored box (NumMatrix whole-array operators)
__global__ __launch_bounds__(NT) void g_axpy(double *dst, const double *x, const double *y, double c, long n2)
{
	for (long q = (long)blockIdx.x * NT + threadIdx.x; q < n2; q += (long)NT * gridDim.x)
		st2(dst + 2 * q, axpy2(ld2(x + 2 * q), ld2(y + 2 * q), c));
}

// Host layout (rows of `w` doubles, no padding) <-> device layout (row pitch px,
// first value of a row in column 1).  One wave per row.
__global__ __launch_bounds__(NT) void g_repack(double *dev, double *packed, int w, long rows, int px, int to_device)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (long r = (long)blockIdx.x * NW + wv; r < rows; r += (long)gridDim.x * NW) {
		double *d = dev + r * px + 1;
		double *p = packed + r * w;
		if (to_device) for (int i = lane; i < w; i += 64) d[i] = p[i];
		else for (int i = lane; i < w; i += 64) p[i] = d[i];
	}
}

// One x plane (dir 0: i = idx, all j, k of the stored box) or one y row set (dir 1:
// j = idx, all i, k) between a field and a packed buffer, for the halo exchange with
// x and y neighbours (BoundaryHandler.C:579-1045 packs the same full tangential box).
// dir 0: buf[(k+1)*(my+3) + (j+1)];  dir 1: buf[(k+1)*(mx+3) + (i+1)].
__global__ __launch_bounds__(NT) void g_face(double *field, double *buf, KryGeo g, int dir, int idx, int to_buf)
{
	const long na = (dir == 0 ? g.my : g.mx) + 3, nb = g.mz + 3;
	for (long t = (long)blockIdx.x * NT + threadIdx.x; t < na * nb; t += (long)gridDim.x * NT) {
		const int a = (int)(t % na) - 1, k = (int)(t / na) - 1;
		const long p = dir == 0 ? kry_idx(g, idx, a, k) : kry_idx(g, a, idx, k);
		if (to_buf) buf[t] = field[p];
		else field[p] = buf[t];
	}
}

// Ghost-plane exchange between processes through mapped device memory.
// k_halo_push copies this rank's planes into the neighbours' landing slots (stores that
// travel over the GPU interconnect) and, when the whole grid has done so, raises the
// neighbours' sequence words; k_halo_pull waits for this rank's own sequence words and
// copies the landed planes into the ghost cells.  A side without neighbour has null
// pointers.  The sequence words only grow, and a slot is reused every second exchange.
struct KryHaloSide {
	const double *src;              // push: own plane            pull: own landing slot
	double *dst;                    // push: neighbour's slot     pull: own ghost plane
	unsigned long long *flag;       // push: neighbour's word     pull: own word
};

__device__ inline void halo_copy(const KryHaloSide &sd, size_t count)
{
	const size_t start = (size_t)blockIdx.x * NT + threadIdx.x, stride = (size_t)gridDim.x * NT;
	const bool vec = ((count & 1) == 0) && (((size_t)sd.src | (size_t)sd.dst) & 15) == 0;
	if (vec) {
		for (size_t q = start; q < count / 2; q += stride) st2s(sd.dst + 2 * q, ld2s(sd.src + 2 * q));
	} else {
		for (size_t q = start; q < count; q += stride) st1s(sd.dst + q, __builtin_nontemporal_load(sd.src + q));
	}
}

__global__ __launch_bounds__(NT) void k_halo_push(KryHaloSide lo, KryHaloSide hi, size_t count, unsigned long long seq,
                                                  unsigned *ticket, const KryState *S)
{
	// an earlier exchange of this solve gave up on a neighbour: nothing more is sent
	if (S && S->done && S->status < 0) return;
	if (lo.src) halo_copy(lo, count);
	if (hi.src) halo_copy(hi, count);
	__threadfence_system();          // this thread's stores have reached the neighbour's memory
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		if (t == gridDim.x - 1) {
			__hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (lo.src) __hip_atomic_store(lo.flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			if (hi.src) __hip_atomic_store(hi.flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

__global__ __launch_bounds__(NT) void k_halo_pull(KryHaloSide lo, KryHaloSide hi, size_t count, unsigned long long seq,
                                                  long long timeout, KryState *S)
{
	__shared__ int failed;
	if (threadIdx.x == 0) {
		// a solve that has already failed (an earlier exchange timed out) does not wait again:
		// every exchange still queued behind the failure returns at once
		int bad = (S->done && S->status < 0) ? 1 : 0;
		const long long t0 = wall_clock64();
		unsigned spins = 0;
		for (int side = 0; side < 2 && !bad; ++side) {
			const unsigned long long *f = side ? hi.flag : lo.flag;
			if (!(side ? hi.src : lo.src)) continue;
			while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
				__builtin_amdgcn_s_sleep(4);
				if ((++spins & 255u) == 0u && wall_clock64() - t0 > timeout) { bad = 1; break; }
			}
		}
		failed = bad;
		if (blockIdx.x == 0) S->wait_halo_ticks += wall_clock64() - t0;
	}
	__syncthreads();
	if (failed) {
		// the neighbour never delivered: stop the solve, the kernels queued behind exit at once
		if (threadIdx.x == 0 && blockIdx.x == 0) { S->status = -3; S->done = 1; }
		return;
	}
	if (lo.src) halo_copy(lo, count);
	if (hi.src) halo_copy(hi, count);
}

// One value per slab or rank, all-reduced the way the reducing kernels do it (through the
// mailboxes when S->mail is set): a check that a freshly made link carries data.
__global__ __launch_bounds__(NT) void k_link_probe(KryRed r, double v)
{
	double acc[1] = {threadIdx.x == 0 ? v : 0.};
	reduce_and_finalize<1>(acc, r);
}

// dst = a - b (Krylov.C:259)
__global__ __launch_bounds__(NT) void g_sub(double *dst, const double *x, const double *y, long n2)
{
	for (long q = (long)blockIdx.x * NT + threadIdx.x; q < n2; q += (long)NT * gridDim.x) {
		double2 a = ld2(x + 2 * q), b = ld2(y + 2 * q), r;
		r.x = a.x - b.x;
		r.y = a.y - b.y;
		st2(dst + 2 * q, r);
	}
}

// BoundaryHandler3D::do_BCs, rim = 1 (BoundaryHandler.C:416-568); one direction per launch
// (the solver runs x and y through g_bc_xy and only the z faces of the generic path through this one).
// dir 0: a thread per (j,k) in [-1..my+1] x [-1..mz+1]
// dir 1: a thread per (i,k) in [0..mx] x [-1..mz+1]
// dir 2: a thread per (i,j) in [0..mx] x [0..my]; only on faces of the global domain
__global__ void g_bc(double *d, KryGeo g, int dir)
{
	const int a = blockIdx.x * blockDim.x + threadIdx.x;
	const int b = blockIdx.y;
	if (dir == 0) {
		const int j = a - 1, k = b - 1;
		if (j > g.my + 1) return;
		double *row = d + kry_idx(g, 0, j, k);   // row[i] = value at grid index i
		if (!g.nb[0]) {
			if (g.bc[0] == 0) { row[-1] = 0.; row[0] = 0.; }
			else if (g.bc[0] == 1) row[-1] = 2. * row[0] - row[1];
		}
		const int m = g.mx;
		if (!g.nb[1]) {
			if (g.bc[1] == 0) { row[m] = 0.; row[m + 1] = 0.; }
			else if (g.bc[1] == 1) {
				row[m] = 2. * row[m - 1] - row[m - 2];
				row[m + 1] = 2. * row[m] - row[m - 1];
			}
		}
	} else if (dir == 1) {
		const int i = a, k = b - 1;
		if (i > g.mx) return;
		double *c = d + kry_idx(g, i, 0, k);     // c[j*px]
		const long s = g.px;
		if (!g.nb[2]) {
			if (g.bc[2] == 0) { c[-s] = 0.; c[0] = 0.; }
			else if (g.bc[2] == 1) c[-s] = 2. * c[0] - c[s];
		}
		const int m = g.my;
		if (!g.nb[3]) {
			if (g.bc[3] == 0) { c[m * s] = 0.; c[(m + 1) * s] = 0.; }
			else if (g.bc[3] == 1) {
				c[m * s] = 2. * c[(m - 1) * s] - c[(m - 2) * s];
				c[(m + 1) * s] = 2. * c[m * s] - c[(m - 1) * s];
			}
		}
	} else {
		const int i = a, j = b;
		if (i > g.mx) return;
		double *c = d + kry_idx(g, i, j, 0);
		const long s = g.ps;
		if (!g.has_lo) {
			if (g.bc[4] == 0) { c[-s] = 0.; c[0] = 0.; }
			else if (g.bc[4] == 1) {
				c[-s] = 2. * c[0] - c[s];        // ghost first, from the OLD plane 0
				c[0] = 2. * c[s] - c[2 * s];
			}
		}
		const int m = g.mz;
		if (!g.has_hi) {
			if (g.bc[5] == 0) { c[m * s] = 0.; c[(m + 1) * s] = 0.; }
			else if (g.bc[5] == 1) {
				c[m * s] = 2. * c[(m - 1) * s] - c[(m - 2) * s];
				c[(m + 1) * s] = 2. * c[m * s] - c[(m - 1) * s];
			}
		}
	}
}

// The x faces and then the y faces of one plane per block: both treatments stay inside the plane
// (BoundaryHandler.C:416-517), and the y treatment reads columns 0 and mx as the x treatment left
// them, so the block's threads meet at a barrier in between.
__device__ inline void bc_xy_plane(double *d, const KryGeo &g, int k, int nthreads)
{
	for (int j = (int)threadIdx.x - 1; j <= g.my + 1; j += nthreads) {
		double *row = d + kry_idx(g, 0, j, k);   // row[i] = value at grid index i
		if (!g.nb[0]) {
			if (g.bc[0] == 0) { row[-1] = 0.; row[0] = 0.; }
			else if (g.bc[0] == 1) row[-1] = 2. * row[0] - row[1];
		}
		const int m = g.mx;
		if (!g.nb[1]) {
			if (g.bc[1] == 0) { row[m] = 0.; row[m + 1] = 0.; }
			else if (g.bc[1] == 1) {
				row[m] = 2. * row[m - 1] - row[m - 2];
				row[m + 1] = 2. * row[m] - row[m - 1];
			}
		}
	}
	__threadfence_block();
	__syncthreads();
	for (int i = (int)threadIdx.x; i <= g.mx; i += nthreads) {
		double *c = d + kry_idx(g, i, 0, k);     // c[j*px]
		const long s = g.px;
		if (!g.nb[2]) {
			if (g.bc[2] == 0) { c[-s] = 0.; c[0] = 0.; }
			else if (g.bc[2] == 1) c[-s] = 2. * c[0] - c[s];
		}
		const int m = g.my;
		if (!g.nb[3]) {
			if (g.bc[3] == 0) { c[m * s] = 0.; c[(m + 1) * s] = 0.; }
			else if (g.bc[3] == 1) {
				c[m * s] = 2. * c[(m - 1) * s] - c[(m - 2) * s];
				c[(m + 1) * s] = 2. * c[m * s] - c[(m - 1) * s];
			}
		}
	}
}

__global__ __launch_bounds__(1024) void g_bc_xy(double *d, KryGeo g)
{
	bc_xy_plane(d, g, (int)blockIdx.x - 1, (int)blockDim.x);
}

// Input treatment of the constant-coefficient operator on general faces, ahead of a fused stencil
// kernel: psi = X + c*Y (the kernel's vector update) after do_BCs (Krylov.C:516) for every stored cell
// that is not strictly inside the box -- planes k <= 0 and k >= mz whole, rows j <= 0 and j >= my of the
// planes between, cells i <= 0 and i >= mx of the rows between -- one cell per thread through bc_psi,
// into Xout.  Strictly inside the box do_BCs changes nothing, and k_fused forms psi there itself.
__global__ __launch_bounds__(256) void g_psi_shell(const KryFused a, int is_a, int defer)
{
	const KryGeo &g = a.g;
	const KryState *S = a.S;
	if (S->done) return;
	PsiSrc src;
	src.X = a.X; src.Y = a.Y; src.Z1 = a.Z1; src.Z2 = a.Z2;
	src.c = is_a ? -S->beta : -S->alpha;
	src.nz = defer ? a.nz : 0;
	src.dz1 = defer ? -S->gamma[S->L] : 0.;
	src.dz2 = defer ? -S->gamma[1] : 0.;
	// (cell counts fit 32 bits: checked by the launcher)
	const unsigned w = g.mx + 3, np = w * (g.my + 3), nrow = 4 * w, nc = nrow + 4u * (g.my - 1);
	unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= 4 * np + (unsigned)(g.mz - 1) * nc) return;
	int i, j, k;
	if (t < 4 * np) {
		const unsigned q = t / np, r = t - q * np, jr = r / w;
		k = q < 2 ? (int)q - 1 : g.mz + (int)q - 2;
		j = (int)jr - 1;
		i = (int)(r - jr * w) - 1;
	} else {
		t -= 4 * np;
		const unsigned kk = t / nc;
		unsigned r = t - kk * nc;
		k = 1 + (int)kk;
		if (r < nrow) {
			const unsigned q = r / w;
			j = q < 2 ? (int)q - 1 : g.my + (int)q - 2;
			i = (int)(r - q * w) - 1;
		} else {
			r -= nrow;
			const int q = (int)(r & 3);
			j = 1 + (int)(r >> 2);
			i = q < 2 ? q - 1 : g.mx + q - 2;
		}
	}
	a.Xout[kry_idx(g, i, j, k)] = bc_psi_batched(src, g, i, j, k);
}

// g_bc_xy for the output of a general-boundary stencil kernel whose tile rows start at j = 0 (a.y_off == 0):
// the rows its tiles leave out -- j = -1 and j = row_hi .. my+1 -- first receive the vector update the
// stencil kernel applies everywhere else (psi = X + c*Y into Xout, through do_BCs with INBC; the
// reference's whole-array updates, solveLin_BICGStab.C:356-362, 392-394, act on every stored cell),
// evaluated cell by cell from memory with the same roundings; then the x and y faces of the output.
// The operator itself is hardly needed on those rows: j = -1 and my+1 are ghost rows, and its values on
// row my (the only grid row that can be left out) are overwritten by the treatment of the upper y face
// before anything reads them -- except that an extrapolating x face, treated first and over all rows
// (BoundaryHandler.C:420-460), forms the edge ghosts (-1, my), (mx, my), (mx+1, my) from the values at
// i = 0, 1 resp. mx-2, mx-1 of that row: with `op_row` those four values per plane are evaluated here,
// psi at their neighbours formed from memory like everything else in this kernel (same roundings as
// in k_fused, same apply_op).
template <bool INBC, bool VAR, bool OFFD, bool EXACT>
__global__ __launch_bounds__(1024) void g_bc_xy_rows(const KryFused a, int row_hi, int is_a, int defer, int op_row)
{
	const KryGeo &g = a.g;
	const int k = (int)blockIdx.x - 1;
	const KryState *S = a.S;
	if (!S->done) {
		PsiSrc src;
		src.X = a.X; src.Y = a.Y; src.Z1 = a.Z1; src.Z2 = a.Z2;
		src.c = is_a ? -S->beta : -S->alpha;
		src.nz = defer ? a.nz : 0;
		src.dz1 = defer ? -S->gamma[S->L] : 0.;
		src.dz2 = defer ? -S->gamma[1] : 0.;
		const int nhi = max(0, g.my + 2 - row_hi), nrows = 1 + nhi, w = g.mx + 3;
		// l = 1: the rows and planes of the output array that the operator does not write keep the old u1 (see k_fused)
		const bool gplane = (k == -1 && !g.has_lo) || (k == g.mz + 1 && !g.has_hi);
		const int hand_last = min(KRY_TXW * a.ntx - 1, g.mx + 1);
		for (int t = (int)threadIdx.x; t < nrows * w; t += (int)blockDim.x) {
			const int r = t / w, i = t - r * w - 1;
			const int j = r == 0 ? -1 : row_hi + r - 1;
			const long p = kry_idx(g, i, j, k);
			if (!INBC) a.Xout[p] = psi_raw(src, g, i, j, k);       // (INBC: g_psi_shell has written these rows)
			if (defer && src.nz >= 1 && i >= 0 && i <= hand_last && (gplane || j == -1 || j == g.my + 1)) a.O[p] = a.Z1[p];
		}
		if (op_row && k >= 0 && k <= g.mz && threadIdx.x < 4) {
			const int i = threadIdx.x < 2 ? (int)threadIdx.x : g.mx - 4 + (int)threadIdx.x, j = g.my;
			auto P = [&](int ii, int jj, int kk) -> double {
				const double v = psi_raw(src, g, ii, jj, kk);
				return INBC ? bc_psi(src, g, v, ii, jj, kk) : v;
			};
			const long p = kry_idx(g, i, j, k), sx = g.px, sz = g.ps;
			Sten s;
			s.pc = P(i, j, k); s.xm = P(i - 1, j, k); s.xp = P(i + 1, j, k);
			s.ym = P(i, j - 1, k); s.yp = P(i, j + 1, k); s.zm = P(i, j, k - 1); s.zp = P(i, j, k + 1);
			s.lam = a.lam[p];
			if (VAR) {
				s.dxx = a.Dxx[p]; s.dyy = a.Dyy[p]; s.dzz = a.Dzz[p];
				s.dxx_m = a.Dxx[p - 1]; s.dxx_p = a.Dxx[p + 1];
				s.dyy_m = a.Dyy[p - sx]; s.dyy_p = a.Dyy[p + sx];
				s.dzz_m = a.Dzz[p - sz]; s.dzz_p = a.Dzz[p + sz];
				if (OFFD) {
					s.dxy = a.Dxy[p];
					s.dxy_xm = a.Dxy[p - 1]; s.dxy_xp = a.Dxy[p + 1];
					s.dxy_ym = a.Dxy[p - sx]; s.dxy_yp = a.Dxy[p + sx];
					s.pmm = P(i - 1, j - 1, k); s.pmp = P(i - 1, j + 1, k);
					s.ppm = P(i + 1, j - 1, k); s.ppp = P(i + 1, j + 1, k);
				}
			}
			a.O[p] = apply_op<VAR, OFFD, EXACT>(s, a.op);
		}
		__threadfence_block();
		__syncthreads();
	}
	bc_xy_plane(a.O, g, k, (int)blockDim.x);
}

// Krylov::multiply_withMat on every point 0..m of the slab, neighbours read
// from memory (ghosts included).  A block owns 4 grid rows (one per wave) and
// walks them in 64-point strips; blockIdx.y counts row groups over (j,k).
template <bool VAR, bool OFFD, bool EXACT>
__global__ __launch_bounds__(NT) void g_spmv(KryGeo g, KryOp op, const double *psi, const double *lam,
                                             const double *Dxx, const double *Dyy, const double *Dzz,
                                             const double *Dxy, double *out)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const long rows = (long)(g.my + 1) * (g.mz + 1);
	const long sx = g.px, sz = g.ps;
	for (long row = (long)blockIdx.x * NW + w; row < rows; row += (long)gridDim.x * NW) {
		const int j = (int)(row % (g.my + 1)), k = (int)(row / (g.my + 1));
		const long p0 = kry_idx(g, 0, j, k);
		for (int i = lane; i <= g.mx; i += 64) {
			const long p = p0 + i;
			Sten s;
			s.pc = psi[p]; s.xm = psi[p - 1]; s.xp = psi[p + 1];
			s.ym = psi[p - sx]; s.yp = psi[p + sx]; s.zm = psi[p - sz]; s.zp = psi[p + sz];
			s.lam = lam[p];
			if (VAR) {
				s.dxx = Dxx[p]; s.dyy = Dyy[p]; s.dzz = Dzz[p];
				s.dxx_m = Dxx[p - 1]; s.dxx_p = Dxx[p + 1];
				s.dyy_m = Dyy[p - sx]; s.dyy_p = Dyy[p + sx];
				s.dzz_m = Dzz[p - sz]; s.dzz_p = Dzz[p + sz];
				if (OFFD) {
					s.dxy = Dxy[p];
					s.dxy_xm = Dxy[p - 1]; s.dxy_xp = Dxy[p + 1];
					s.dxy_ym = Dxy[p - sx]; s.dxy_yp = Dxy[p + sx];
					s.pmm = psi[p - 1 - sx]; s.pmp = psi[p - 1 + sx];
					s.ppm = psi[p + 1 - sx]; s.ppp = psi[p + 1 + sx];
				}
			}
			out[p] = apply_op<VAR, OFFD, EXACT>(s, op);
		}
	}
}

// Weighted dot product / squared norm over points 0..m (Krylov.C:161-244, 36-157):
// faces 1/2, edges 1/4, corners 1/8 (dot) or 1/4 (norm).  Slab interfaces are
// counted once, by the lower slab.  Wave w of block b takes grid rows
// 4b+w, 4b+w+4*gridDim.x, ...; a lane takes points lane, lane+64, ... of a row.
__global__ __launch_bounds__(NT) void g_wdot(KryGeo g, const double *x, const double *y, int norm_mode,
                                             KryRed red)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const long rows = (long)(g.my + 1) * (g.mz + 1);
	double acc[1] = {0.};
	for (long row = (long)blockIdx.x * NW + w; row < rows; row += (long)gridDim.x * NW) {
		const int j = (int)(row % (g.my + 1)), k = (int)(row / (g.my + 1));
		const bool by = (j == 0 && !g.nb[2]) || (j == g.my && !g.nb[3]);
		const bool bz = (k == 0 && !g.has_lo) || (k == g.mz && !g.has_hi);
		if ((k == 0 && g.has_lo) || (j == 0 && g.nb[2])) continue;   // counted by the lower neighbour
		const long p0 = kry_idx(g, 0, j, k);
		const int ilast = g.nb[1] ? g.mx - 1 : g.mx;                 // an x interface counts on its upper side
		for (int i = lane; i <= ilast; i += 64) {
			const bool bx = (i == 0 && !g.nb[0]) || (i == g.mx && !g.nb[1]);
			double wgt = (bx ? 0.5 : 1.) * (by ? 0.5 : 1.) * (bz ? 0.5 : 1.);
			if (norm_mode && bx && by && bz) wgt = 0.25;
			acc[0] += wgt * (x[p0 + i] * y[p0 + i]);
		}
	}
	reduce_and_finalize<1>(acc, red);
}

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Blocks per compute unit.  A compute unit that is left with ONE block of a fused kernel (4 waves) draws
// about three quarters of its share of the memory bandwidth; with two or more it draws all of it.  Grids
// are therefore sized in multiples of 2 x the number of compute units (choose_chunk_depth), which leaves
// no such tail for kernels that fit 2 or 4 blocks on a compute unit.  A kernel whose registers allow 3
// would run 1024 blocks as 3 + 1 per compute unit: it is held to 2 by asking for dynamic LDS it does not
// use (constant coefficients, 256^3: 0.191 -> 0.177 ms).  KRY_LDS_PAD=bytes overrides (tuning).
#define KRY_LDS_PER_CU (160 * 1024)
template <void (*KERN)(const KryFused)>
static int fused_pad()
{
	static int pad = -1;     // one per kernel instantiation; all GPUs of a process are alike
	if (pad < 0) {
		const char *e = getenv("KRY_LDS_PAD");
		int n = 0, p = 0;
		if (e) p = std::max(0, atoi(e));
		else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, KERN, NT, 0) == hipSuccess && n == 3) {
			hipFuncAttributes fa;
			if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(KERN)) == hipSuccess) {
				// two blocks fit, three do not: a little over a third of the LDS per block
				const int want = KRY_LDS_PER_CU / 3 + 2048;
				p = std::max(0, want - (int)fa.sharedSizeBytes);
			}
		}
		(void)hipGetLastError();
		pad = p;
	}
	return pad;
}

template <int KIND, int TY>
static hipError_t launch_fused_t(const KryFused &a, bool var, bool offd, bool exact, hipStream_t st)
{
	dim3 grid(a.nblk), block(NT);
#define KRY_LAUNCH(V, O, E) \
	hipLaunchKernelGGL((k_fused<KIND, TY, V, O, E, false>), grid, block, (fused_pad<k_fused<KIND, TY, V, O, E, false>>()), st, a)
	if (!var) KRY_LAUNCH(false, false, true);
	else if (!offd) { if (exact) KRY_LAUNCH(true, false, true); else KRY_LAUNCH(true, false, false); }
	else { if (exact) KRY_LAUNCH(true, true, true); else KRY_LAUNCH(true, true, false); }
#undef KRY_LAUNCH
	return hipGetLastError();
}

// general-boundary variants, 8-row tiles (constant coefficients: only where no face
// of the global domain needs the operator's input treated, i.e. all of them Dirichlet)
template <int KIND>
static hipError_t launch_fused_gen_t(const KryFused &a, bool var, bool offd, bool exact, bool inbc, hipStream_t st)
{
	dim3 grid(a.nblk), block(NT);
#define KRY_LAUNCH(V, O, E) \
	hipLaunchKernelGGL((k_fused<KIND, 8, V, O, E, true>), grid, block, (fused_pad<k_fused<KIND, 8, V, O, E, true>>()), st, a)
	if (!var && inbc)
		hipLaunchKernelGGL((k_fused<KIND, 8, false, false, true, true, true>), grid, block,
		                   (fused_pad<k_fused<KIND, 8, false, false, true, true, true>>()), st, a);
	else if (!var) KRY_LAUNCH(false, false, true);
	else if (!offd) { if (exact) KRY_LAUNCH(true, false, true); else KRY_LAUNCH(true, false, false); }
	else { if (exact) KRY_LAUNCH(true, true, true); else KRY_LAUNCH(true, true, false); }
#undef KRY_LAUNCH
	return hipGetLastError();
}

hipError_t kry_launch_psi_shell(int kind, const KryFused &a, hipStream_t st)
{
	const KryGeo &g = a.g;
	if (g.mx < 1 || g.my < 1 || g.mz < 1) return hipErrorInvalidValue;
	const long w = g.mx + 3, total = 4 * w * (g.my + 3) + (long)(g.mz - 1) * (4 * w + 4L * (g.my - 1));
	if (total > 0x7fffffffL) return hipErrorInvalidValue;
	const int is_a = (kind == KF_A0 || kind == KF_A1) ? 1 : 0, defer = (kind == KF_A0) ? 1 : 0;
	hipLaunchKernelGGL(g_psi_shell, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, is_a, defer);
	return hipGetLastError();
}

hipError_t kry_launch_fused_gen(int kind, const KryFused &a, bool var, bool offd, bool exact, bool inbc, hipStream_t st)
{
	switch (kind) {
	case KF_A0:  return launch_fused_gen_t<KF_A0>(a, var, offd, exact, inbc, st);
	case KF_A1:  return launch_fused_gen_t<KF_A1>(a, var, offd, exact, inbc, st);
	case KF_B0:  return launch_fused_gen_t<KF_B0>(a, var, offd, exact, inbc, st);
	case KF_B0L: return launch_fused_gen_t<KF_B0L>(a, var, offd, exact, inbc, st);
	case KF_B1L: return launch_fused_gen_t<KF_B1L>(a, var, offd, exact, inbc, st);
	}
	return hipErrorInvalidValue;
}

hipError_t kry_launch_surf(int kind, const KryFused &a, int nblk, hipStream_t st)
{
	dim3 grid(nblk), block(NT);
	switch (kind) {
	case KF_A0:  hipLaunchKernelGGL(k_surf<KF_A0>, grid, block, 0, st, a); break;
	case KF_A1:  hipLaunchKernelGGL(k_surf<KF_A1>, grid, block, 0, st, a); break;
	case KF_B0:  hipLaunchKernelGGL(k_surf<KF_B0>, grid, block, 0, st, a); break;
	case KF_B0L: hipLaunchKernelGGL(k_surf<KF_B0L>, grid, block, 0, st, a); break;
	case KF_B1L: hipLaunchKernelGGL(k_surf<KF_B1L>, grid, block, 0, st, a); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t kry_launch_fused(int kind, int ty, const KryFused &a, bool var, bool offd, bool exact, hipStream_t st)
{
#define KRY_KIND(K)                                                              \
	case K:                                                                      \
		return ty == 8 ? launch_fused_t<K, 8>(a, var, offd, exact, st)           \
		               : launch_fused_t<K, 16>(a, var, offd, exact, st);
	switch (kind) {
		KRY_KIND(KF_A0)
		KRY_KIND(KF_A1)
		KRY_KIND(KF_B0)
		KRY_KIND(KF_B0L)
		KRY_KIND(KF_B1L)
		KRY_KIND(KF_RES)
		KRY_KIND(KF_SPMV)
	}
#undef KRY_KIND
	return hipErrorInvalidValue;
}

// Blocks of the fused stencil kernels that one compute unit holds at a time, as they are launched
// (fused_pad): the smallest over the kernels of a pass; 0 when the runtime cannot tell.
int kry_fused_blocks_per_cu(int ty, bool var, bool offd, bool exact, bool gen)
{
	int least = 0;
	bool ok = true;
	auto take = [&](auto kern, int pad) {
		int n = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, NT, pad) != hipSuccess) { ok = false; return; }
		least = least ? std::min(least, n) : n;
	};
#define KRY_OCC(K, T, V, O, E, G) take(k_fused<K, T, V, O, E, G>, fused_pad<k_fused<K, T, V, O, E, G>>())
#define KRY_OCC_VOE(K, T, G)                                                                   \
	if (!var) KRY_OCC(K, T, false, false, true, G);                                            \
	else if (!offd) { if (exact) KRY_OCC(K, T, true, false, true, G); else KRY_OCC(K, T, true, false, false, G); } \
	else { if (exact) KRY_OCC(K, T, true, true, true, G); else KRY_OCC(K, T, true, true, false, G); }
#define KRY_OCC_KINDS(T, G) { KRY_OCC_VOE(KF_A0, T, G) KRY_OCC_VOE(KF_B0, T, G) }
	if (gen) KRY_OCC_KINDS(8, true)
	else if (ty == 8) KRY_OCC_KINDS(8, false)
	else KRY_OCC_KINDS(16, false)
#undef KRY_OCC_KINDS
#undef KRY_OCC_VOE
#undef KRY_OCC
	(void)hipGetLastError();
	return ok ? least : 0;
}

hipError_t kry_launch_mgs2(const KryStream &a, int nbx, int nby, hipStream_t st)
{
	if (a.gen) hipLaunchKernelGGL(k_mgs2<true>, dim3(nbx, nby), dim3(NT), 0, st, a);
	else hipLaunchKernelGGL(k_mgs2<false>, dim3(nbx, nby), dim3(NT), 0, st, a);
	return hipGetLastError();
}

hipError_t kry_launch_final(int L, const KryStream &a, int nbx, int nby, hipStream_t st)
{
	if (L == 1 && !a.gen) hipLaunchKernelGGL((k_final<1, false>), dim3(nbx, nby), dim3(NT), 0, st, a);
	else if (L == 2 && !a.gen) hipLaunchKernelGGL((k_final<2, false>), dim3(nbx, nby), dim3(NT), 0, st, a);
	else if (L == 1) hipLaunchKernelGGL((k_final<1, true>), dim3(nbx, nby), dim3(NT), 0, st, a);
	else if (L == 2) hipLaunchKernelGGL((k_final<2, true>), dim3(nbx, nby), dim3(NT), 0, st, a);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

hipError_t kry_launch_bicgn(const KryStream &a, const KryPointN &f, int nbx, int nby, hipStream_t st)
{
	hipLaunchKernelGGL(k_bicgn, dim3(nbx, nby), dim3(NT), 0, st, a, f);
	return hipGetLastError();
}

hipError_t kry_launch_mgsn(const KryStream &a, int jj, int iip, int last, int nbx, int nby, hipStream_t st)
{
	if (a.gen) hipLaunchKernelGGL(k_mgsn<true>, dim3(nbx, nby), dim3(NT), 0, st, a, jj, iip, last);
	else hipLaunchKernelGGL(k_mgsn<false>, dim3(nbx, nby), dim3(NT), 0, st, a, jj, iip, last);
	return hipGetLastError();
}

hipError_t kry_launch_finaln(const KryStream &a, const KryFinalN &f, int nbx, int nby, hipStream_t st)
{
	if (a.gen) hipLaunchKernelGGL(k_finaln<true>, dim3(nbx, nby), dim3(NT), 0, st, a, f);
	else hipLaunchKernelGGL(k_finaln<false>, dim3(nbx, nby), dim3(NT), 0, st, a, f);
	return hipGetLastError();
}

hipError_t kry_launch_scalar(KryState *S, int phase, hipStream_t st)
{
	hipLaunchKernelGGL(k_scalar, dim3(1), dim3(64), 0, st, S, phase);
	return hipGetLastError();
}

hipError_t kry_launch_sum_slabs(const double *gather, int nslabs, double *out, hipStream_t st)
{
	hipLaunchKernelGGL(k_sum_slabs, dim3(1), dim3(64), 0, st, gather, nslabs, out);
	return hipGetLastError();
}

hipError_t kry_launch_axpy(double *dst, const double *x, const double *y, double c, long n, hipStream_t st)
{
	const long n2 = n / 2;
	hipLaunchKernelGGL(g_axpy, dim3(min(cdiv(n2, NT), 8192)), dim3(NT), 0, st, dst, x, y, c, n2);
	return hipGetLastError();
}

hipError_t kry_launch_repack(double *dev, double *packed, int w, long rows, int px, bool to_device, hipStream_t st)
{
	hipLaunchKernelGGL(g_repack, dim3((unsigned)std::min<long>((rows + NW - 1) / NW, 32768)), dim3(NT), 0, st, dev, packed, w,
	                   rows, px, to_device ? 1 : 0);
	return hipGetLastError();
}

hipError_t kry_launch_halo_push(const double *send_lo, double *slot_lo, unsigned long long *flag_lo, const double *send_hi,
                                double *slot_hi, unsigned long long *flag_hi, size_t count, unsigned long long seq,
                                unsigned *ticket, const KryState *S, hipStream_t st)
{
	KryHaloSide lo = {send_lo, slot_lo, flag_lo}, hi = {send_hi, slot_hi, flag_hi};
	const int nb = (int)std::max<size_t>(1, std::min<size_t>((count / 2 + NT - 1) / NT, 256));
	hipLaunchKernelGGL(k_halo_push, dim3(nb), dim3(NT), 0, st, lo, hi, count, seq, ticket, S);
	return hipGetLastError();
}

hipError_t kry_launch_halo_pull(double *recv_lo, const double *slot_lo, unsigned long long *flag_lo, double *recv_hi,
                                const double *slot_hi, unsigned long long *flag_hi, size_t count, unsigned long long seq,
                                long long timeout, KryState *S, hipStream_t st)
{
	KryHaloSide lo = {slot_lo, recv_lo, flag_lo}, hi = {slot_hi, recv_hi, flag_hi};
	const int nb = (int)std::max<size_t>(1, std::min<size_t>((count / 2 + NT - 1) / NT, 128));
	hipLaunchKernelGGL(k_halo_pull, dim3(nb), dim3(NT), 0, st, lo, hi, count, seq, timeout, S);
	return hipGetLastError();
}

hipError_t kry_launch_link_probe(const KryRed &r, double v, hipStream_t st)
{
	hipLaunchKernelGGL(k_link_probe, dim3(1), dim3(NT), 0, st, r, v);
	return hipGetLastError();
}

hipError_t kry_launch_face(double *field, double *buf, const KryGeo &g, int dir, int idx, bool to_buf, hipStream_t st)
{
	const long n = (long)((dir == 0 ? g.my : g.mx) + 3) * (g.mz + 3);
	hipLaunchKernelGGL(g_face, dim3((unsigned)std::min<long>((n + NT - 1) / NT, 4096)), dim3(NT), 0, st, field, buf, g, dir, idx,
	                   to_buf ? 1 : 0);
	return hipGetLastError();
}

hipError_t kry_launch_sub(double *dst, const double *x, const double *y, long n, hipStream_t st)
{
	const long n2 = n / 2;
	hipLaunchKernelGGL(g_sub, dim3(min(cdiv(n2, NT), 8192)), dim3(NT), 0, st, dst, x, y, n2);
	return hipGetLastError();
}

hipError_t kry_launch_bcs(double *d, const KryGeo &g, hipStream_t st, int ndirs)
{
	// one sweep of the block over the rows, one over the columns where the plane allows it
	const int nt = std::min(1024, std::max(NT, ((std::max(g.mx + 1, g.my + 3) + 63) / 64) * 64));
	hipLaunchKernelGGL(g_bc_xy, dim3(g.mz + 3), dim3(nt), 0, st, d, g);
	if (ndirs > 2) hipLaunchKernelGGL(g_bc, dim3(cdiv(g.mx + 1, 64), g.my + 1), dim3(64), 0, st, d, g, 2);
	return hipGetLastError();
}

hipError_t kry_launch_bcs_rows(int kind, const KryFused &a, int row_hi, bool inbc, bool var, bool offd, bool exact,
                               hipStream_t st)
{
	const KryGeo &g = a.g;
	const int nt = std::min(1024, std::max(NT, ((std::max(g.mx + 1, g.my + 3) + 63) / 64) * 64));
	const int is_a = (kind == KF_A0 || kind == KF_A1) ? 1 : 0, defer = (kind == KF_A0) ? 1 : 0;
	// row my is left out by the tiles and an x face of the domain extrapolates
	const int op_row = (row_hi <= g.my && ((!g.nb[0] && g.bc[0] == 1) || (!g.nb[1] && g.bc[1] == 1))) ? 1 : 0;
	dim3 grid(g.mz + 3), block(nt);
#define KRY_ROWS(I, V, O, E) hipLaunchKernelGGL((g_bc_xy_rows<I, V, O, E>), grid, block, 0, st, a, row_hi, is_a, defer, op_row)
	if (!var && inbc) KRY_ROWS(true, false, false, true);
	else if (!var) KRY_ROWS(false, false, false, true);
	else if (!offd) { if (exact) KRY_ROWS(false, true, false, true); else KRY_ROWS(false, true, false, false); }
	else { if (exact) KRY_ROWS(false, true, true, true); else KRY_ROWS(false, true, true, false); }
#undef KRY_ROWS
	return hipGetLastError();
}

hipError_t kry_launch_gspmv(const KryGeo &g, const KryOp &op, bool var, bool offd, const double *psi,
                            const double *lam, const double *Dxx, const double *Dyy, const double *Dzz,
                            const double *Dxy, double *out, hipStream_t st)
{
	const long rows = (long)(g.my + 1) * (g.mz + 1);
	dim3 grid((unsigned)std::min<long>((rows + NW - 1) / NW, 16384)), block(NT);
	const bool ex = op.exact_rcp;
#define KRY_G(V, O, E) hipLaunchKernelGGL((g_spmv<V, O, E>), grid, block, 0, st, g, op, psi, lam, Dxx, Dyy, Dzz, Dxy, out)
	if (!var) KRY_G(false, false, true);
	else if (!offd) { if (ex) KRY_G(true, false, true); else KRY_G(true, false, false); }
	else { if (ex) KRY_G(true, true, true); else KRY_G(true, true, false); }
#undef KRY_G
	return hipGetLastError();
}

// Scratch of the allocation-free weighted dot: one partial sum per block of the largest grid
// g_wdot is ever launched with, then a 256-byte cell holding the ticket counter.  Every GPU
// holds its own copy of the array until the process ends (about 33 KB).  In host code the
// array's name is only a handle for hipGetSymbolAddress: kernels get the per-device address.
#define KRY_STATIC_PARTIALS KRY_WDOT_BLOCKS
static __device__ __align__(256) double kry_static_scratch[KRY_STATIC_PARTIALS + 32];
#define KRY_STATIC_MAX_DEVICES 64
static double *g_static_ptr[KRY_STATIC_MAX_DEVICES];   // address of the array on each device, 0 = not asked yet

// The address of the array on `device` (which the caller has made current), asked from the runtime once
// per device.  That first request may load the code object on the device -- it can allocate, and it is
// not a call to make while a stream capture is open -- so the library makes it where it initialises a
// device (kry_device_check, kry_create); the entry point itself only repeats it for a device neither has seen.
hipError_t kry_static_scratch_ready(int device, double **out)
{
	if (device < 0 || device >= KRY_STATIC_MAX_DEVICES) return hipErrorInvalidDevice;
	double *scratch = __atomic_load_n(&g_static_ptr[device], __ATOMIC_ACQUIRE);
	if (!scratch) {
		hipError_t e = hipGetSymbolAddress((void **)&scratch, HIP_SYMBOL(kry_static_scratch));
		if (e != hipSuccess) return e;
		if (!scratch) return hipErrorInvalidSymbol;
		__atomic_store_n(&g_static_ptr[device], scratch, __ATOMIC_RELEASE);
	}
	if (out) *out = scratch;
	return hipSuccess;
}

// Calls on one GPU share that GPU's array: two of them in flight at once race.
hipError_t kry_launch_wdot_static(int device, const KryGeo &g, const double *x, const double *y, int norm_mode,
                                  double *result, hipStream_t st)
{
	double *scratch = nullptr;
	{
		hipError_t e = kry_static_scratch_ready(device, &scratch);
		if (e != hipSuccess) return e;
	}
	const long rows = (long)(g.my + 1) * (g.mz + 1);
	dim3 grid((unsigned)std::min<long>((rows + NW - 1) / NW, KRY_WDOT_BLOCKS)), block(NT);
	if (grid.x < 1 || grid.x > KRY_STATIC_PARTIALS) return hipErrorInvalidValue;
	unsigned *counter = reinterpret_cast<unsigned *>(scratch + KRY_STATIC_PARTIALS);
	// the ticket must start at zero whatever an earlier (possibly aborted) call left behind
	hipError_t e = hipMemsetAsync(counter, 0, 256, st);
	if (e != hipSuccess) return e;
	KryRed red;
	red.partial = scratch;
	red.counter = counter;
	red.S = nullptr;
	red.nblk = (int)grid.x;
	red.phase = PH_NONE;
	red.out = result;
	red.out_sqrt = norm_mode ? 1 : 0;
	hipLaunchKernelGGL(g_wdot, grid, block, 0, st, g, x, y, norm_mode, red);
	return hipGetLastError();
}

hipError_t kry_launch_wdot(const KryGeo &g, const double *x, const double *y, int norm_mode, KryRed red,
                           hipStream_t st)
{
	const long rows = (long)(g.my + 1) * (g.mz + 1);
	dim3 grid((unsigned)std::min<long>((rows + NW - 1) / NW, KRY_WDOT_BLOCKS)), block(NT);
	red.nblk = (int)grid.x;
	hipLaunchKernelGGL(g_wdot, grid, block, 0, st, g, x, y, norm_mode, red);
	return hipGetLastError();
}
