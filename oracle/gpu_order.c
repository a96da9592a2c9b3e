/*
 * gpu_order.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Repeats the dot products and norms of the oracle's solve in the summation
 * order of the HIP kernels (krylov_amd/csrc/kry_kernels.hip).  Everything else
 * in the solve stays the line-faithful restatement of the reference; with this
 * hook installed the oracle and the device differ in nothing, so their residual
 * histories can be compared bit for bit.  Run without the hook, the same oracle
 * sums in the reference's order (Krylov.C:161-244); the two runs differ through
 * summation order only.
 *
 * Orders reproduced (256-thread blocks, 4 wavefronts of 64 lanes):
 *   per thread   products added in the order the kernel visits its points
 *   per wave     shuffle tree: v[i] += v[i+o], o = 32,16,8,4,2,1
 *   per block    wave sums added in wave order
 *   grid         thread t adds partials t, t+256, ...; then wave tree; then waves
 * Kernel shapes: the fused stencil kernels (tiles of 128 x ty points marched over
 * zc planes, XCD-swizzled block order; the general-boundary variant shifts the
 * tiles onto the ghost-inclusive box and is followed by a boundary-plane kernel),
 * the streaming kernels (2048-double plane segments x plane groups) and the
 * generic weighted dot (waves walking grid rows).
 */
#include "krylov_oracle.h"

#include <math.h>
#include <stdlib.h>

#define NT 256
#define NW 4

static int g_mode = 0;          /* 0 off, 1 fused Dirichlet path, 2 generic path, 3 fused general-boundary path */
static int g_ty = 8, g_zc = 16, g_sb = 2048;
static int g_nslabs = 1;        /* z slabs of the device run (sums are formed per slab, then added in slab order) */
static int g_split = 0;         /* 1: the stencil kernels of a multi-slab run are launched in two parts (edge chunks first) */
static int g_gen_y0 = 0;        /* 1: the general-boundary tiles start at grid row 0 (rows 0 .. my-1 tiled) instead of -1 */

/* one z slab of the global box: local plane k is global plane z0 + k */
typedef struct { int z0, mz, has_lo, has_hi; } zslab;
static zslab slab_of(const orc_cfg *c, int s)
{
	zslab q;
	q.mz = c->mx[2] / g_nslabs;
	q.z0 = s * q.mz;
	q.has_lo = s > 0;
	q.has_hi = s < g_nslabs - 1;
	return q;
}


typedef struct { int mx, my, mz; long sx, sy; } hbox;
static hbox hb(const orc_cfg *c)
{
	hbox b;
	b.mx = c->mx[0]; b.my = c->mx[1]; b.mz = c->mx[2];
	b.sx = b.mx + 3;
	b.sy = b.sx * (long)(b.my + 3);
	return b;
}
#define HAT(b, a, i, j, k) ((a)[((long)(k) + 1) * (b).sy + ((long)(j) + 1) * (b).sx + ((i) + 1)])

static double wave_tree(double *v)
{
	for (int o = 32; o > 0; o >>= 1)
		for (int i = 0; i < o; ++i) v[i] = v[i] + v[i + o];
	return v[0];
}

static double block_total(double *thread_acc /* NT */)
{
	double w[NW];
	for (int k = 0; k < NW; ++k) w[k] = wave_tree(thread_acc + 64 * k);
	double v = w[0];
	for (int k = 1; k < NW; ++k) v += w[k];
	return v;
}

static double grid_total(const double *partial, long nblk)
{
	double tot[NT];
	for (int t = 0; t < NT; ++t) {
		double s = 0.;
		for (long b = t; b < nblk; b += NT) s += partial[b];
		tot[t] = s;
	}
	return block_total(tot);
}

/* A block's partial sums are stored under the index of the TILE it worked on, not under its place in
 * the launch: the order in which the blocks take the tiles (XCD-aware, ascending or descending) does not
 * enter the sums.  Position in the partial array -> tile index: the identity. */
static long swizzle(long bp, long nb)
{
	(void)nb;
	return bp;
}

/* fused stencil kernels: k_fused, on one slab */
static int fused_geometry(const orc_cfg *c, zslab sl, int *nchunk, int *n_edge)
{
	const int zlo = sl.has_lo ? 0 : 1, zhi = sl.has_hi ? sl.mz : sl.mz - 1;
	*nchunk = (zhi - zlo + 1 + g_zc - 1) / g_zc;
	const int c_hi = (sl.mz - 1 - zlo) / g_zc;
	*n_edge = (c_hi == *nchunk - 1) ? 2 : 3;
	(void)c;
	return *nchunk > *n_edge;
}


static double fused_sum_slab(const orc_cfg *c, const double *A, const double *B, zslab sl, int split)
{
	const hbox b = hb(c);
	const int ty = g_ty, zc = g_zc, VD = ty / NW;
	const int ntx = (b.mx + 127) / 128, nty = (b.my + ty - 1) / ty;
	const int zlo = sl.has_lo ? 0 : 1, zhi = sl.has_hi ? sl.mz : sl.mz - 1;
	/* split launch: chunk 0 and the last n_edge-1 chunks first, the rest second (kry_solver.cpp, stencil_step) */
	int nchunk, n_edge;
	fused_geometry(c, sl, &nchunk, &n_edge);
	const long tiles = (long)ntx * nty, nblk = tiles * nchunk;
	double *partial = (double *)malloc((size_t)nblk * sizeof(double));
	for (long pi = 0; pi < nblk; ++pi) {
		long bid;
		int ch;
		if (!split) {
			bid = swizzle(pi, nblk);
			ch = (int)(bid / tiles);
		} else if (pi < n_edge * tiles) {
			bid = swizzle(pi, n_edge * tiles);
			ch = (int)(bid / tiles);
			ch = ch ? nchunk - n_edge + ch : 0;
		} else {
			bid = swizzle(pi - n_edge * tiles, (nchunk - n_edge) * tiles);
			ch = (int)(bid / tiles) + 1;
		}
		const int tx = (int)(bid % ntx), tyi = (int)((bid / ntx) % nty);
		const int z0 = zlo + ch * zc;
		const int z1 = (z0 + zc < zhi + 1) ? z0 + zc : zhi + 1;
		const int y0 = tyi * ty;
		double acc[NT];
		for (int t = 0; t < NT; ++t) {
			const int w = t >> 6, lane = t & 63;
			const int i0 = 128 * tx + 2 * lane;
			double s = 0.;
			for (int z = z0; z < z1; ++z)
				for (int e = 0; e < VD; ++e) {
					const int j = y0 + w * VD + e;
					if (j < 1 || j > b.my - 1 || z < 1) continue;
					const int zg = sl.z0 + z;
					if (i0 >= 1 && i0 <= b.mx - 1) s += HAT(b, A, i0, j, zg) * HAT(b, B, i0, j, zg);
					if (i0 + 1 >= 1 && i0 + 1 <= b.mx - 1) s += HAT(b, A, i0 + 1, j, zg) * HAT(b, B, i0 + 1, j, zg);
				}
			acc[t] = s;
		}
		partial[pi] = block_total(acc);
	}
	double v = grid_total(partial, nblk);
	free(partial);
	return v;
}

static double fused_sum_opt(const orc_cfg *c, const double *A, const double *B, int may_split)
{
	/* the launches are split only when every slab has chunks between its edge chunks */
	int split = may_split && g_split && g_nslabs > 1;
	for (int s = 0; s < g_nslabs && split; ++s) {
		int nc, ne;
		split = fused_geometry(c, slab_of(c, s), &nc, &ne);
	}
	double v = fused_sum_slab(c, A, B, slab_of(c, 0), split);
	for (int s = 1; s < g_nslabs; ++s) v += fused_sum_slab(c, A, B, slab_of(c, s), split);
	return v;
}

static double fused_sum(const orc_cfg *c, const double *A, const double *B) { return fused_sum_opt(c, A, B, 1); }
/* the residual kernel of the prologue is launched in one part */
static double fused_sum_whole(const orc_cfg *c, const double *A, const double *B) { return fused_sum_opt(c, A, B, 0); }

/* streaming kernels: k_final, k_mgs2, on one slab */
static double stream_sum_slab(const orc_cfg *c, const double *A, const double *B, int min_planes, zslab sl)
{
	const hbox b = hb(c);
	const long px = ((b.mx + 4 + 15) / 16) * 16, ps = px * (b.my + 3);
	const int nbx = (int)((ps + 2047) / 2048);
	const int zfirst = sl.has_lo ? -1 : 1, zlast = sl.has_hi ? sl.mz + 1 : sl.mz - 1;
	const int nplanes = zlast - zfirst + 1;
	const int cnt_lo = 1, cnt_hi = sl.has_hi ? sl.mz : sl.mz - 1;
	int nby = (g_sb + nbx - 1) / nbx;
	const int cap = nplanes / min_planes > 1 ? nplanes / min_planes : 1;   /* planes per block: 4 in k_mgs2, 1 in k_final */
	if (nby > cap) nby = cap;
	if (nby < 1) nby = 1;
	const long nblk = (long)nbx * nby;
	double *partial = (double *)malloc((size_t)nblk * sizeof(double));
	for (int by = 0; by < nby; ++by)
		for (int bx = 0; bx < nbx; ++bx) {
			double acc[NT];
			for (int t = 0; t < NT; ++t) {
				double s = 0.;
				for (int pl = by; pl < nplanes; pl += nby) {
					const int z = zfirst + pl;
					if (z < cnt_lo || z > cnt_hi) continue;
					for (int i = 0; i < 4; ++i) {
						const long q = 2 * ((long)bx * 1024 + i * NT + t);
						if (q >= ps) break;
						for (int d = 0; d < 2; ++d) {
							const long qq = q + d + 2;   /* the kernels start a plane at column 2 */
							const int jj = (int)(qq / px) - 1, ii = (int)(qq % px) - 2;
							if (ii >= 1 && ii <= b.mx - 1 && jj >= 1 && jj <= b.my - 1 && qq < ps)
								s += HAT(b, A, ii, jj, sl.z0 + z) * HAT(b, B, ii, jj, sl.z0 + z);
						}
					}
				}
				acc[t] = s;
			}
			partial[bx + (long)nbx * by] = block_total(acc);
		}
	double v = grid_total(partial, nblk);
	free(partial);
	return v;
}

static double stream_sum(const orc_cfg *c, const double *A, const double *B, int min_planes)
{
	double v = stream_sum_slab(c, A, B, min_planes, slab_of(c, 0));
	for (int s = 1; s < g_nslabs; ++s) v += stream_sum_slab(c, A, B, min_planes, slab_of(c, s));
	return v;
}

/* ---- general-boundary fused path ---- */

/* weight of local point (i,j,k) of a slab: 1/2 per face of the GLOBAL domain it lies on */
static double face_weight(const hbox *b, zslab sl, int i, int j, int k)
{
	const int fx = (i == 0 || i == b->mx), fy = (j == 0 || j == b->my);
	const int fz = (k == 0 && !sl.has_lo) || (k == sl.mz && !sl.has_hi);
	return (fx ? 0.5 : 1.) * (fy ? 0.5 : 1.) * (fz ? 0.5 : 1.);
}

/* k_fused<GEN> (tiles start at i = 0 and row j = -1, planes 0..mz, strict interior
 * counted) followed by k_surf (boundary planes, weighted), one reduction over both; on one slab */
static double fused_gen_sum_slab(const orc_cfg *c, const double *A, const double *B, zslab sl)
{
	const hbox b = hb(c);
	const int ty = 8, zc = g_zc, VD = ty / NW;
	const int ntx = (b.mx + 127) / 128, nty = g_gen_y0 ? (b.my + ty - 1) / ty : (b.my + 3 + ty - 1) / ty;
	const int zlo = 0, zhi = sl.mz;
	const int nchunk = (zhi - zlo + 1 + zc - 1) / zc;
	const long nf = (long)ntx * nty * nchunk;
	const long nxy = (long)(b.mx + 1) * (b.my + 1);
	const int nzf = (sl.has_lo ? 0 : 1) + (sl.has_hi ? 0 : 1);
	const int k1 = 1, k2 = sl.has_hi ? sl.mz : sl.mz - 1, nk = k2 - k1 + 1;
	const long tz = nzf * nxy, tyy = 2L * nk * (b.mx + 1), txx = 2L * nk * (b.my - 1), total = tz + tyy + txx;
	long ns = (total + 255) / 256;
	if (ns > 512) ns = 512;
	if (ns < 1) ns = 1;
	double *partial = (double *)malloc((size_t)(nf + ns) * sizeof(double));
	for (long bp = 0; bp < nf; ++bp) {
		const long bid = swizzle(bp, nf);
		const int tx = (int)(bid % ntx), tyi = (int)((bid / ntx) % nty), ch = (int)(bid / ((long)ntx * nty));
		const int z0 = zlo + ch * zc;
		const int z1 = (z0 + zc < zhi + 1) ? z0 + zc : zhi + 1;
		const int y0 = tyi * ty - (g_gen_y0 ? 0 : 1);
		double acc[NT];
		for (int t = 0; t < NT; ++t) {
			const int w = t >> 6, lane = t & 63;
			const int i0 = 128 * tx + 2 * lane;
			double s = 0.;
			for (int z = z0; z < z1; ++z)
				for (int e = 0; e < VD; ++e) {
					const int j = y0 + w * VD + e;
					if (j < 1 || j > b.my - 1 || z < 1 || z > k2) continue;
					const int zg = sl.z0 + z;
					if (i0 >= 1 && i0 <= b.mx - 1) s += HAT(b, A, i0, j, zg) * HAT(b, B, i0, j, zg);
					if (i0 + 1 >= 1 && i0 + 1 <= b.mx - 1) s += HAT(b, A, i0 + 1, j, zg) * HAT(b, B, i0 + 1, j, zg);
				}
			acc[t] = s;
		}
		partial[bp] = block_total(acc);
	}
	for (long blk = 0; blk < ns; ++blk) {
		double acc[NT];
		for (int t0 = 0; t0 < NT; ++t0) {
			double s = 0.;
			for (long t = blk * NT + t0; t < total; t += ns * NT) {
				int i, j, k;
				if (t < tz) {
					const int f = (int)(t / nxy);
					const long rem = t - f * nxy;
					j = (int)(rem / (b.mx + 1));
					i = (int)(rem - (long)j * (b.mx + 1));
					k = (f == 0 && !sl.has_lo) ? 0 : sl.mz;
				} else if (t < tz + tyy) {
					const long rem = t - tz, r2 = rem / (b.mx + 1);
					i = (int)(rem - r2 * (b.mx + 1));
					j = (r2 & 1) ? b.my : 0;
					k = k1 + (int)(r2 >> 1);
				} else {
					const long rem = t - tz - tyy, r2 = rem / (b.my - 1);
					j = 1 + (int)(rem - r2 * (b.my - 1));
					i = (r2 & 1) ? b.mx : 0;
					k = k1 + (int)(r2 >> 1);
				}
				s += face_weight(&b, sl, i, j, k) * (HAT(b, A, i, j, sl.z0 + k) * HAT(b, B, i, j, sl.z0 + k));
			}
			acc[t0] = s;
		}
		partial[nf + blk] = block_total(acc);
	}
	double v = grid_total(partial, nf + ns);
	free(partial);
	return v;
}

static double fused_gen_sum(const orc_cfg *c, const double *A, const double *B)
{
	double v = fused_gen_sum_slab(c, A, B, slab_of(c, 0));
	for (int s = 1; s < g_nslabs; ++s) v += fused_gen_sum_slab(c, A, B, slab_of(c, s));
	return v;
}

/* k_final<GEN>, k_mgs2<GEN>: planes -1..mz+1, weighted points; on one slab */
static double stream_gen_sum_slab(const orc_cfg *c, const double *A, const double *B, int min_planes, int norm_mode,
                                  zslab sl)
{
	const hbox b = hb(c);
	const long px = ((b.mx + 4 + 15) / 16) * 16, ps = px * (b.my + 3);
	const int nbx = (int)((ps + 2047) / 2048);
	const int nplanes = sl.mz + 3;
	const int cnt_lo = sl.has_lo ? 1 : 0, cnt_hi = sl.mz;
	int nby = (g_sb + nbx - 1) / nbx;
	const int cap = nplanes / min_planes > 1 ? nplanes / min_planes : 1;
	if (nby > cap) nby = cap;
	if (nby < 1) nby = 1;
	const long nblk = (long)nbx * nby;
	double *partial = (double *)malloc((size_t)nblk * sizeof(double));
	for (int by = 0; by < nby; ++by)
		for (int bx = 0; bx < nbx; ++bx) {
			double acc[NT];
			for (int t = 0; t < NT; ++t) {
				double s = 0.;
				for (int pl = by; pl < nplanes; pl += nby) {
					const int z = -1 + pl;
					if (z < cnt_lo || z > cnt_hi) continue;
					for (int i = 0; i < 4; ++i) {
						const long q = 2 * ((long)bx * 1024 + i * NT + t);
						if (q >= ps) break;
						for (int d = 0; d < 2; ++d) {
							const long qq = q + d + 2;
							const int jj = (int)(qq / px) - 1, ii = (int)(qq % px) - 2;
							if (ii < 0 || ii > b.mx || jj < 0 || jj > b.my) continue;
							double wgt = face_weight(&b, sl, ii, jj, z);
							if (norm_mode && wgt == 0.125) wgt = 0.25;
							s += wgt * (HAT(b, A, ii, jj, sl.z0 + z) * HAT(b, B, ii, jj, sl.z0 + z));
						}
					}
				}
				acc[t] = s;
			}
			partial[bx + (long)nbx * by] = block_total(acc);
		}
	double v = grid_total(partial, nblk);
	free(partial);
	return v;
}

static double stream_gen_sum(const orc_cfg *c, const double *A, const double *B, int min_planes, int norm_mode)
{
	double v = stream_gen_sum_slab(c, A, B, min_planes, norm_mode, slab_of(c, 0));
	for (int s = 1; s < g_nslabs; ++s) v += stream_gen_sum_slab(c, A, B, min_planes, norm_mode, slab_of(c, s));
	return v;
}

/* generic weighted dot: g_wdot (wave w of block b walks grid rows 4b+w, +4*nblk, ...;
 * a lane takes points lane, lane+64, ... of each row), on one slab */
static double wdot_sum_slab(const orc_cfg *c, const double *A, const double *B, int norm_mode, zslab sl)
{
	const hbox b = hb(c);
	const long rows = (long)(b.my + 1) * (sl.mz + 1);
	long nblk = (rows + NW - 1) / NW;
	if (nblk > 4096) nblk = 4096;
	double *partial = (double *)malloc((size_t)nblk * sizeof(double));
	for (long blk = 0; blk < nblk; ++blk) {
		double acc[NT];
		for (int t = 0; t < NT; ++t) {
			const int lane = t & 63, w = t >> 6;
			double s = 0.;
			for (long row = blk * NW + w; row < rows; row += nblk * NW) {
				const int j = (int)(row % (b.my + 1)), k = (int)(row / (b.my + 1));
				if (k == 0 && sl.has_lo) continue;      /* counted by the lower slab */
				const int fy = (j == 0 || j == b.my);
				const int fz = (k == 0 && !sl.has_lo) || (k == sl.mz && !sl.has_hi);
				for (int i = lane; i <= b.mx; i += 64) {
					const int fx = (i == 0 || i == b.mx);
					double wgt = (fx ? 0.5 : 1.) * (fy ? 0.5 : 1.) * (fz ? 0.5 : 1.);
					if (norm_mode && fx && fy && fz) wgt = 0.25;
					s += wgt * (HAT(b, A, i, j, sl.z0 + k) * HAT(b, B, i, j, sl.z0 + k));
				}
			}
			acc[t] = s;
		}
		partial[blk] = block_total(acc);
	}
	double v = grid_total(partial, nblk);
	free(partial);
	return v;
}

static double wdot_sum(const orc_cfg *c, const double *A, const double *B, int norm_mode)
{
	double v = wdot_sum_slab(c, A, B, norm_mode, slab_of(c, 0));
	for (int s = 1; s < g_nslabs; ++s) v += wdot_sum_slab(c, A, B, norm_mode, slab_of(c, s));
	return v;
}

static double hook(void *ctx, int site, int iter, int jj, const orc_cfg *c, const double *a, const double *b)
{
	(void)ctx;
	/* l > 2: every dot of the Gram-Schmidt part has a streaming kernel of its own (k_mgsn) */
	const int big = c->L > 2;
	if (g_mode == 2) {   /* generic path: every reduction is a g_wdot launch */
		if (site == ORC_SITE_NORM_RHS || site == ORC_SITE_NORM_INIT || site == ORC_SITE_NORM)
			return sqrt(wdot_sum(c, a, b, 1));
		return wdot_sum(c, a, b, 0);
	}
	if (g_mode == 3) {   /* general-boundary fused path */
		switch (site) {
		case ORC_SITE_NORM_RHS:
		case ORC_SITE_NORM_INIT: return sqrt(wdot_sum(c, a, b, 1));
		case ORC_SITE_RHO:
			if (jj == 0) return iter == 1 ? wdot_sum(c, a, b, 0) : stream_gen_sum(c, a, b, 1, 0);
			return fused_gen_sum(c, a, b);
		case ORC_SITE_ALPHA:     return fused_gen_sum(c, a, b);
		case ORC_SITE_TAU:       return big ? stream_gen_sum(c, a, b, 4, 0) : fused_gen_sum(c, a, b);
		case ORC_SITE_SIGMA:     return (jj == 1 && !big) ? fused_gen_sum(c, a, b) : stream_gen_sum(c, a, b, 4, 0);
		case ORC_SITE_GAMMAP:    /* l = 2: the dot (r0, r1) rides in k_mgs2 */
			return (jj == 1 && c->L == 1) ? fused_gen_sum(c, a, b) : stream_gen_sum(c, a, b, 4, 0);
		case ORC_SITE_NORM:      return sqrt(stream_gen_sum(c, a, b, 1, 1));
		}
		return orc_dot(c, a, b);
	}
	switch (site) {
	case ORC_SITE_NORM_RHS:  return sqrt(wdot_sum(c, a, b, 1));
	case ORC_SITE_NORM_INIT: return sqrt(fused_sum_whole(c, a, b));    /* residual kernel */
	case ORC_SITE_RHO:
		if (jj == 0) return iter == 1 ? fused_sum_whole(c, a, b)          /* |r0|^2 of the residual kernel */
		                              : stream_sum(c, a, b, 1);           /* end-of-pass kernel */
		return fused_sum(c, a, b);                                        /* B kernel of jj-1 */
	case ORC_SITE_ALPHA:     return fused_sum(c, a, b);                   /* A kernels */
	case ORC_SITE_TAU:       return big ? stream_sum(c, a, b, 4) : fused_sum(c, a, b);   /* k_mgsn : last B kernel */
	case ORC_SITE_SIGMA:     return (jj == 1 && !big) ? fused_sum(c, a, b) : stream_sum(c, a, b, 4);   /* k_mgs2, k_mgsn */
	case ORC_SITE_GAMMAP:    /* l = 2: the dot (r0, r1) rides in k_mgs2 */
		return (jj == 1 && c->L == 1) ? fused_sum(c, a, b) : stream_sum(c, a, b, 4);
	case ORC_SITE_NORM:      return sqrt(stream_sum(c, a, b, 1));
	}
	return orc_dot(c, a, b);
}

void orc_gpu_order_slabs(int nslabs, int split_launch)
{
	g_nslabs = nslabs > 0 ? nslabs : 1;
	g_split = split_launch != 0;
}

void orc_use_gpu_order(int enable, int ty, int zc, int stream_blocks)
{
	g_mode = enable;
	if (ty > 0) g_ty = ty;
	g_gen_y0 = ty < 0;              /* kry_stats.tile_rows < 0 (general-boundary path: always 8 rows) */
	if (zc > 0) g_zc = zc;
	if (stream_blocks > 0) g_sb = stream_blocks;
	orc_set_reduce_hook(enable ? hook : NULL, NULL);
}
