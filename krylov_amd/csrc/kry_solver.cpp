// Host side of the MI355X BiCGStab(l) stencil solver and its C ABI.
//
// One kry_solver owns `slab_count` z slabs (one HIP device and stream each).
// Slabs of other processes are reached through RCCL (kry_comm.cpp).  The outer
// iteration is a fixed sequence of kernel launches; all recurrences' scalars
// live on the device (KryState), so a pass needs no host round trip, and a
// device-side `done` flag turns launches queued past convergence into no-ops.
#include "../../include/krylov_amd.h"
#include "kry_comm.h"
#include "kry_kernels.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;

static void set_err(const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
}

#define HIPCHK(expr)                                                                      \
	do {                                                                                  \
		hipError_t e_ = (expr);                                                           \
		if (e_ != hipSuccess) {                                                           \
			set_err("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
			return KRY_ERR_HIP;                                                           \
		}                                                                                 \
	} while (0)

#define CHK(expr)                \
	do {                         \
		int s_ = (expr);         \
		if (s_ < 0) return s_;   \
	} while (0)

namespace {

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct Field {
	double *alloc = nullptr;
	double *p = nullptr;       // (c=0, j=-1, k=-1)
};

enum { NPINGPONG = 2 };

struct Slab {
	int index = 0;             // global slab index
	int device = 0;
	hipStream_t st = nullptr;
	hipStream_t st_comm = nullptr;   // halo traffic, overlapped with the interior chunks
	hipEvent_t ev = nullptr;   // "this slab's last kernel of the phase is queued"
	hipEvent_t ev_in = nullptr;
	hipEvent_t ev_edge = nullptr;    // the outermost chunks of a stencil kernel are done
	hipEvent_t ev_halo = nullptr;    // this slab's ghost planes are filled
	hipEvent_t ev_read = nullptr;    // this slab's copies out of its neighbours' planes are queued
	KryGeo g;
	long guard = 0;
	Field phi, rhs, lam, Dxx, Dyy, Dzz, Dxy, rt, tmp;
	Field rb[KRY_LMAX + 1][NPINGPONG], ub[KRY_LMAX + 1][NPINGPONG];
	KryState *S = nullptr;
	KryState *Sh = nullptr;    // pinned host mirror (head of the struct only)
	double *partial = nullptr;
	long partial_cap = 0;
	unsigned *counter = nullptr;
	double *gather = nullptr;  // slab_count * KRY_NSUMS doubles (used on the first slab)
	double *stage = nullptr;   // device staging buffer of the host<->device copies (freed after use)
	size_t stage_cap = 0;
	double *xy_buf = nullptr;  // packed x / y faces of the Cartesian split: 3 planes out, 3 planes in
	size_t xy_plane = 0;       // doubles per packed plane slot
	// mailbox all-reduce between processes (kry_mail_export / kry_mail_connect)
	double *mail_box = nullptr;        // own mailbox, uncached device memory, exported over HIP IPC
	KryMail *mail_dev = nullptr;       // its description for the kernels
	std::vector<void *> mail_mapped;   // peers' mailboxes as opened here
	bool mail_on = false;
	size_t link_cap = 0;               // doubles per landing slot of the ghost-plane exchange
	KryComm link;                      // link to the other slabs of THIS process (kry_solver::slab_link)
	std::vector<Field *> all;
};

struct Prof {
	std::vector<hipEvent_t> ev;
	std::vector<int> kind;
	size_t used = 0;
};

} // namespace

struct kry_solver {
	kry_desc d;
	int L = 1;
	int mzl = 0;               // slab-local mz
	bool fused = false;        // fused path in use (device-resident recurrences)
	bool gen = false;          // its general-boundary variant (any face types)
	bool inbc = false;         // ... with the operator's input passed through do_BCs (constant coefficients, Krylov.C:516)
	bool gen_y0 = false;       // ... with tile rows 0 .. instead of -1 .. (the rows -1, my, my+1 are left to the face kernel:
	                           // 64 instead of 65 tile rows at my = 512, which fills the block slots exactly); not for a
	                           // Cartesian split (a y neighbour above needs the operator on the whole row my)
	// Cartesian split of the global domain over processes (mpi_manager.C:56-305): np
	// sub-domains per direction, this process at co.  {1,1,nslabs} for plain z slabs.
	int np[3] = {1, 1, 1}, co[3] = {0, 0, 0};
	bool cart = false;         // split in x or y as well (one sub-domain per process)
	int rank_of(int cx, int cy, int cz) const { return (cx * np[1] + cy) * np[2] + cz; }   // reorder = 0
	int my_rank() const { return rank_of(co[0], co[1], co[2]); }
	int n_ranks() const { return np[0] * np[1] * np[2]; }
	bool var = false, offd = false;
	bool loaded = false;
	bool reset_needed = false;  // a field was replaced (kry_update_*) and the prologue has not run since
	KryOp op;
	int ty = 8, zc = 16;
	bool zc_fixed = false;     // KRY_ZC given: no automatic choice
	int stream_blocks = 65536; // target grid size of the streaming kernels
	std::vector<Slab> slabs;
	KryComm comm;
	// run state
	int it_base = 0;           // outer iterations completed (mirrors S->iter)
	int status = 0;
	bool done = false;
	// Several slabs in this process and no other process: the slabs exchange like the processes of a
	// split do through mapped memory -- every slab owns a mailbox and landing slots in uncached memory
	// of its device, which the other slabs' kernels address directly (peer access), the kernel that
	// finishes a reduction adds the slabs' sums itself, and the ghost planes are pushed by a kernel of
	// the sender.  No copies, cross-slab event waits or extra kernels are queued per dot product.
	bool slab_link = false;
	std::string slab_link_note = "one slab";   // why the link is on or off
	bool halo_pending = false;  // an exchange of an operator output is running on the exchange stream (Cartesian split)
	bool alt = true;            // every kernel of a pass takes its work in the opposite order of the one before it
	bool tail_bc_done = false;  // the post-loop do_BCs(phi) of a constant-coefficient solve has been applied
	double norm_rhs = 0, norm_init = 0, norm_final = 0;
	double solve_ms = 0, stage_ms = 0;
	double pending_stage_ms = 0;   // staging done by kry_update_* since the last solve
	int staged_fields = 0;         // field copies into the library for the current / last solve
	int overlapped_exchanges = 0;  // stencil launches split around a pending exchange since the last load
	double enqueue_ms = 0;     // host time spent queueing the passes of the last run
	std::vector<double> hist;  // generic path history (fused path keeps it on the device)
	// two consecutive passes (even, odd: the ping-pong buffers swap roles) captured as a
	// HIP graph: one launch call per pair instead of 6-20 kernel launches.  Off unless
	// KRY_GRAPH=1: the passes are chains of dependent, GPU-bound kernels queued far
	// ahead of the device, and replaying them as a graph measured 0 to -2 % from 32^3
	// to 512^3.
	hipGraphExec_t pass_pair = nullptr;
	bool use_graph = false;
	std::vector<double> step_ms;   // wall time per pass (a batch's time spread evenly over its passes)
	bool profile = false;
	bool overlap = true;       // halo exchange concurrent with the interior chunks
	Prof prof;
	double kernel_ms[10] = {0};   // 0..7: kernel kinds; 8: all-reduces of the transport, 9: waits for ghost planes
	int kernel_calls[10] = {0};
	double tick_ms = 1e-5;        // milliseconds per wall_clock64 tick of the first slab's device
	int debug = 0;
	// generic path scalars
	double g_rho0 = 1, g_alpha = 0, g_omega = 1;
};

namespace {

// ------------------------------------------------------------------ memory

int alloc_field(Slab &s, Field &f)
{
	HIPCHK(hipSetDevice(s.device));
	const size_t total = (size_t)(s.g.n + 2 * s.guard + 32);
	HIPCHK(hipMalloc((void **)&f.alloc, total * sizeof(double)));
	HIPCHK(hipMemsetAsync(f.alloc, 0, total * sizeof(double), s.st));
	// shift by 14 doubles: column 2 (grid index 0) then sits on a 128-byte boundary
	f.p = f.alloc + s.guard + 14;
	s.all.push_back(&f);
	return 0;
}

void free_fields(Slab &s)
{
	hipSetDevice(s.device);
	for (Field *f : s.all) {
		if (f->alloc) hipFree(f->alloc);
		f->alloc = f->p = nullptr;
	}
	s.all.clear();
}

// host box <-> device box.  Host planes of the slab start at k_host = index_local*mzl.
// The host block of a slab is contiguous; it crosses PCIe as one linear copy into /
// out of a device staging buffer and is re-pitched by a kernel (a pitched 2D copy
// from pageable memory moves row by row and is an order of magnitude slower).
int ensure_stage(Slab &s, size_t doubles)
{
	if (s.stage_cap >= doubles) return 0;
	HIPCHK(hipSetDevice(s.device));
	if (s.stage) HIPCHK(hipFree(s.stage));
	s.stage = nullptr;
	s.stage_cap = 0;
	HIPCHK(hipMalloc((void **)&s.stage, doubles * sizeof(double)));
	s.stage_cap = doubles;
	return 0;
}

void drop_stage(Slab &s)
{
	if (s.stage) { hipSetDevice(s.device); hipFree(s.stage); }
	s.stage = nullptr;
	s.stage_cap = 0;
}

int upload(kry_solver *h, Slab &s, int local_idx, Field &f, const double *host)
{
	HIPCHK(hipSetDevice(s.device));
	if (local_idx == 0) h->staged_fields += 1;
	const int mxh = h->d.mx[0], myh = h->d.mx[1];
	const size_t hrow = (size_t)(mxh + 3);
	const double *src = host + (size_t)local_idx * h->mzl * hrow * (myh + 3);
	const size_t rows = (size_t)(myh + 3) * (s.g.mz + 3);
	CHK(ensure_stage(s, rows * hrow));
	HIPCHK(hipMemcpyAsync(s.stage, src, rows * hrow * sizeof(double), hipMemcpyDefault, s.st));
	HIPCHK(kry_launch_repack(f.p, s.stage, (int)hrow, (long)rows, s.g.px, true, s.st));
	return 0;
}

int download(kry_solver *h, Slab &s, int local_idx, const Field &f, double *host)
{
	HIPCHK(hipSetDevice(s.device));
	const int mxh = h->d.mx[0], myh = h->d.mx[1];
	const size_t hrow = (size_t)(mxh + 3);
	const size_t hplane = hrow * (myh + 3);
	// owned planes: 0..mz, plus ghost -1 on the first local slab and mz+1 on the last
	int k0 = (local_idx == 0) ? -1 : 0;
	int k1 = (local_idx == (int)h->slabs.size() - 1) ? s.g.mz + 1 : s.g.mz;
	double *dst = host + ((size_t)local_idx * h->mzl + (k0 + 1)) * hplane;
	double *src = f.p + (long)(k0 + 1) * s.g.ps;
	const size_t rows = (size_t)(myh + 3) * (k1 - k0 + 1);
	CHK(ensure_stage(s, rows * hrow));
	HIPCHK(kry_launch_repack(src, s.stage, (int)hrow, (long)rows, s.g.px, false, s.st));
	HIPCHK(hipMemcpyAsync(dst, s.stage, rows * hrow * sizeof(double), hipMemcpyDefault, s.st));
	return 0;
}

int sync_all(kry_solver *h)
{
	for (Slab &s : h->slabs) {
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(hipStreamSynchronize(s.st));
	}
	return 0;
}

// --------------------------------------------------------------- profiling

void prof_begin(kry_solver *h, Slab &s, int kind)
{
	if (!h->profile || s.index != h->slabs[0].index) return;
	Prof &p = h->prof;
	if (p.used + 2 > p.ev.size()) {
		for (int i = 0; i < 2; ++i) {
			hipEvent_t e;
			hipEventCreate(&e);
			p.ev.push_back(e);
		}
	}
	hipEventRecord(p.ev[p.used], s.st);
	p.kind.resize(p.ev.size() / 2);
	p.kind[p.used / 2] = kind;
}

void prof_end(kry_solver *h, Slab &s)
{
	if (!h->profile || s.index != h->slabs[0].index) return;
	Prof &p = h->prof;
	hipEventRecord(p.ev[p.used + 1], s.st);
	p.used += 2;
}

void prof_collect(kry_solver *h)
{
	Prof &p = h->prof;
	for (size_t i = 0; i + 1 < p.used; i += 2) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) {
			int k = p.kind[i / 2];
			h->kernel_ms[k] += ms;
			h->kernel_calls[k] += 1;
		}
	}
	p.used = 0;
}

// ----------------------------------------------------- cross-slab plumbing

// Make the values of S->sums identical on every slab of every process: the sum
// over all slabs, added in slab order.  No-op for a single slab (the kernel's
// last block already advanced the recurrences).
int halo_wait(kry_solver *h);
int reduce_sums(kry_solver *h, int phase)
{
	const int ns = (int)h->slabs.size();
	const bool remote = h->comm.active();
	if (ns == 1 && (!remote || h->slabs[0].mail_on)) return 0;   // the kernel's last block did it all
	if (h->slab_link) return 0;                                   // ... over the slabs' mailboxes
	Slab &s0 = h->slabs[0];
	// The total is formed in a buffer of its own (the first slab's `gather` tail):
	// S->sums of a slab is rewritten by that slab's next reducing kernel, which
	// may finish before a slower slab has fetched the total.
	double *tot = (ns > 1) ? s0.gather + (size_t)KRY_MAX_SLABS * KRY_NSUMS : s0.S->sums;
	if (ns > 1) {
		for (int i = 0; i < ns; ++i) {
			Slab &s = h->slabs[i];
			HIPCHK(hipSetDevice(s.device));
			HIPCHK(hipMemcpyAsync(s0.gather + i * KRY_NSUMS, s.S->sums, KRY_NSUMS * sizeof(double),
			                      hipMemcpyDeviceToDevice, s.st));
			HIPCHK(hipEventRecord(s.ev, s.st));
		}
		HIPCHK(hipSetDevice(s0.device));
		for (int i = 1; i < ns; ++i) HIPCHK(hipStreamWaitEvent(s0.st, h->slabs[i].ev, 0));
		HIPCHK(kry_launch_sum_slabs(s0.gather, ns, tot, s0.st));
	}
	if (remote) {
		// (RCCL: the library's calls on one communicator stay in one order on one stream at a time -- a pending
		// exchange on the exchange stream is let finish first)
		if (h->comm.halo_by_rccl()) CHK(halo_wait(h));
		prof_begin(h, s0, 8);
		CHK(h->comm.allreduce_sum(tot, KRY_NSUMS, s0.st));
		prof_end(h, s0);
	}
	if (ns > 1) {
		HIPCHK(hipEventRecord(s0.ev_in, s0.st));
		for (int i = 0; i < ns; ++i) {
			Slab &s = h->slabs[i];
			HIPCHK(hipSetDevice(s.device));
			if (i > 0) HIPCHK(hipStreamWaitEvent(s.st, s0.ev_in, 0));
			HIPCHK(hipMemcpyAsync(s.S->sums, tot, KRY_NSUMS * sizeof(double), hipMemcpyDeviceToDevice, s.st));
		}
	}
	if (phase != PH_NONE) {
		for (Slab &s : h->slabs) {
			HIPCHK(hipSetDevice(s.device));
			HIPCHK(kry_launch_scalar(s.S, phase, s.st));
		}
	}
	return 0;
}

// Fill the neighbour-owned ghost planes of a stencil output: plane 1 goes to the
// lower neighbour's ghost mz+1, plane mz-1 to the upper neighbour's ghost -1
// (BoundaryHandler.C:615-618, 761-765 for direction 2).
// x and y neighbours of a Cartesian split (one sub-domain per process).  Per
// direction: planes 1 and m-1 fill the neighbours' ghosts m+1 and -1 over the full
// tangential box, ghost layers included, so edges and corners arrive in two or three
// hops as in BoundaryHandler.C:579-1045.  In x the interface plane travels too (0 ->
// the lower neighbour's mx): the fused kernels evaluate the operator on the upper
// side of an x interface only.
int exchange_xy(kry_solver *h, double *f, hipStream_t st)
{
	Slab &s = h->slabs[0];
	const KryGeo &g = s.g;
	HIPCHK(hipSetDevice(s.device));
	for (int dir = 0; dir < 2; ++dir) {
		if (!g.nb[2 * dir] && !g.nb[2 * dir + 1]) continue;
		const int m = dir == 0 ? g.mx : g.my;
		const size_t n = (size_t)((dir == 0 ? g.my : g.mx) + 3) * (g.mz + 3);
		// slots of xy_buf: [send_lo x2 | send_hi x2 | recv_lo x2 | recv_hi x2]; x uses two planes
		// per message in both directions (the second one of the upward message is padding)
		const size_t slot = s.xy_plane;
		double *send_lo = s.xy_buf, *send_hi = s.xy_buf + 2 * slot;
		double *recv_lo = s.xy_buf + 4 * slot, *recv_hi = s.xy_buf + 6 * slot;
		const int nl = dir == 0 ? 2 : 1;
		int cc[3] = {h->co[0], h->co[1], h->co[2]};
		int peer_lo = -1, peer_hi = -1;
		if (g.nb[2 * dir]) { cc[dir] -= 1; peer_lo = h->rank_of(cc[0], cc[1], cc[2]); cc[dir] += 1; }
		if (g.nb[2 * dir + 1]) { cc[dir] += 1; peer_hi = h->rank_of(cc[0], cc[1], cc[2]); cc[dir] -= 1; }
		if (peer_lo >= 0) {
			HIPCHK(kry_launch_face(f, send_lo, g, dir, 1, true, st));
			if (nl == 2) HIPCHK(kry_launch_face(f, send_lo + n, g, dir, 0, true, st));
		}
		if (peer_hi >= 0) {
			HIPCHK(kry_launch_face(f, send_hi, g, dir, m - 1, true, st));
			if (nl == 2) HIPCHK(kry_launch_face(f, send_hi + n, g, dir, m - 2, true, st));
		}
		CHK(h->comm.halo(peer_lo, send_lo, recv_lo, peer_hi, send_hi, recv_hi, (size_t)nl * n, st, dir));
		if (peer_hi >= 0) {
			HIPCHK(kry_launch_face(f, recv_hi, g, dir, m + 1, false, st));
			if (nl == 2) HIPCHK(kry_launch_face(f, recv_hi + n, g, dir, m, false, st));
		}
		if (peer_lo >= 0) HIPCHK(kry_launch_face(f, recv_lo, g, dir, -1, false, st));
	}
	return 0;
}

// Ghost planes of slab i through the link between the slabs of this process: its planes 1 and mz-1
// are stored into the neighbours' landing slots, its own slots are awaited and copied into the
// ghost planes (two kernels on `st`; slab numbers play the part of ranks).
int link_halo(kry_solver *h, int i, double *b, hipStream_t st)
{
	Slab &s = h->slabs[i];
	const KryGeo &g = s.g;
	HIPCHK(hipSetDevice(s.device));
	if (s.link.halo(i > 0 ? i - 1 : -1, b + 2L * g.ps, b, i + 1 < (int)h->slabs.size() ? i + 1 : -1,
	                b + (long)g.mz * g.ps, b + (long)(g.mz + 2) * g.ps, (size_t)g.ps, st) < 0) {
		set_err("%s", KryComm::last_error());
		return KRY_ERR_COMM;
	}
	return 0;
}

int exchange_halo(kry_solver *h, std::vector<double *> &f)
{
	const int ns = (int)h->slabs.size();
	if (ns == 1 && !h->comm.active()) return 0;
	if (h->cart) CHK(exchange_xy(h, f[0], h->slabs[0].st));
	if (h->slab_link) {
		for (int i = 0; i < ns; ++i) CHK(link_halo(h, i, f[i], h->slabs[i].st));
		return 0;
	}
	const size_t bytes = (size_t)h->slabs[0].g.ps * sizeof(double);
	for (int i = 0; i < ns; ++i) {
		HIPCHK(hipSetDevice(h->slabs[i].device));
		HIPCHK(hipEventRecord(h->slabs[i].ev, h->slabs[i].st));
	}
	for (int i = 0; i + 1 < ns; ++i) {
		Slab &lo = h->slabs[i], &hi = h->slabs[i + 1];
		const KryGeo &g = lo.g;
		// lo plane mz-1 -> hi ghost -1 ; hi plane 1 -> lo ghost mz+1
		HIPCHK(hipSetDevice(hi.device));
		HIPCHK(hipStreamWaitEvent(hi.st, lo.ev, 0));
		HIPCHK(hipMemcpyAsync(f[i + 1] + 0 * g.ps, f[i] + (long)(g.mz - 1 + 1) * g.ps, bytes,
		                      hipMemcpyDeviceToDevice, hi.st));
		HIPCHK(hipSetDevice(lo.device));
		HIPCHK(hipStreamWaitEvent(lo.st, hi.ev, 0));
		HIPCHK(hipMemcpyAsync(f[i] + (long)(g.mz + 1 + 1) * g.ps, f[i + 1] + (long)(1 + 1) * g.ps, bytes,
		                      hipMemcpyDeviceToDevice, lo.st));
	}
	if (ns > 1) {
		// a slab may go on (and update the copied planes in place) only when its neighbours have
		// read them: every stream waits for the copies queued on the neighbouring streams
		for (int i = 0; i < ns; ++i) {
			HIPCHK(hipSetDevice(h->slabs[i].device));
			HIPCHK(hipEventRecord(h->slabs[i].ev_read, h->slabs[i].st));
		}
		for (int i = 0; i < ns; ++i) {
			HIPCHK(hipSetDevice(h->slabs[i].device));
			if (i > 0) HIPCHK(hipStreamWaitEvent(h->slabs[i].st, h->slabs[i - 1].ev_read, 0));
			if (i + 1 < ns) HIPCHK(hipStreamWaitEvent(h->slabs[i].st, h->slabs[i + 1].ev_read, 0));
		}
	}
	if (h->comm.active()) {
		// with RCCL a process owns exactly one slab (checked in kry_comm_init)
		Slab &s = h->slabs[0];
		const KryGeo &g = s.g;
		double *b = f[0];
		HIPCHK(hipSetDevice(s.device));
		prof_begin(h, s, 9);
		CHK(h->comm.halo(g.has_lo ? h->rank_of(h->co[0], h->co[1], h->co[2] - 1) : -1, b + 2L * g.ps, b,
		                 g.has_hi ? h->rank_of(h->co[0], h->co[1], h->co[2] + 1) : -1, b + (long)g.mz * g.ps,
		                 b + (long)(g.mz + 2) * g.ps, (size_t)g.ps, s.st));
		prof_end(h, s);
	}
	return 0;
}

// The same exchange on the slabs' communication streams: it starts when the
// outermost chunks of the producing kernel are done (ev_edge) and overlaps with
// the chunks in between, which the caller queues on the main streams next.
// Every main stream then waits for its ghost planes (ev_halo).
int exchange_halo_begin(kry_solver *h, std::vector<double *> &f)
{
	const int ns = (int)h->slabs.size();
	const size_t bytes = (size_t)h->slabs[0].g.ps * sizeof(double);
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(hipEventRecord(s.ev_edge, s.st));
	}
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		const KryGeo &g = s.g;
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(hipStreamWaitEvent(s.st_comm, s.ev_edge, 0));
		if (h->slab_link) {
			// (the kernels wait for the neighbours' planes themselves)
			CHK(link_halo(h, i, f[i], s.st_comm));
			HIPCHK(hipEventRecord(s.ev_halo, s.st_comm));
			continue;
		}
		if (i > 0) {         // ghost -1 <- lower slab's plane mz-1
			HIPCHK(hipStreamWaitEvent(s.st_comm, h->slabs[i - 1].ev_edge, 0));
			HIPCHK(hipMemcpyAsync(f[i], f[i - 1] + (long)g.mz * g.ps, bytes, hipMemcpyDeviceToDevice, s.st_comm));
		}
		if (i + 1 < ns) {    // ghost mz+1 <- upper slab's plane 1
			HIPCHK(hipStreamWaitEvent(s.st_comm, h->slabs[i + 1].ev_edge, 0));
			HIPCHK(hipMemcpyAsync(f[i] + (long)(g.mz + 2) * g.ps, f[i + 1] + 2L * g.ps, bytes, hipMemcpyDeviceToDevice,
			                      s.st_comm));
		}
		if (h->comm.active()) {
			double *b = f[i];
			CHK(h->comm.halo(g.has_lo ? h->rank_of(h->co[0], h->co[1], h->co[2] - 1) : -1, b + 2L * g.ps, b,
			                 g.has_hi ? h->rank_of(h->co[0], h->co[1], h->co[2] + 1) : -1, b + (long)g.mz * g.ps,
			                 b + (long)(g.mz + 2) * g.ps, (size_t)g.ps, s.st_comm));
		}
		HIPCHK(hipEventRecord(s.ev_halo, s.st_comm));
	}
	return 0;
}

int exchange_halo_end(kry_solver *h)
{
	// a slab may go on once its own ghosts are filled AND its neighbours have
	// read the planes they copy from it (later kernels update those in place)
	const int ns = (int)h->slabs.size();
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		// (timed: how long the main stream sits idle until the planes are there)
		if (h->comm.active()) prof_begin(h, s, 9);
		HIPCHK(hipStreamWaitEvent(s.st, s.ev_halo, 0));
		// (through the link a slab's planes are read by its own kernel only)
		if (i > 0 && !h->slab_link) HIPCHK(hipStreamWaitEvent(s.st, h->slabs[i - 1].ev_halo, 0));
		if (i + 1 < ns && !h->slab_link) HIPCHK(hipStreamWaitEvent(s.st, h->slabs[i + 1].ev_halo, 0));
		if (h->comm.active()) prof_end(h, s);
	}
	return 0;
}

// Cartesian split (one sub-domain per process): the whole exchange of an operator output -- x faces,
// y faces, z planes, in that order, each over the full tangential box -- on the exchange stream.  The
// main stream does not wait: the kernel that consumes the output runs the tiles that read no exchanged
// cell meanwhile and waits (halo_wait) before the others (stencil_step).
int exchange_cart_begin(kry_solver *h, double *f)
{
	Slab &s = h->slabs[0];
	const KryGeo &g = s.g;
	HIPCHK(hipSetDevice(s.device));
	HIPCHK(hipEventRecord(s.ev_edge, s.st));
	HIPCHK(hipStreamWaitEvent(s.st_comm, s.ev_edge, 0));
	CHK(exchange_xy(h, f, s.st_comm));
	if (g.has_lo || g.has_hi)
		CHK(h->comm.halo(g.has_lo ? h->rank_of(h->co[0], h->co[1], h->co[2] - 1) : -1, f + 2L * g.ps, f,
		                 g.has_hi ? h->rank_of(h->co[0], h->co[1], h->co[2] + 1) : -1, f + (long)g.mz * g.ps,
		                 f + (long)(g.mz + 2) * g.ps, (size_t)g.ps, s.st_comm));
	HIPCHK(hipEventRecord(s.ev_halo, s.st_comm));
	h->halo_pending = true;
	return 0;
}

// the main stream goes on only when a pending exchange has filled the ghost cells
int halo_wait(kry_solver *h)
{
	if (!h->halo_pending) return 0;
	Slab &s = h->slabs[0];
	HIPCHK(hipSetDevice(s.device));
	HIPCHK(hipStreamWaitEvent(s.st, s.ev_halo, 0));
	h->halo_pending = false;
	return 0;
}

// The tiles of a general-boundary stencil launch that read no cell an exchange with a neighbour fills
// (ghost columns -1, mx, mx+1; ghost rows -1, my+1; ghost planes -1, mz+1): a box of tile indices.
// A tile reads columns 128 tx - 1 .. 128 tx + 129, rows 8 ty - 2 .. 8 ty + 7, planes z0 - 1 .. z1.
// Returns the number of such tiles (0: none, or no direction has a neighbour).
int inner_tiles(const KryGeo &g, const KryFused &a, int ib[6])
{
	int tx0 = g.nb[0] ? 1 : 0, tx1 = a.ntx - 1;
	while (tx1 >= tx0 && g.nb[1] && KRY_TXW * tx1 + KRY_TXW + 1 > g.mx - 1) --tx1;
	int ty0 = g.nb[2] ? 1 : 0, ty1 = a.nty - 1;
	while (ty1 >= ty0 && g.nb[3] && 8 * ty1 + 7 > g.my) --ty1;
	int c0 = 0, c1 = a.nchunk - 1;
	while (c0 <= c1 && g.has_lo && a.zlo + c0 * a.zc - 1 < 0) ++c0;
	while (c1 >= c0 && g.has_hi && std::min(a.zlo + (c1 + 1) * a.zc, a.zhi + 1) > g.mz) --c1;
	if (tx1 < tx0 || ty1 < ty0 || c1 < c0) return 0;
	ib[0] = tx0; ib[1] = tx1; ib[2] = ty0; ib[3] = ty1; ib[4] = c0; ib[5] = c1;
	return (tx1 - tx0 + 1) * (ty1 - ty0 + 1) * (c1 - c0 + 1);
}

// 1: the kernel that finishes a reduction also advances the recurrences (one slab in
// this process, and either no other process or the mailbox all-reduce between them)
int local_scalars(const kry_solver *h)
{
	if (h->slab_link) return 1;
	return (h->slabs.size() == 1 && (!h->comm.active() || h->slabs[0].mail_on)) ? 1 : 0;
}

KryRed make_red(Slab &s, int nblk, int phase)
{
	KryRed r;
	r.partial = s.partial;
	r.counter = s.counter;
	r.S = s.S;
	r.nblk = nblk;
	r.phase = phase;
	r.out = nullptr;
	r.out_sqrt = 0;
	return r;
}

// ------------------------------------------------------------ fused path

struct Tiling {
	int ntx, nty, nchunk, nblk, zlo, zhi;
};

Tiling tiling(const kry_solver *h, const Slab &s)
{
	Tiling t;
	if (h->gen) {
		t.ntx = (s.g.mx + KRY_TXW - 1) / KRY_TXW;       // columns i = 0 .. mx-1; -1, mx, mx+1 by the halo lanes
		t.nty = h->gen_y0 ? (s.g.my + 8 - 1) / 8         // rows    j = 0 .. my-1 (the others: kry_launch_bcs_rows)
		                  : (s.g.my + 3 + 8 - 1) / 8;    // rows    j = -1 .. my+1
		t.zlo = 0;
		t.zhi = s.g.mz;
	} else {
		t.ntx = (s.g.mx + KRY_TXW - 1) / KRY_TXW;       // columns i = 0 .. mx-1
		t.nty = (s.g.my + h->ty - 1) / h->ty;             // rows    j = 0 .. my-1
		t.zlo = s.g.has_lo ? 0 : 1;
		t.zhi = s.g.has_hi ? s.g.mz : s.g.mz - 1;
	}
	const int nz = t.zhi - t.zlo + 1;
	t.nchunk = (nz + h->zc - 1) / h->zc;
	t.nblk = t.ntx * t.nty * t.nchunk;
	return t;
}

KryFused fused_base(kry_solver *h, Slab &s, int phase)
{
	KryFused a;
	memset(&a, 0, sizeof a);
	Tiling t = tiling(h, s);
	a.g = s.g;
	a.op = h->op;
	a.S = s.S;
	a.lam = s.lam.p;
	a.Dxx = s.Dxx.p; a.Dyy = s.Dyy.p; a.Dzz = s.Dzz.p; a.Dxy = s.Dxy.p;
	a.RT = s.rt.p;
	a.RHS = s.rhs.p;
	a.zlo = t.zlo; a.zhi = t.zhi; a.zc = h->zc;
	a.ntx = t.ntx; a.nty = t.nty; a.nchunk = t.nchunk; a.nblk = t.nblk;
	a.ch_sel = 0;
	a.y_off = h->gen_y0 ? 0 : 1;
	a.blk_off = 0;
	a.red = make_red(s, t.nblk, phase);
	return a;
}

// prof: 1 = open the timing bracket before the launch, 2 = close it after, 3 = both
int launch_fused(kry_solver *h, Slab &s, int kind, const KryFused &a, int prof = 3)
{
	HIPCHK(hipSetDevice(s.device));
	if (prof & 1) prof_begin(h, s, kind);
	if (h->gen && h->inbc) HIPCHK(kry_launch_psi_shell(kind, a, s.st));
	if (h->gen) HIPCHK(kry_launch_fused_gen(kind, a, h->var, h->offd, h->op.exact_rcp != 0, h->inbc, s.st));
	else HIPCHK(kry_launch_fused(kind, h->ty, a, h->var, h->offd, h->op.exact_rcp != 0, s.st));
	if (prof & 2) prof_end(h, s);
	return 0;
}

// blocks of the boundary-plane dot kernel that follows a general-boundary stencil kernel
int surf_blocks(const KryGeo &g)
{
	// the item count of k_surf
	const long nx1 = (g.nb[1] ? g.mx - 1 : g.mx) + 1;
	const int nzf = (g.has_lo ? 0 : 1) + (g.has_hi ? 0 : 1);
	const int nyf = (g.nb[2] ? 0 : 1) + (g.nb[3] ? 0 : 1), nxf = (g.nb[0] ? 0 : 1) + (g.nb[1] ? 0 : 1);
	const int nk = (g.has_hi ? g.mz : g.mz - 1);
	const long nym = (g.nb[3] ? g.my : g.my - 1);
	const long total = (long)nzf * (g.mx + 1) * (g.my + 1) + (long)nyf * nk * nx1 + (long)nxf * nk * nym;
	return (int)std::max<long>(1, std::min<long>((total + 255) / 256, KRY_SURF_BLOCKS));
}

KryStream stream_base(kry_solver *h, Slab &s, int phase, int *nbx, int *nby, int min_planes)
{
	KryStream a;
	memset(&a, 0, sizeof a);
	a.g = s.g;
	a.S = s.S;
	// planes touched: every plane whose values this slab must keep current
	// (with general boundaries the ghost planes are live data too)
	const int zf = (h->gen || s.g.has_lo) ? -1 : 1;
	const int zl = (h->gen || s.g.has_hi) ? s.g.mz + 1 : s.g.mz - 1;
	a.zfirst = zf;
	a.nplanes = zl - zf + 1;
	a.cnt_lo = (h->gen && !s.g.has_lo) ? 0 : 1;
	a.cnt_hi = (h->gen || s.g.has_hi) ? s.g.mz : s.g.mz - 1;
	a.gen = h->gen ? 1 : 0;
	// segments of a plane in x, groups of planes in y
	*nbx = (int)((s.g.ps + KRY_SEG - 1) / KRY_SEG);
	// min_planes per block: the read-only kernel amortises its end-of-block reduction
	// over several planes, the store-heavy one runs best with short-lived blocks
	*nby = std::max(1, std::min(std::max(1, a.nplanes / min_planes), (h->stream_blocks + *nbx - 1) / *nbx));
	a.red = make_red(s, (*nbx) * (*nby), phase);
	return a;
}

// Launch one fused stencil kernel on every slab and fill the neighbours' ghost
// planes of its output.  With several slabs the two outermost chunks go first,
// the exchange runs on the communication streams, and the chunks in between are
// computed meanwhile.
template <class MakeArgs>
int stencil_step(kry_solver *h, int kind, MakeArgs &&make_args)
{
	const int ns = (int)h->slabs.size();
	const bool multi = ns > 1 || h->comm.active();
	std::vector<KryFused> args(ns);
	std::vector<double *> outs(ns);
	for (int i = 0; i < ns; ++i) {
		args[i] = make_args(h->slabs[i]);
		outs[i] = args[i].O;
	}
	if (h->gen) {
		// stencil on all points, then the boundary treatment of its output
		// (Krylov.C:493-495), then the boundary-plane terms of the dots
		for (int i = 0; i < ns; ++i) {
			Slab &s = h->slabs[i];
			KryFused a = args[i];
			const int nsurf = surf_blocks(s.g);
			a.red.nblk = a.nblk + nsurf;
			int ib[6];
			// (with the operator's input treated ahead of the stencil kernel -- inbc -- that treatment reads ghost
			// cells of the operands at the faces of the whole box: no tiles ahead of the exchange then)
			const int inner = (h->halo_pending && !h->inbc) ? inner_tiles(s.g, a, ib) : 0;
			if (inner > 0 && inner < a.nblk) {
				// the ghost cells of an operand are still on their way: the tiles that read none of them
				// first, the others when the exchange is done (one reduction over both launches)
				const int all = a.nblk;
				h->overlapped_exchanges += 1;
				for (int k = 0; k < 6; ++k) a.ib[k] = ib[k];
				a.ch_sel = 3;
				a.nblk = inner;
				CHK(launch_fused(h, s, kind, a, 1));
				CHK(halo_wait(h));
				a.ch_sel = 4;
				a.nblk = all - inner;
				CHK(launch_fused(h, s, kind, a, 0));
			} else {
				CHK(halo_wait(h));
				CHK(launch_fused(h, s, kind, a, 1));
			}
			// the x and y faces of the output (the z faces follow in k_surf)
			if (h->gen_y0) HIPCHK(kry_launch_bcs_rows(kind, a, 8 * a.nty, h->inbc, h->var, h->offd, h->op.exact_rcp != 0, s.st));
			else HIPCHK(kry_launch_bcs(a.O, s.g, s.st, 2));
		}
		// planes 1 and mz-1, the ones the z neighbours need, are final once the x and y faces
		// are treated: their exchange runs beside k_surf (z faces, boundary-plane dot terms)
		const bool beside = multi && h->overlap && !h->cart;
		if (beside) CHK(exchange_halo_begin(h, outs));
		for (int i = 0; i < ns; ++i) {
			Slab &s = h->slabs[i];
			KryFused a = args[i];
			const int nsurf = surf_blocks(s.g);
			a.red.nblk = a.nblk + nsurf;
			a.blk_off = a.nblk;
			HIPCHK(hipSetDevice(s.device));
			HIPCHK(kry_launch_surf(kind, a, nsurf, s.st));
			prof_end(h, s);
		}
		if (beside) CHK(exchange_halo_end(h));
		else if (multi && h->cart && h->overlap && !h->comm.halo_blocks_host()) CHK(exchange_cart_begin(h, outs[0]));
		else if (multi) CHK(exchange_halo(h, outs));
		return 0;
	}
	CHK(halo_wait(h));
	// the neighbours need planes 1 and mz-1: chunk 0 and the chunk holding plane mz-1
	// (the last one, or the one before it when the last chunk is plane mz alone)
	std::vector<int> n_edge(ns);
	bool split = multi && h->overlap;
	for (int i = 0; i < ns; ++i) {
		const KryFused &a = args[i];
		const int c_hi = (a.g.mz - 1 - a.zlo) / a.zc;
		n_edge[i] = (c_hi == a.nchunk - 1) ? 2 : 3;
		split = split && a.nchunk > n_edge[i];
	}
	if (!split) {
		for (int i = 0; i < ns; ++i) CHK(launch_fused(h, h->slabs[i], kind, args[i]));
		if (multi) CHK(exchange_halo(h, outs));
		return 0;
	}
	for (int i = 0; i < ns; ++i) {
		KryFused a = args[i];
		const int tiles = a.ntx * a.nty;
		a.ch_sel = 1;
		a.n_edge = n_edge[i];
		a.nblk = n_edge[i] * tiles;
		a.blk_off = 0;
		CHK(launch_fused(h, h->slabs[i], kind, a, 1));
	}
	CHK(exchange_halo_begin(h, outs));
	for (int i = 0; i < ns; ++i) {
		KryFused a = args[i];
		const int tiles = a.ntx * a.nty;
		a.ch_sel = 2;
		a.n_edge = n_edge[i];
		a.nblk = (a.nchunk - n_edge[i]) * tiles;
		a.blk_off = n_edge[i] * tiles;
		CHK(launch_fused(h, h->slabs[i], kind, a, 2));
	}
	CHK(exchange_halo_end(h));
	return 0;
}

// One outer pass for l > 2, solveLin_BICGStab.C:341-464 statement by statement.  Per BiCG
// step jj: an A kernel (u_jj = r_jj - beta*u_jj ; u_jj+1 = A u_jj ; (u_jj+1, r~)), one
// pointwise sweep for the statements on the vectors below jj and phi, a B kernel
// (r_jj -= alpha*u_jj+1 ; r_jj+1 = A r_jj ; (r_jj+1, r~)).  Then the modified Gram-Schmidt
// part, one streaming kernel per dot product (the update a dot makes possible rides in the
// next kernel), and the end-of-pass kernel.
// Buffers: every r_k and u_k, k < l, alternates between its two copies with the pass: the
// stencil kernel that produces it (k > 0) writes over the latest version, copy p, as the
// reference's one array is overwritten in place (the ghost rows it leaves alone feed the
// x-face treatment of the output); the update of step jj = k moves it to copy q, where the
// later pointwise statements keep it.  r_l, u_l: one copy.
int fused_pass_general(kry_solver *h, int it)
{
	const int p = it & 1;
	const int L = h->L;
	int cr[KRY_LMAX + 1], cu[KRY_LMAX + 1];
	for (int k = 0; k <= L; ++k) cr[k] = cu[k] = (k < L) ? p : 0;
	auto flip = [&](int *c, int k) { c[k] = 1 - p; };
	auto for_slabs = [&](auto &&fn) -> int {
		CHK(halo_wait(h));      // the streaming kernels read and update ghost cells too
		for (size_t i = 0; i < h->slabs.size(); ++i) CHK(fn(h->slabs[i], (int)i));
		return 0;
	};

	for (int jj = 0; jj < L; ++jj) {
		{
			const int from = cu[jj];
			flip(cu, jj);
			CHK(stencil_step(h, KF_A1, [&](Slab &s) {
				KryFused a = fused_base(h, s, PH_A);
				a.X = s.rb[jj][cr[jj]].p; a.Y = s.ub[jj][from].p; a.Xout = s.ub[jj][cu[jj]].p;
				a.O = s.ub[jj + 1][cu[jj + 1]].p;
				return a;
			}));
			CHK(reduce_sums(h, PH_A));
		}
		// :356-362 and :392-394 for ii < jj, and :403 of the step before (the last step's rides in the end-of-pass kernel)
		if (jj > 0) {
			CHK(for_slabs([&](Slab &s, int) -> int {
				int nbx, nby;
				KryStream a = stream_base(h, s, PH_NONE, &nbx, &nby, 1);
				KryPointN f;
				memset(&f, 0, sizeof f);
				f.jj = jj;
				for (int ii = 0; ii < jj; ++ii) f.r[ii] = s.rb[ii][cr[ii]].p;
				for (int ii = 0; ii <= jj; ++ii) f.u[ii] = s.ub[ii][cu[ii]].p;
				a.w[0] = s.phi.p;
				HIPCHK(hipSetDevice(s.device));
				prof_begin(h, s, 6);
				HIPCHK(kry_launch_bicgn(a, f, nbx, nby, s.st));
				prof_end(h, s);
				return 0;
			}));
		}
		{
			const int from = cr[jj];
			flip(cr, jj);
			const int phase = (jj < L - 1) ? PH_B : PH_BN;
			CHK(stencil_step(h, KF_B0, [&](Slab &s) {
				KryFused a = fused_base(h, s, phase);
				a.rev = 1;
				a.X = s.rb[jj][from].p; a.Y = s.ub[jj + 1][cu[jj + 1]].p; a.Xout = s.rb[jj][cr[jj]].p;
				a.O = s.rb[jj + 1][cr[jj + 1]].p;
				return a;
			}));
			CHK(reduce_sums(h, phase));
		}
	}

	// MR part, :419-446
	for (int jj = 1; jj <= L; ++jj) {
		for (int step = 1; step <= jj; ++step) {
			// step < jj: the dot (r_jj, r_step) after the update with r_step-1; step = jj: sigma_jj, (r_0, r_jj)
			const int last = (step == jj), iip = step - 1;
			const int phase = last ? KRY_PHASE(PH_SIG, jj, 0) : KRY_PHASE(PH_TAU, jj, step);
			CHK(for_slabs([&](Slab &s, int) -> int {
				int nbx, nby;
				KryStream a = stream_base(h, s, phase, &nbx, &nby, 4);
				a.v[0] = s.rb[jj][cr[jj]].p;
				a.v[1] = iip > 0 ? s.rb[iip][cr[iip]].p : nullptr;
				a.v[2] = last ? s.rb[0][cr[0]].p : s.rb[step][cr[step]].p;
				a.w[0] = s.rb[jj][cr[jj]].p;
				HIPCHK(hipSetDevice(s.device));
				prof_begin(h, s, 6);
				HIPCHK(kry_launch_mgsn(a, jj, iip, last, nbx, nby, s.st));
				prof_end(h, s);
				return 0;
			}));
			CHK(reduce_sums(h, phase));
		}
	}

	// :450-464
	CHK(for_slabs([&](Slab &s, int) -> int {
		int nbx, nby;
		KryStream a = stream_base(h, s, PH_F, &nbx, &nby, 1);
		KryFinalN f;
		memset(&f, 0, sizeof f);
		f.L = L;
		for (int k = 0; k <= L; ++k) { f.r[k] = s.rb[k][cr[k]].p; f.u[k] = s.ub[k][cu[k]].p; }
		f.r0 = s.rb[0][cr[0]].p;
		f.u0 = s.ub[0][cu[0]].p;
		a.v[0] = s.rt.p;
		a.w[0] = s.phi.p;
		HIPCHK(hipSetDevice(s.device));
		prof_begin(h, s, 7);
		HIPCHK(kry_launch_finaln(a, f, nbx, nby, s.st));
		prof_end(h, s);
		return 0;
	}));
	CHK(reduce_sums(h, PH_F));
	return 0;
}

// one outer pass, solveLin_BICGStab.C:327-496, as fused kernels
int fused_pass(kry_solver *h, int it)
{
	if (h->L > 2) return fused_pass_general(h, it);
	const int p = it & 1, q = 1 - p;
	const int L = h->L;
	// Order of the work inside a launch: every kernel starts where the one before it ended (the
	// memory-side cache still holds the end of its data), so the kernels of the run alternate
	// between ascending and descending order.  k: place of the kernel in the pass.
	const int nk = (L == 1) ? 3 : 6;
	auto rev_of = [&](int k, bool b_kernel) { return h->alt ? ((it * nk + k) & 1) : (b_kernel ? 1 : 0); };
	auto for_slabs = [&](auto &&fn) -> int {
		CHK(halo_wait(h));      // the streaming kernels read and update ghost cells too
		for (size_t i = 0; i < h->slabs.size(); ++i) CHK(fn(h->slabs[i], (int)i));
		return 0;
	};

	// u1 lives in ub[1][.]: l = 1 alternates the two buffers with the pass (the A kernel
	// reads the old u1 from one and writes the new one to the other); l = 2 writes the
	// operator output over the latest u1 (copy p, as the reference's one array: its ghost rows
	// feed the x-face treatment of the output) and the updated u1 (A kernel of jj = 1) to copy q
	// --- jj = 0, A: u0 -= gamma_L*u_L [- gamma_1*u_1] (previous pass) ; u0 = r0 - beta*u0 ; u1 = A u0 ; (u1, r~)
	CHK(stencil_step(h, KF_A0, [&](Slab &s) {
		KryFused a = fused_base(h, s, PH_A);
		a.rev = rev_of(0, false);
		a.X = s.rb[0][p].p; a.Y = s.ub[0][p].p; a.Xout = s.ub[0][q].p;
		if (L == 1) { a.nz = 1; a.Z1 = s.ub[1][p].p; a.O = s.ub[1][q].p; }
		else { a.nz = 0; a.O = s.ub[1][p].p; }
		return a;
	}));
	CHK(reduce_sums(h, PH_A));

	if (L == 1) {
		// --- B (last): r0 -= alpha*u1 ; r1 = A r0 ; (r0,r1), (r1,r1)
		CHK(stencil_step(h, KF_B0L, [&](Slab &s) {
			KryFused a = fused_base(h, s, PH_BL1);
			a.rev = rev_of(1, true);
			a.X = s.rb[0][p].p; a.Y = s.ub[1][q].p; a.Xout = s.rb[0][q].p;
			a.O = s.rb[1][0].p;
			return a;
		}));
		CHK(reduce_sums(h, PH_BL1));
	} else {
		// --- jj = 0, B: r0 -= alpha*u1 ; r1 = A r0 ; (r1, r~)
		CHK(stencil_step(h, KF_B0, [&](Slab &s) {
			KryFused a = fused_base(h, s, PH_B);
			a.rev = rev_of(1, true);
			a.X = s.rb[0][p].p; a.Y = s.ub[1][p].p; a.Xout = s.rb[0][q].p;
			a.O = s.rb[1][p].p;
			return a;
		}));
		CHK(reduce_sums(h, PH_B));
		// --- jj = 1, A: u1 = r1 - beta*u1 ; u2 = A u1 ; (u2, r~)
		CHK(stencil_step(h, KF_A1, [&](Slab &s) {
			KryFused a = fused_base(h, s, PH_A);
			a.rev = rev_of(2, false);
			a.X = s.rb[1][p].p; a.Y = s.ub[1][p].p; a.Xout = s.ub[1][q].p;
			a.O = s.ub[2][0].p;
			return a;
		}));
		CHK(reduce_sums(h, PH_A));
		// --- jj = 1, B (last): r1 -= alpha*u2 ; r2 = A r1 ; (r1,r1), (r2,r1)
		CHK(stencil_step(h, KF_B1L, [&](Slab &s) {
			KryFused a = fused_base(h, s, PH_BL2);
			a.rev = rev_of(3, true);
			a.X = s.rb[1][p].p; a.Y = s.ub[2][0].p; a.Xout = s.rb[1][q].p;
			a.O = s.rb[2][0].p;
			return a;
		}));
		CHK(reduce_sums(h, PH_BL2));
		// --- MR: r2 -= tau12*r1 ; (r2,r2), (r0,r2), and (r0,r1) of jj = 1
		CHK(for_slabs([&](Slab &s, int) -> int {
			int nbx, nby;
			KryStream a = stream_base(h, s, PH_M2, &nbx, &nby, 4);
			a.rev = rev_of(4, false);
			a.v[0] = s.rb[0][q].p; a.v[1] = s.rb[1][q].p; a.v[2] = s.rb[2][0].p; a.v[3] = s.ub[1][q].p;
			a.w[0] = s.rb[2][0].p;
			HIPCHK(hipSetDevice(s.device));
			prof_begin(h, s, 6);
			HIPCHK(kry_launch_mgs2(a, nbx, nby, s.st));
			prof_end(h, s);
			return 0;
		}));
		CHK(reduce_sums(h, PH_M2));
	}

	// --- end of pass: phi and r0 updates (l = 2: also the pointwise statements of jj = 1) ; |r0|^2, (r0, r~)
	CHK(for_slabs([&](Slab &s, int) -> int {
		int nbx, nby;
		KryStream a = stream_base(h, s, PH_F, &nbx, &nby, 1);
		a.rev = rev_of(nk - 1, false);
		if (L == 1) {
			a.v[0] = s.rb[0][q].p; a.v[1] = s.rb[1][0].p;
			a.v[2] = s.ub[0][q].p;
			a.v[3] = s.rt.p;
		} else {
			a.v[0] = s.rb[0][q].p; a.v[1] = s.rb[1][q].p; a.v[2] = s.rb[2][0].p;
			a.v[3] = s.ub[0][q].p;
			a.v[4] = s.rt.p;
			a.v[5] = s.ub[1][q].p;
			a.v[6] = s.ub[2][0].p;
			a.w[2] = s.ub[0][q].p;
			a.w[3] = s.rb[2][0].p;
		}
		a.w[0] = s.phi.p; a.w[1] = s.rb[0][q].p;
		HIPCHK(hipSetDevice(s.device));
		prof_begin(h, s, 7);
		HIPCHK(kry_launch_final(L, a, nbx, nby, s.st));
		prof_end(h, s);
		return 0;
	}));
	CHK(reduce_sums(h, PH_F));
	return 0;
}

// fetch iter/done/status/norm of the first slab (identical on all)
int poll_state(kry_solver *h)
{
	Slab &s = h->slabs[0];
	HIPCHK(hipSetDevice(s.device));
	HIPCHK(hipMemcpyAsync(s.Sh, s.S, offsetof(KryState, hist), hipMemcpyDeviceToHost, s.st));
	HIPCHK(hipStreamSynchronize(s.st));
	h->it_base = s.Sh->iter;
	h->done = s.Sh->done != 0;
	h->status = s.Sh->status;
	h->norm_final = s.Sh->norm;
	h->norm_init = s.Sh->norm_init;
	return 0;
}

// ---------------------------------------------------------- generic path

struct Gen {
	kry_solver *h;
	int ns() const { return (int)h->slabs.size(); }
};

int g_do_bcs(kry_solver *h, Field Slab::*f)
{
	for (Slab &s : h->slabs) {
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_bcs((s.*f).p, s.g, s.st));
	}
	return 0;
}

int g_do_bcs_p(kry_solver *h, std::vector<double *> &f)
{
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_bcs(f[i], s.g, s.st));
	}
	return 0;
}

// weighted dot over all slabs -> host value
int g_dot(kry_solver *h, std::vector<double *> &x, std::vector<double *> &y, int norm_mode, double *out)
{
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_wdot(s.g, x[i], y[i], norm_mode, make_red(s, 0, PH_NONE), s.st));
	}
	CHK(reduce_sums(h, PH_NONE));
	Slab &s0 = h->slabs[0];
	HIPCHK(hipSetDevice(s0.device));
	HIPCHK(hipMemcpyAsync(s0.Sh->sums, s0.S->sums, sizeof(double), hipMemcpyDeviceToHost, s0.st));
	HIPCHK(hipStreamSynchronize(s0.st));
	*out = s0.Sh->sums[0];
	return 0;
}

std::vector<double *> ptrs(kry_solver *h, Field Slab::*f)
{
	std::vector<double *> v;
	for (Slab &s : h->slabs) v.push_back((s.*f).p);
	return v;
}
std::vector<double *> ptrs_r(kry_solver *h, int k)
{
	std::vector<double *> v;
	for (Slab &s : h->slabs) v.push_back(s.rb[k][0].p);
	return v;
}
std::vector<double *> ptrs_u(kry_solver *h, int k)
{
	std::vector<double *> v;
	for (Slab &s : h->slabs) v.push_back(s.ub[k][0].p);
	return v;
}

// dst = x + y*c on every slab (whole stored box)
int g_axpy(kry_solver *h, std::vector<double *> &dst, std::vector<double *> &x, std::vector<double *> &y, double c)
{
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_axpy(dst[i], x[i], y[i], c, s.g.n, s.st));
	}
	return 0;
}

// Krylov::multiply_withMat incl. its boundary handling (Krylov.C:500-541 / 407-496)
int g_spmv(kry_solver *h, std::vector<double *> &psi, std::vector<double *> &out, bool apply_bcs)
{
	if (!h->var) CHK(g_do_bcs_p(h, psi));      // Krylov.C:516
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_gspmv(s.g, h->op, h->var, h->offd, psi[i], s.lam.p, s.Dxx.p, s.Dyy.p, s.Dzz.p,
		                        s.Dxy.p, out[i], s.st));
	}
	if (apply_bcs) CHK(g_do_bcs_p(h, out));
	CHK(exchange_halo(h, out));
	return 0;
}

double g_norm_from(double sumsq) { return std::sqrt(sumsq); }

bool host_really_is_nan(double xd)
{
	float x = (float)xd;
	unsigned bits;
	memcpy(&bits, &x, sizeof bits);
	return (bits & 0x7F800000u) == 0x7F800000u;
}

// Krylov::get_Residual (Krylov.C:248-306, 311-402) into r[0]
int g_residual(kry_solver *h)
{
	auto phi = ptrs(h, &Slab::phi), rhs = ptrs(h, &Slab::rhs), r0 = ptrs_r(h, 0);
	CHK(g_spmv(h, phi, r0, false));
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_sub(r0[i], rhs[i], r0[i], s.g.n, s.st));
	}
	CHK(g_do_bcs_p(h, r0));
	CHK(exchange_halo(h, r0));
	return 0;
}

// one outer pass with host scalars, statement by statement as
// solveLin_BICGStab.C:341-496
int generic_pass(kry_solver *h, double *sigma, double *gammap, double *gamma, double *gammapp,
                 double (*tau)[KRY_LMAX + 1])
{
	const int L = h->L;
	auto rt = ptrs(h, &Slab::rt), phi = ptrs(h, &Slab::phi);
	std::vector<std::vector<double *>> r(L + 1), u(L + 1);
	for (int k = 0; k <= L; ++k) { r[k] = ptrs_r(h, k); u[k] = ptrs_u(h, k); }

	h->g_rho0 *= -h->g_omega;
	for (int jj = 0; jj < L; ++jj) {
		double rho1;
		CHK(g_dot(h, r[jj], rt, 0, &rho1));
		double beta = h->g_alpha * rho1 / h->g_rho0;
		h->g_rho0 = rho1;
		for (int ii = 0; ii <= jj; ++ii) CHK(g_axpy(h, u[ii], r[ii], u[ii], -beta));
		CHK(g_spmv(h, u[jj], u[jj + 1], true));
		double d;
		CHK(g_dot(h, u[jj + 1], rt, 0, &d));
		h->g_alpha = h->g_rho0 / d;
		for (int ii = 0; ii <= jj; ++ii) CHK(g_axpy(h, r[ii], r[ii], u[ii + 1], -h->g_alpha));
		CHK(g_spmv(h, r[jj], r[jj + 1], true));
		CHK(g_axpy(h, phi, phi, u[0], h->g_alpha));
		if (host_really_is_nan(h->g_alpha)) {
			if (h->debug > 0) printf(" alpha is nan %d %g %g %g\n", jj, h->g_rho0, rho1, beta);
			h->status = KRY_NAN_ALPHA;
			h->done = true;
			return 0;
		}
	}
	for (int jj = 1; jj <= L; ++jj) {
		for (int ii = 1; ii < jj; ++ii) {
			double d;
			CHK(g_dot(h, r[jj], r[ii], 0, &d));
			tau[ii][jj] = d / sigma[ii];
			CHK(g_axpy(h, r[jj], r[jj], r[ii], -tau[ii][jj]));
		}
		CHK(g_dot(h, r[jj], r[jj], 0, &sigma[jj]));
		double d;
		CHK(g_dot(h, r[0], r[jj], 0, &d));
		gammap[jj] = d / sigma[jj];
	}
	h->g_omega = gamma[L] = gammap[L];
	for (int jj = L - 1; jj >= 1; --jj) {
		gamma[jj] = gammap[jj];
		for (int ii = jj + 1; ii <= L; ++ii) gamma[jj] -= tau[jj][ii] * gamma[ii];
	}
	for (int jj = 1; jj < L; ++jj) {
		gammapp[jj] = gamma[jj + 1];
		for (int ii = jj + 1; ii < L; ++ii) gammapp[jj] += tau[jj][ii] * gamma[ii + 1];
	}
	CHK(g_axpy(h, phi, phi, r[0], gamma[1]));
	CHK(g_axpy(h, r[0], r[0], r[L], -gammap[L]));
	CHK(g_axpy(h, u[0], u[0], u[L], -gamma[L]));
	for (int jj = 1; jj < L; ++jj) {
		CHK(g_axpy(h, u[0], u[0], u[jj], -gamma[jj]));
		CHK(g_axpy(h, phi, phi, r[jj], gammapp[jj]));
		CHK(g_axpy(h, r[0], r[0], r[jj], -gammap[jj]));
	}
	double ss;
	CHK(g_dot(h, r[0], r[0], 1, &ss));
	double norm = g_norm_from(ss);
	h->norm_final = norm;
	h->hist.push_back(norm);
	h->it_base += 1;
	if (host_really_is_nan(norm)) {
		if (h->debug > 0) printf(" Norm is nan at iteration %d\n", h->it_base);
		h->status = KRY_NAN_NORM;
		h->done = true;
		return 0;
	}
	if (!(norm > h->d.eps * h->norm_rhs)) { h->done = true; return 0; }
	if (h->d.max_outer > 0 && h->it_base >= h->d.max_outer) { h->status = KRY_MAX_ITER; h->done = true; }
	return 0;
}

// --------------------------------------------------------- setup helpers

bool is_pow2(double v)
{
	int e;
	return std::frexp(v, &e) == 0.5;
}

void make_op(kry_solver *h, bool var, const double Dc[3])
{
	KryOp &op = h->op;
	const double *dx = h->d.delx;
	for (int d = 0; d < 3; ++d) {
		op.c[d] = var ? 1. / (dx[d] * dx[d]) : Dc[d] / (dx[d] * dx[d]);   // Krylov.C:424 / 513
		op.two_d[d] = 2. * dx[d];
		op.inv_two_d[d] = 1. / op.two_d[d];
	}
	op.cd2 = 2. * (op.c[0] + op.c[1] + op.c[2]);
	op.cxy = 1. / (2. * dx[0] * dx[1]);
	op.exact_rcp = (is_pow2(op.two_d[0]) && is_pow2(op.two_d[1]) && is_pow2(op.two_d[2])) ? 1 : 0;
}

int stage_common(kry_solver *h, const double *phi, const double *rhs, const double *lam,
                 const double *Dxx, const double *Dyy, const double *Dzz, const double *Dxy)
{
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		CHK(upload(h, s, (int)i, s.phi, phi));
		CHK(upload(h, s, (int)i, s.rhs, rhs));
		CHK(upload(h, s, (int)i, s.lam, lam));
		if (h->var) {
			CHK(upload(h, s, (int)i, s.Dxx, Dxx));
			CHK(upload(h, s, (int)i, s.Dyy, Dyy));
			CHK(upload(h, s, (int)i, s.Dzz, Dzz));
			if (h->offd) CHK(upload(h, s, (int)i, s.Dxy, Dxy));
		}
	}
	return 0;
}

// zero a field's interior + ghosts (keeps guard zones zero as well)
int zero_field(Slab &s, Field &f)
{
	HIPCHK(hipSetDevice(s.device));
	HIPCHK(hipMemsetAsync(f.p, 0, (size_t)s.g.n * sizeof(double), s.st));
	return 0;
}

// prologue of solve_int (solveLin_BICGStab.C:268-320)
int prologue(kry_solver *h)
{
	h->it_base = 0;
	h->status = 0;
	h->done = false;
	h->tail_bc_done = false;
	h->hist.clear();
	h->step_ms.clear();
	h->g_rho0 = 1.; h->g_alpha = 0.; h->g_omega = 1.;

	// the reference allocates zeroed work vectors once (make_Arrays) and starts
	// the recurrences with u = 0 via beta = 0; start from zeros on every load
	for (Slab &s : h->slabs) {
		for (int k = 0; k <= h->L; ++k)
			for (int b = 0; b < NPINGPONG; ++b) {
				if (s.rb[k][b].p) CHK(zero_field(s, s.rb[k][b]));
				if (s.ub[k][b].p) CHK(zero_field(s, s.ub[k][b]));
			}
		CHK(zero_field(s, s.rt));
	}

	// normRHS = l2(rhs), zero -> one (:268-269)
	auto rhs = ptrs(h, &Slab::rhs);
	double ss;
	CHK(g_dot(h, rhs, rhs, 1, &ss));
	h->norm_rhs = std::sqrt(ss);
	if (h->norm_rhs == 0.) h->norm_rhs = 1.;

	if (h->fused && h->gen) {
		// residual, its norm and (r0, r~) with the one-kernel-per-primitive path, then
		// the head of the first pass on the host (:301-320, :341, :349-352 with jj = 0)
		CHK(g_residual(h));
		auto r0 = ptrs_r(h, 0), rt = ptrs(h, &Slab::rt);
		double rho1;
		CHK(g_dot(h, r0, r0, 1, &ss));
		CHK(g_dot(h, r0, r0, 0, &rho1));
		KryState init;
		memset(&init, 0, offsetof(KryState, hist));
		init.eps = h->d.eps;
		init.norm_rhs = h->norm_rhs;
		init.max_outer = h->d.max_outer;
		init.L = h->L;
		init.local_scalars = local_scalars(h);
		init.norm_init = std::sqrt(ss);
		init.norm = 1.e99;
		if (init.norm_init < h->d.eps * h->norm_rhs) {
			init.norm = init.norm_init;
			init.done = 1;
		} else {
			double rho0 = 1., alpha = 0., omega = 1.;
			rho0 = rho0 * (-omega);
			init.beta = alpha * rho1 / rho0;
			init.rho0 = rho1;
			init.rho1 = rho1;
			init.alpha = alpha;
			init.omega = omega;
		}
		for (size_t i = 0; i < h->slabs.size(); ++i) {
			Slab &s = h->slabs[i];
			HIPCHK(hipSetDevice(s.device));
			HIPCHK(hipMemcpyAsync(rt[i], r0[i], (size_t)s.g.n * sizeof(double), hipMemcpyDeviceToDevice, s.st));
			init.mail = s.mail_on ? s.mail_dev : nullptr;
			memcpy(s.Sh, &init, offsetof(KryState, hist));
			HIPCHK(hipMemcpyAsync(s.S, s.Sh, offsetof(KryState, hist), hipMemcpyHostToDevice, s.st));
			HIPCHK(hipStreamSynchronize(s.st));
		}
		CHK(poll_state(h));
	} else if (h->fused) {
		for (Slab &s : h->slabs) {
			HIPCHK(hipSetDevice(s.device));
			KryState init;
			memset(&init, 0, offsetof(KryState, hist));
			init.eps = h->d.eps;
			init.norm_rhs = h->norm_rhs;
			init.max_outer = h->d.max_outer;
			init.L = h->L;
			init.local_scalars = local_scalars(h);
			init.mail = s.mail_on ? s.mail_dev : nullptr;
			memcpy(s.Sh, &init, offsetof(KryState, hist));
			HIPCHK(hipMemcpyAsync(s.S, s.Sh, offsetof(KryState, hist), hipMemcpyHostToDevice, s.st));
			HIPCHK(hipStreamSynchronize(s.st));
		}
		// constant-coefficient operator: do_BCs on its input (Krylov.C:516)
		if (!h->var) CHK(g_do_bcs(h, &Slab::phi));
		std::vector<double *> outs(h->slabs.size()), outs2(h->slabs.size());
		for (size_t i = 0; i < h->slabs.size(); ++i) {
			Slab &s = h->slabs[i];
			KryFused a = fused_base(h, s, PH_INIT);
			a.X = s.phi.p;
			a.O = s.rb[0][0].p;
			a.O2 = s.rt.p;
			outs[i] = a.O;
			outs2[i] = a.O2;
			CHK(launch_fused(h, s, KF_RES, a));
		}
		CHK(exchange_halo(h, outs));
		CHK(exchange_halo(h, outs2));
		CHK(reduce_sums(h, PH_INIT));
		CHK(poll_state(h));
	} else {
		CHK(g_residual(h));
		auto r0 = ptrs_r(h, 0), rt = ptrs(h, &Slab::rt);
		CHK(g_dot(h, r0, r0, 1, &ss));
		h->norm_init = std::sqrt(ss);
		h->norm_final = 1.e99;
		if (h->norm_init < h->d.eps * h->norm_rhs) {
			h->done = true;
			h->norm_final = h->norm_init;
		} else {
			for (size_t i = 0; i < h->slabs.size(); ++i) {
				Slab &s = h->slabs[i];
				HIPCHK(hipSetDevice(s.device));
				HIPCHK(hipMemcpyAsync(rt[i], r0[i], (size_t)s.g.n * sizeof(double), hipMemcpyDeviceToDevice, s.st));
			}
		}
	}
	return 0;
}

void drop_graph(kry_solver *h)
{
	if (h->pass_pair) {
		hipSetDevice(h->slabs[0].device);
		hipGraphExecDestroy(h->pass_pair);
	}
	h->pass_pair = nullptr;
}

// Capture passes 0 and 1 of the loaded problem (kernel arguments depend on the pass
// parity only; all scalars live in device memory).  Single slab, no profiling events.
int build_graph(kry_solver *h)
{
	Slab &s = h->slabs[0];
	HIPCHK(hipSetDevice(s.device));
	hipGraph_t g = nullptr;
	HIPCHK(hipStreamBeginCapture(s.st, hipStreamCaptureModeThreadLocal));
	int st = fused_pass(h, 0);
	if (st == 0) st = fused_pass(h, 1);
	hipError_t e = hipStreamEndCapture(s.st, &g);
	if (st != 0 || e != hipSuccess || !g) {
		if (g) hipGraphDestroy(g);
		(void)hipGetLastError();
		h->use_graph = false;      // run with plain launches
		return 0;
	}
	e = hipGraphInstantiate(&h->pass_pair, g, nullptr, nullptr, 0);
	hipGraphDestroy(g);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		h->pass_pair = nullptr;
		h->use_graph = false;
	}
	return 0;
}

int run_iterations(kry_solver *h, int n_outer)
{
	if (!h->loaded) { set_err("kry_iterate: no problem loaded"); return KRY_ERR_STATE; }
	if (h->reset_needed) { set_err("kry_iterate: fields were replaced, call kry_reset or kry_resolve first"); return KRY_ERR_STATE; }
	if (h->done) return h->status;
	hipEvent_t e0, e1;
	Slab &s0 = h->slabs[0];
	HIPCHK(hipSetDevice(s0.device));
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	HIPCHK(hipEventRecord(e0, s0.st));
	int left = n_outer;   // <= 0: until done
	if (h->fused) {
		// launch in batches; kernels past convergence are no-ops on the device
		const int batch_max = (h->debug > 2) ? 1 : 8;
		while (!h->done && (n_outer <= 0 || left > 0)) {
			int nb = (n_outer <= 0) ? batch_max : std::min(batch_max, left);
			const int it0 = h->it_base;
			double t0 = now_ms();
			const bool graph_ok = h->use_graph && !h->profile && h->slabs.size() == 1 && !h->comm.active() &&
			                      h->debug <= 2;
			if (graph_ok && !h->pass_pair) CHK(build_graph(h));
			for (int i = 0; i < nb;) {
				if (graph_ok && h->pass_pair && ((it0 + i) & 1) == 0 && i + 1 < nb) {
					HIPCHK(hipGraphLaunch(h->pass_pair, h->slabs[0].st));
					i += 2;
				} else {
					CHK(fused_pass(h, it0 + i));
					i += 1;
				}
			}
			h->enqueue_ms += now_ms() - t0;
			CHK(poll_state(h));
			{
				const int ran = std::max(1, h->it_base - it0);
				const double each = (now_ms() - t0) / ran;
				for (int i = 0; i < h->it_base - it0; ++i) h->step_ms.push_back(each);
			}
			if (h->debug > 2) {
				Slab &s = h->slabs[0];
				(void)s;
				printf(" Iteration: %d (%.3f ms)\n", h->it_base, now_ms() - t0);
				printf(" My error norm: %g %g %g %g %g \n", h->norm_final, std::sqrt(h->norm_final),
				       h->d.eps * h->norm_rhs, h->d.eps, h->norm_rhs);
			}
			if (n_outer > 0) left -= nb;
			if (!h->done && h->it_base != it0 + nb) {
				set_err("device iteration counter %d != host %d", h->it_base, it0 + nb);
				return KRY_ERR_STATE;
			}
		}
	} else {
		double sigma[KRY_LMAX + 1] = {0}, gammap[KRY_LMAX + 1] = {0}, gamma[KRY_LMAX + 1] = {0},
		       gammapp[KRY_LMAX + 1] = {0};
		static thread_local double tau[KRY_LMAX + 1][KRY_LMAX + 1];
		while (!h->done && (n_outer <= 0 || left > 0)) {
			double t0 = now_ms();
			CHK(generic_pass(h, sigma, gammap, gamma, gammapp, tau));
			h->step_ms.push_back(now_ms() - t0);
			if (h->debug > 2) {
				printf(" Iteration: %d (%.3f ms)\n", h->it_base, now_ms() - t0);
				printf(" My error norm: %g %g %g %g %g \n", h->norm_final, std::sqrt(h->norm_final),
				       h->d.eps * h->norm_rhs, h->d.eps, h->norm_rhs);
			}
			if (n_outer > 0) left -= 1;
		}
	}
	CHK(halo_wait(h));
	HIPCHK(hipSetDevice(s0.device));
	HIPCHK(hipEventRecord(e1, s0.st));
	CHK(sync_all(h));
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	h->solve_ms += ms;
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	if (h->profile) prof_collect(h);
	return h->status;
}

// Planes per chunk of the fused stencil kernels.  All blocks take equally long and a compute unit
// that is left with one block draws only three quarters of its bandwidth share, so the block count
// should fill the device's block slots a whole number of times.  Among the counts that do, the
// smallest wins (the deepest chunks: fewer halo planes re-read, fewer block prologues, and the
// blocks of a launch sweep the arrays in fewer fronts -- 512^3: 2 chunks of 256 planes run 2-8 %
// faster than 8 of 64); a split whose exchange is to overlap with the interior chunks needs at
// least 4.  Where no count fills the slots exactly (tile counts like 260 on general faces, small
// grids) the older search applies: about a thousand blocks, and a last round that is mostly full.
void choose_chunk_depth(kry_solver *h)
{
	if (h->zc_fixed) return;
	const KryGeo &g = h->slabs[0].g;
	const int ty = h->gen ? 8 : h->ty;
	const int tiles = ((g.mx + KRY_TXW - 1) / KRY_TXW) * ((g.my + ((h->gen && !h->gen_y0) ? 3 : 0) + ty - 1) / ty);
	const int chunks = std::max(1, (1024 + tiles - 1) / tiles);
	// planes a stencil kernel evaluates: the faces of the domain are skipped on Dirichlet faces,
	// interface planes and general faces are not (taken equal on all slabs of a split)
	const int nz = (h->gen || h->np[2] > 1) ? g.mz + 1 : g.mz - 1;
	// (small grids: short chunks, the march over a chunk's planes is a latency chain)
	const int zc0 = std::min(64, std::max(g.mz <= 64 ? 2 : 4, (nz + chunks - 1) / chunks));
	int nch = (nz + zc0 - 1) / zc0;
	int cus = 0;
	const int per_cu = kry_fused_blocks_per_cu(h->ty, h->var, h->offd, h->op.exact_rcp != 0, h->gen);
	if (per_cu > 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->slabs[0].device) == hipSuccess &&
	    cus > 0) {
		const long slots = (long)per_cu * cus;
		const int n_min = (h->slabs.size() > 1 || h->np[2] > 1) ? 4 : 1;
		int exact = 0;
		for (int n = n_min; (n == n_min || (long)tiles * n <= 4 * slots) && (nz + n - 1) / n >= 8; ++n)
			if (((long)tiles * n) % slots == 0) { exact = n; break; }
		if (exact) {
			nch = exact;
		} else {
			for (int n = nch; (nz + n - 1) / n >= std::max(8, zc0 / 4); ++n) {
				const double rounds = tiles * (double)n / slots;
				const double last = rounds - std::floor(rounds);
				if (rounds < 1.5) break;                     // the grid does not fill the device anyway
				if (last == 0. || last >= 0.7 || rounds >= 12.) { nch = n; break; }
			}
			// Grids that are no multiple of the tile often do better with fewer, deeper chunks that fill the block
			// slots nearly a whole number of times.  Estimated time of a launch = rounds over the slots x plane steps
			// of a block (depth + the planes of its prologue); a partly filled round counts by its share of the memory
			// bandwidth (a compute unit left with one block draws three quarters of its share), and never less than
			// half a round (its blocks still march through their chunk).  The chunk count with the least estimate
			// among those with chunks of 8 planes or more that fill the device replaces the choice above where the
			// estimate is 5 % better (measured: n = 132: +9 %, 160: +4-7 %, 200: 0-7 %, 260: +5 %; 276: unchanged).
			auto estimate = [&](int n) {
				const double r = tiles * (double)n / slots;
				const double full = std::floor(r), f = r - full;
				const double part = f > 0. ? std::max(0.5, f / (0.75 + 0.5 * std::max(0., f - 0.5))) : 0.;
				return (full + part) * ((nz + n - 1) / n + 3);
			};
			int best = 0;
			for (int n = n_min; (nz + n - 1) / n >= 8 && (long)tiles * n <= 12 * slots; ++n) {
				if (tiles * (double)n / slots < 0.9) continue;
				if (!best || estimate(n) < estimate(best)) best = n;
			}
			if (best && estimate(best) < 0.95 * estimate(nch)) nch = best;
		}
	}
	h->zc = (nz + nch - 1) / nch;      // even chunks
}

int load_common(kry_solver *h, bool var, bool offd, const double Dc[3], const double *phi, const double *rhs,
                const double *lam, const double *Dxx, const double *Dyy, const double *Dzz, const double *Dxy)
{
	if (!phi || !rhs || !lam || (var && (!Dxx || !Dyy || !Dzz)) || (var && offd && !Dxy)) {
		set_err("null field pointer");
		return KRY_ERR_ARG;
	}
	double t0 = now_ms();
	drop_graph(h);
	h->staged_fields = 0;
	h->overlapped_exchanges = 0;
	h->pending_stage_ms = 0;
	h->var = var;
	h->offd = var && offd;
	make_op(h, var, Dc);
	bool dirichlet = true;
	for (int f = 0; f < 6; ++f) dirichlet = dirichlet && (h->d.bc_type[f] == 0);
	// Dirichlet faces: fused kernels that keep the zero boundary implicit.  Other
	// face types: their general-boundary variant; with constant coefficients the
	// operator re-applies the boundary treatment to its INPUT (Krylov.C:516), which
	// that variant evaluates cell by cell while it forms the input (`inbc`).
	// A split in x or y runs the general-boundary variant too: its ghost-cell handling
	// is what neighbours in those directions need.
	const bool can_fuse = !(h->d.flags & KRY_FORCE_GENERIC);
	bool extrapolating = true;     // every face of type 0 or 1
	for (int f = 0; f < 6; ++f) extrapolating = extrapolating && (h->d.bc_type[f] == 0 || h->d.bc_type[f] == 1);
	h->gen = can_fuse && extrapolating && (!dirichlet || h->cart);
	h->inbc = h->gen && !var && !dirichlet;
	// (KRY_GEN_Y0=0: tile rows from -1 everywhere, for comparisons)
	h->gen_y0 = h->gen && !h->cart && !(getenv("KRY_GEN_Y0") && atoi(getenv("KRY_GEN_Y0")) == 0);
	h->fused = can_fuse && ((dirichlet && !h->cart) || h->gen);
	choose_chunk_depth(h);
	CHK(stage_common(h, phi, rhs, lam, Dxx, Dyy, Dzz, Dxy));
	if (h->slabs.size() > 1 || h->comm.active()) {
		// interface ghosts of the inputs, as solve_int does under `parallel`
		// (solveLin_BICGStab.C:259-263) and the operator does for its input (Krylov.C:516)
		Field Slab::*fl[] = {&Slab::phi, &Slab::Dxx, &Slab::Dyy, &Slab::Dzz, &Slab::Dxy};
		for (int k = 0; k < (h->var ? (h->offd ? 5 : 4) : 1); ++k) {
			auto v = ptrs(h, fl[k]);
			CHK(exchange_halo(h, v));
		}
	}
	h->loaded = true;
	h->reset_needed = false;
	for (int k = 8; k < 10; ++k) { h->kernel_ms[k] = 0; h->kernel_calls[k] = 0; }
	h->solve_ms = 0;
	h->enqueue_ms = 0;
	CHK(prologue(h));
	CHK(sync_all(h));
	for (Slab &s : h->slabs) drop_stage(s);
	h->stage_ms = now_ms() - t0;
	return h->status;
}

int finish_solve(kry_solver *h, double *phi)
{
	double t0 = now_ms();
	for (size_t i = 0; i < h->slabs.size(); ++i) CHK(download(h, h->slabs[i], (int)i, h->slabs[i].phi, phi));
	CHK(sync_all(h));
	for (Slab &s : h->slabs) drop_stage(s);
	h->stage_ms += now_ms() - t0;
	return 0;
}

static size_t mail_bytes(int nranks) { return sizeof(double) * KRY_MAIL_SLOT * (1 + 2 * (size_t)nranks); }

long long mail_timeout_ticks(int device, long fallback_ms)
{
	int khz = 0;
	if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
	const char *e = getenv("KRY_MAIL_TIMEOUT_MS");
	const long ms = e ? std::max(1L, atol(e)) : fallback_ms;
	return (long long)khz * ms;
}

// (Re)writes the kernels' description of slab i's mailbox and the link's timeout.
int link_describe(kry_solver *h, int i, long long timeout)
{
	const int ns = (int)h->slabs.size();
	Slab &s = h->slabs[i];
	KryMail m;
	memset(&m, 0, sizeof m);
	m.rank = i;
	m.nranks = ns;
	m.timeout = timeout;
	m.box = s.mail_box;
	std::vector<char *> bases(ns);
	for (int r = 0; r < ns; ++r) {
		m.peer[r] = h->slabs[r].mail_box;
		bases[r] = (char *)h->slabs[r].mail_box;
	}
	HIPCHK(hipSetDevice(s.device));
	if (!s.mail_dev) HIPCHK(hipMalloc((void **)&s.mail_dev, sizeof m));
	HIPCHK(hipMemcpy(s.mail_dev, &m, sizeof m, hipMemcpyHostToDevice));
	if (s.link.set_link(i, (char *)s.mail_box, bases, s.link_cap, timeout, s.S) < 0) {
		set_err("%s", KryComm::last_error());
		return KRY_ERR_HIP;
	}
	return 0;
}

int link_switch(kry_solver *h, bool on)
{
	h->slab_link = on;
	for (Slab &s : h->slabs) {
		s.mail_on = on;
		s.link.link_planes(on);
		const int ls = local_scalars(h);
		const KryMail *mp = on ? s.mail_dev : nullptr;
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(hipMemcpy(&s.S->local_scalars, &ls, sizeof ls, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(&s.S->mail, &mp, sizeof mp, hipMemcpyHostToDevice));
	}
	return 0;
}

// One all-reduce and one ghost-plane exchange of known numbers through a fresh link, with a short
// time-out: 0 when every slab received what it should.
int link_probe(kry_solver *h)
{
	const int ns = (int)h->slabs.size();
	const int cnt = 16, send = 64, rlo = 128, rhi = 192;   // offsets into the slab's partial-sum array
	std::vector<double> host(256);
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		for (int k = 0; k < 256; ++k) host[k] = -1.;
		for (int k = 0; k < cnt; ++k) host[send + k] = 100. * (i + 1) + k;
		HIPCHK(hipMemcpy(s.partial, host.data(), 256 * sizeof(double), hipMemcpyHostToDevice));
	}
	// every slab's kernels are queued before any is waited for: they wait for each other
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		if (s.link.halo(i > 0 ? i - 1 : -1, s.partial + send, s.partial + rlo, i + 1 < ns ? i + 1 : -1, s.partial + send,
		                s.partial + rhi, (size_t)cnt, s.st_comm) < 0) {
			set_err("%s", KryComm::last_error());
			return KRY_ERR_COMM;
		}
	}
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(hipStreamSynchronize(s.st_comm));
		HIPCHK(hipMemcpy(host.data(), s.partial, 256 * sizeof(double), hipMemcpyDeviceToHost));
		for (int k = 0; k < cnt; ++k) {
			const bool ok_lo = i == 0 || host[rlo + k] == 100. * i + k;
			const bool ok_hi = i + 1 == ns || host[rhi + k] == 100. * (i + 2) + k;
			if (!ok_lo || !ok_hi) { set_err("slab %d did not receive its neighbours' planes", i); return KRY_ERR_COMM; }
		}
	}
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(kry_launch_link_probe(make_red(s, 1, PH_NONE), (double)(i + 1), s.st));
	}
	for (int i = 0; i < ns; ++i) {
		Slab &s = h->slabs[i];
		HIPCHK(hipSetDevice(s.device));
		HIPCHK(hipMemcpyAsync(s.Sh, s.S, offsetof(KryState, hist), hipMemcpyDeviceToHost, s.st));
		HIPCHK(hipStreamSynchronize(s.st));
		if (s.Sh->status != 0 || s.Sh->sums[0] != 0.5 * ns * (ns + 1)) {
			set_err("slab %d: the sums of the other slabs did not arrive", i);
			return KRY_ERR_COMM;
		}
	}
	return 0;
}

// Decides whether the slabs of this process exchange through the link, sets it up and tries it.
// Kernels of different slabs then wait for each other, so they must be able to run side by side:
// by default the link is made when every slab has a device of its own and all those devices can
// address each other; KRY_SLAB_LINK=1 also allows two slabs on one device (their four streams must
// then sit on four different hardware queues of the device -- a process hands its queues out in turn,
// so that holds in a fresh process; where two of them share a queue the trial below times out),
// KRY_SLAB_LINK=0 keeps copies and events.  Never fatal: when anything fails the slabs exchange by
// copies as before and kry_stats.slab_link stays 0.
int link_local_slabs(kry_solver *h)
{
	const int ns = (int)h->slabs.size();
	auto off = [&](const char *why) { h->slab_link = false; h->slab_link_note = why; return 0; };
	if (ns < 2) return off("one slab");
	if (h->cart || h->d.nslabs != ns) return off("other processes hold sub-domains too");
	const char *e = getenv("KRY_SLAB_LINK");
	const int want = e ? atoi(e) : -1;
	if (want == 0) return off("KRY_SLAB_LINK=0");
	int most = 0;
	for (int i = 0; i < ns; ++i) {
		int same = 0;
		for (int j = 0; j < ns; ++j) same += h->slabs[j].device == h->slabs[i].device;
		most = std::max(most, same);
	}
	if (most > 2 || (most > 1 && want != 1)) return off("several slabs share a device");
	for (int i = 0; i < ns; ++i)
		for (int j = 0; j < ns; ++j) {
			const int a = h->slabs[i].device, b = h->slabs[j].device;
			if (a == b) continue;
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
				(void)hipGetLastError();
				return off("a device cannot address another one's memory");
			}
			hipSetDevice(a);
			hipDeviceEnablePeerAccess(b, 0);      // (already enabled: an error to forget)
			(void)hipGetLastError();
		}
	auto undo = [&](const char *why) {
		link_switch(h, false);
		for (Slab &s : h->slabs) {
			hipSetDevice(s.device);
			hipStreamSynchronize(s.st);
			hipStreamSynchronize(s.st_comm);
			s.link.destroy();
			if (s.mail_dev) hipFree(s.mail_dev);
			if (s.mail_box) hipFree(s.mail_box);
			s.mail_dev = nullptr;
			s.mail_box = nullptr;
		}
		(void)hipGetLastError();
		return off(why);
	};
	for (Slab &s : h->slabs) {
		if (hipSetDevice(s.device) != hipSuccess) return undo("hipSetDevice failed");
		s.link_cap = (size_t)s.g.ps;
		const size_t bytes = std::max(KryComm::link_bytes(s.link_cap), mail_bytes(ns));
		// uncached: what another device stores is seen by a kernel already running here
		if (hipExtMallocWithFlags((void **)&s.mail_box, bytes, hipDeviceMallocUncached) != hipSuccess ||
		    hipMemset(s.mail_box, 0, bytes) != hipSuccess)
			return undo("no memory for the mailboxes");
	}
	for (int i = 0; i < ns; ++i)
		if (link_describe(h, i, mail_timeout_ticks(h->slabs[i].device, 30000L) / 15) < 0) return undo("set-up failed");   // 2 s for the trial
	if (link_switch(h, true) < 0) return undo("set-up failed");
	if (link_probe(h) < 0) return undo("the trial exchange failed");
	for (int i = 0; i < ns; ++i)
		if (link_describe(h, i, mail_timeout_ticks(h->slabs[i].device, 30000L)) < 0) return undo("set-up failed");
	h->slab_link_note = "on";
	return 0;
}

} // namespace

// A process may hold a second copy of the HIP runtime (a Python package that bundles
// its own): whichever copy touches the GPU first must be this library's, or this
// one's later initialisation fails.  Initialising when the library is loaded makes
// "load this library first" sufficient.
__attribute__((constructor)) static void kry_runtime_init()
{
	int n = 0;
	(void)hipGetDeviceCount(&n);
	(void)hipGetLastError();
}

// ===================================================================== C ABI

extern "C" {

const char *kry_last_error(void) { return g_err.c_str(); }

const char *kry_status_string(int st)
{
	switch (st) {
	case KRY_OK: return "ok";
	case KRY_NAN_ALPHA: return "alpha is nan";
	case KRY_NAN_NORM: return "norm is nan";
	case KRY_MAX_ITER: return "iteration cap reached";
	case KRY_ERR_ARG: return "invalid argument";
	case KRY_ERR_HIP: return "HIP error";
	case KRY_ERR_COMM: return "communication error";
	case KRY_ERR_STATE: return "invalid state";
	}
	return "unknown";
}

kry_handle kry_create(const kry_desc *d, int *status)
{
	auto fail = [&](int st) -> kry_handle {
		if (status) *status = st;
		return nullptr;
	};
	if (!d) { set_err("null descriptor"); return fail(KRY_ERR_ARG); }
	if (d->L < 1 || d->L > KRY_LMAX) { set_err("L=%d outside 1..%d", d->L, KRY_LMAX); return fail(KRY_ERR_ARG); }
	if (d->nslabs < 1 || d->slab_count < 1 || d->slab_count > KRY_MAX_SLABS || d->slab_begin < 0 ||
	    d->slab_begin + d->slab_count > d->nslabs) {
		set_err("bad slab range: nslabs=%d begin=%d count=%d", d->nslabs, d->slab_begin, d->slab_count);
		return fail(KRY_ERR_ARG);
	}
	for (int k = 0; k < 3; ++k)
		if (d->mx[k] < 4 || !(d->delx[k] > 0.)) { set_err("bad grid: mx[%d]=%d delx=%g", k, d->mx[k], d->delx[k]); return fail(KRY_ERR_ARG); }
	const bool has_cart = d->nproc[0] > 0 || d->nproc[1] > 0 || d->nproc[2] > 0;
	if (has_cart) {
		for (int k = 0; k < 3; ++k)
			if (d->nproc[k] < 1 || d->coord[k] < 0 || d->coord[k] >= d->nproc[k]) {
				set_err("bad Cartesian split: nproc[%d]=%d coord=%d", k, d->nproc[k], d->coord[k]);
				return fail(KRY_ERR_ARG);
			}
		if (d->slab_count != 1) { set_err("a Cartesian split takes one sub-domain per process (slab_count=%d)", d->slab_count); return fail(KRY_ERR_ARG); }
	}
	if (d->mx[2] % d->slab_count != 0 || d->mx[2] / d->slab_count < 4) {
		set_err("mz=%d not divisible into %d slabs of >= 4 planes", d->mx[2], d->slab_count);
		return fail(KRY_ERR_ARG);
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
		set_err("no HIP device available");
		return fail(KRY_ERR_HIP);
	}
	for (int i = 0; i < d->slab_count; ++i)
		if (d->devices[i] < 0 || d->devices[i] >= ndev) { set_err("device %d out of range (have %d)", d->devices[i], ndev); return fail(KRY_ERR_ARG); }

	kry_solver *h = new kry_solver();
	h->d = *d;
	h->L = d->L;
	h->mzl = d->mx[2] / d->slab_count;
	if (has_cart) {
		for (int k = 0; k < 3; ++k) { h->np[k] = d->nproc[k]; h->co[k] = d->coord[k]; }
		h->cart = h->np[0] > 1 || h->np[1] > 1;
		h->d.nslabs = h->np[2];
		h->d.slab_begin = h->co[2];
	} else {
		h->np[2] = d->nslabs;
		h->co[2] = d->slab_begin;
	}
	if (const char *e = getenv("KRY_TY")) h->ty = (atoi(e) == 8) ? 8 : 16;
	if (const char *e = getenv("KRY_ZC")) { h->zc = std::max(1, atoi(e)); h->zc_fixed = true; }
	if (const char *e = getenv("KRY_STREAM_BLOCKS")) h->stream_blocks = std::max(64, atoi(e));
	if (const char *e = getenv("KRY_OVERLAP")) h->overlap = atoi(e) != 0;
	if (const char *e = getenv("KRY_PROFILE")) h->profile = atoi(e) != 0;
	if (const char *e = getenv("KRY_GRAPH")) h->use_graph = atoi(e) != 0;
	if (const char *e = getenv("KRY_ALT")) h->alt = atoi(e) != 0;
	if (d->flags & KRY_NO_GRAPH) h->use_graph = false;
	h->slabs.resize(d->slab_count);
	int st = 0;
	auto build = [&]() -> int {
		for (int i = 0; i < d->slab_count; ++i) {
			Slab &s = h->slabs[i];
			s.index = h->d.slab_begin + i;
			s.device = d->devices[i];
			HIPCHK(hipSetDevice(s.device));
			HIPCHK(kry_static_scratch_ready(s.device, nullptr));   // (see kry_device_check)
			HIPCHK(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
			HIPCHK(hipStreamCreateWithFlags(&s.st_comm, hipStreamNonBlocking));
			HIPCHK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
			HIPCHK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
			HIPCHK(hipEventCreateWithFlags(&s.ev_edge, hipEventDisableTiming));
			HIPCHK(hipEventCreateWithFlags(&s.ev_halo, hipEventDisableTiming));
			HIPCHK(hipEventCreateWithFlags(&s.ev_read, hipEventDisableTiming));
			KryGeo &g = s.g;
			g.mx = d->mx[0]; g.my = d->mx[1]; g.mz = h->mzl;
			g.px = ((g.mx + 4 + 15) / 16) * 16;
			g.ps = (long)g.px * (g.my + 3);
			g.n = g.ps * (g.mz + 3);
			g.has_lo = s.index > 0;
			g.has_hi = s.index < h->d.nslabs - 1;
			for (int k = 0; k < 2; ++k) { g.nb[2 * k] = h->co[k] > 0; g.nb[2 * k + 1] = h->co[k] < h->np[k] - 1; }
			g.nb[4] = g.has_lo; g.nb[5] = g.has_hi;
			for (int f = 0; f < 6; ++f) g.bc[f] = d->bc_type[f];
			// guard zones: tile overshoot and halo reads of the stencil kernels (a few rows), and the
			// streaming kernels, whose blocks load whole KRY_SEG-double segments past the end of the
			// last plane they touch before masking the stores
			s.guard = std::max<long>(20L * g.px + 256, KRY_SEG + 64);
			CHK(alloc_field(s, s.phi));
			CHK(alloc_field(s, s.rhs));
			CHK(alloc_field(s, s.lam));
			CHK(alloc_field(s, s.Dxx));
			CHK(alloc_field(s, s.Dyy));
			CHK(alloc_field(s, s.Dzz));
			CHK(alloc_field(s, s.Dxy));
			CHK(alloc_field(s, s.rt));
			for (int k = 0; k <= h->L; ++k) {
				// vectors that feed a fused stencil after an update need a ping-pong partner;
				// so does u1, which the first stencil kernel of a pass both reads (deferred
				// u0 update) and produces
				const int nb = (k < h->L || k == 1) ? 2 : 1;
				for (int b = 0; b < nb; ++b) {
					CHK(alloc_field(s, s.rb[k][b]));
					CHK(alloc_field(s, s.ub[k][b]));
				}
			}
			HIPCHK(hipMalloc((void **)&s.S, sizeof(KryState)));
			HIPCHK(hipMemsetAsync(s.S, 0, sizeof(KryState), s.st));
			HIPCHK(hipHostMalloc((void **)&s.Sh, offsetof(KryState, hist), hipHostMallocDefault));
			memset(s.Sh, 0, offsetof(KryState, hist));
			// partial sums: the generic dot uses one block per grid row
			s.partial_cap = (long)KRY_NSUMS * std::max<long>((long)(g.my + 1) * (g.mz + 1) * ((g.mx + 256) / 256), 65536);
			HIPCHK(hipMalloc((void **)&s.partial, s.partial_cap * sizeof(double)));
			HIPCHK(hipMalloc((void **)&s.counter, 64));
			HIPCHK(hipMemsetAsync(s.counter, 0, 64, s.st));
			HIPCHK(hipMalloc((void **)&s.gather, sizeof(double) * KRY_NSUMS * (KRY_MAX_SLABS + 1)));
			if (h->cart) {
				s.xy_plane = (size_t)(std::max(g.mx, g.my) + 3) * (g.mz + 3);
				HIPCHK(hipMalloc((void **)&s.xy_buf, 8 * s.xy_plane * sizeof(double)));
			}
			HIPCHK(hipStreamSynchronize(s.st));
		}
		// peer access between the devices of neighbouring local slabs
		for (int i = 0; i + 1 < d->slab_count; ++i) {
			int a = h->slabs[i].device, b = h->slabs[i + 1].device;
			if (a == b) continue;
			hipSetDevice(a); hipDeviceEnablePeerAccess(b, 0);
			hipSetDevice(b); hipDeviceEnablePeerAccess(a, 0);
			(void)hipGetLastError();
		}
		return 0;
	};
	st = build();
	if (st >= 0) {
		int khz = 0;
		if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->slabs[0].device) == hipSuccess && khz > 0)
			h->tick_ms = 1. / khz;
	}
	if (st >= 0 && !h->zc_fixed) {
		// planes per fused chunk: long chunks re-read fewer halo planes, but the grid
		// should still hold about a thousand blocks
		const KryGeo &g = h->slabs[0].g;
		const int tiles = ((g.mx + KRY_TXW - 1) / KRY_TXW) * ((g.my + h->ty - 1) / h->ty);
		const int chunks = std::max(1, (1024 + tiles - 1) / tiles);
		// (small grids: short chunks, the march over a chunk's planes is a latency chain)
		// (planes a stencil kernel evaluates: the faces of the domain are skipped on Dirichlet
		// faces, interface planes and general faces are not; equal on all slabs of a split)
		const int nz = (h->gen || h->np[2] > 1) ? g.mz + 1 : g.mz - 1;
		h->zc = std::min(64, std::max(g.mz <= 64 ? 2 : 4, (nz + chunks - 1) / chunks));
		// even chunks: no short last chunk that pays the full three-plane prologue for a plane or two
		const int nch = (nz + h->zc - 1) / h->zc;
		h->zc = (nz + nch - 1) / nch;
	}
	if (st >= 0) st = link_local_slabs(h);
	if (st < 0) {
		kry_destroy(h);
		return fail(st);
	}
	if (status) *status = KRY_OK;
	return h;
}

void kry_destroy(kry_handle h)
{
	if (!h) return;
	drop_graph(h);
	for (Slab &s : h->slabs) {
		hipSetDevice(s.device);
		if (s.st) hipStreamSynchronize(s.st);
		if (s.st_comm) hipStreamSynchronize(s.st_comm);
	}
	h->comm.destroy();
	for (Slab &s : h->slabs) {
		hipSetDevice(s.device);
		s.link.destroy();
		free_fields(s);
		if (s.S) hipFree(s.S);
		if (s.Sh) hipHostFree(s.Sh);
		if (s.partial) hipFree(s.partial);
		if (s.counter) hipFree(s.counter);
		if (s.gather) hipFree(s.gather);
		if (s.stage) hipFree(s.stage);
		if (s.xy_buf) hipFree(s.xy_buf);
		for (void *m : s.mail_mapped) hipIpcCloseMemHandle(m);
		if (s.mail_dev) hipFree(s.mail_dev);
		if (s.mail_box) hipFree(s.mail_box);
		if (s.ev) hipEventDestroy(s.ev);
		if (s.ev_in) hipEventDestroy(s.ev_in);
		if (s.ev_edge) hipEventDestroy(s.ev_edge);
		if (s.ev_halo) hipEventDestroy(s.ev_halo);
		if (s.ev_read) hipEventDestroy(s.ev_read);
		if (s.st_comm) hipStreamDestroy(s.st_comm);
		if (s.st) hipStreamDestroy(s.st);
	}
	for (hipEvent_t e : h->prof.ev) hipEventDestroy(e);
	delete h;
}

int kry_comm_unique_id(void *uid)
{
	int st = KryComm::unique_id(uid);
	if (st < 0) set_err("%s", KryComm::last_error());
	return st;
}

int kry_comm_init(kry_handle h, const void *uid, int rank, int nranks)
{
	if (!h || !uid) { set_err("null argument"); return KRY_ERR_ARG; }
	if (h->d.slab_count != 1 || h->n_ranks() != nranks || h->my_rank() != rank) {
		set_err("kry_comm_init: rank r must own exactly sub-domain r (split %dx%dx%d, this one is number %d; rank=%d/%d)",
		        h->np[0], h->np[1], h->np[2], h->my_rank(), rank, nranks);
		return KRY_ERR_ARG;
	}
	hipSetDevice(h->slabs[0].device);
	int st = h->comm.init(uid, rank, nranks);
	if (st < 0) set_err("%s", KryComm::last_error());
	return st;
}

int kry_comm_selftest(int device)
{
	if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice(%d) failed", device); return KRY_ERR_HIP; }
	int st = KryComm::selftest();
	if (st < 0) set_err("%s", KryComm::last_error());
	return st;
}

int kry_device_check(int device)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err("no HIP device available"); return KRY_ERR_HIP; }
	if (device < 0 || device >= ndev) { set_err("device %d out of range (have %d)", device, ndev); return KRY_ERR_HIP; }
	double *d = nullptr, v = 42., w = 0.;
	bool ok = hipSetDevice(device) == hipSuccess && hipMalloc((void **)&d, sizeof(double)) == hipSuccess &&
	          hipMemcpy(d, &v, sizeof v, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(&w, d, sizeof w, hipMemcpyDeviceToHost) == hipSuccess && w == v;
	if (d) hipFree(d);
	if (!ok) { set_err("device %d does not accept work", device); return KRY_ERR_HIP; }
	// the allocation-free entry point's scratch array: its address on this device, asked here once so
	// that kry_op_dot_static itself never has to (that request may load code and allocate)
	if (kry_static_scratch_ready(device, nullptr) != hipSuccess) {
		(void)hipGetLastError();
		set_err("device %d: the static scratch array has no address", device);
		return KRY_ERR_HIP;
	}
	return KRY_OK;
}

int kry_comm_init_callbacks(kry_handle h, int rank, int nranks, kry_allreduce_cb allreduce, kry_halo_cb halo,
                            void *ctx)
{
	if (!h) { set_err("null argument"); return KRY_ERR_ARG; }
	if (h->d.slab_count != 1 || h->n_ranks() != nranks || h->my_rank() != rank) {
		set_err("kry_comm_init_callbacks: rank r must own exactly sub-domain r (split %dx%dx%d, this one is number %d; rank=%d/%d)",
		        h->np[0], h->np[1], h->np[2], h->my_rank(), rank, nranks);
		return KRY_ERR_ARG;
	}
	int st = h->comm.init_callbacks(rank, nranks, allreduce, halo, ctx);
	if (st < 0) set_err("%s", KryComm::last_error());
	return st;
}

int kry_comm_init_host_callbacks(kry_handle h, int rank, int nranks, kry_allreduce_cb allreduce, kry_halo_cb halo,
                                 void *ctx)
{
	if (!h) { set_err("null argument"); return KRY_ERR_ARG; }
	if (h->d.slab_count != 1 || h->n_ranks() != nranks || h->my_rank() != rank) {
		set_err("kry_comm_init_host_callbacks: rank r must own exactly sub-domain r (split %dx%dx%d, this one is number %d; rank=%d/%d)",
		        h->np[0], h->np[1], h->np[2], h->my_rank(), rank, nranks);
		return KRY_ERR_ARG;
	}
	hipSetDevice(h->slabs[0].device);
	int st = h->comm.init_host_callbacks(rank, nranks, allreduce, halo, ctx);
	if (st < 0) set_err("%s", KryComm::last_error());
	return st;
}

int kry_mail_export(kry_handle h, void *handle)
{
	if (!h || !handle) { set_err("null argument"); return KRY_ERR_ARG; }
	static_assert(sizeof(hipIpcMemHandle_t) == KRY_MAIL_HANDLE_BYTES, "IPC handle size");
	if (h->d.slab_count != 1) { set_err("kry_mail_export: one sub-domain per process"); return KRY_ERR_ARG; }
	if (h->n_ranks() > KRY_MAIL_MAX) { set_err("kry_mail_export: at most %d ranks", KRY_MAIL_MAX); return KRY_ERR_ARG; }
	Slab &s = h->slabs[0];
	HIPCHK(hipSetDevice(s.device));
	if (!s.mail_box) {
		// uncached, so that what a peer GPU stores is seen by a kernel already running here
		// behind the mailbox: the landing slots of the ghost-plane exchange (KryComm::link_bytes)
		s.link_cap = std::max<size_t>((size_t)s.g.ps, 2 * s.xy_plane);
		HIPCHK(hipExtMallocWithFlags((void **)&s.mail_box, KryComm::link_bytes(s.link_cap), hipDeviceMallocUncached));
		HIPCHK(hipMemset(s.mail_box, 0, KryComm::link_bytes(s.link_cap)));
	}
	hipIpcMemHandle_t hd;
	HIPCHK(hipIpcGetMemHandle(&hd, s.mail_box));
	memcpy(handle, &hd, sizeof hd);
	return KRY_OK;
}

int kry_mail_connect(kry_handle h, int rank, int nranks, const void *handles)
{
	if (!h || !handles) { set_err("null argument"); return KRY_ERR_ARG; }
	if (h->d.slab_count != 1 || h->n_ranks() != nranks || h->my_rank() != rank || nranks > KRY_MAIL_MAX) {
		set_err("kry_mail_connect: rank r must own exactly sub-domain r, at most %d ranks (rank=%d/%d, sub-domain %d of %d)",
		        KRY_MAIL_MAX, rank, nranks, h->my_rank(), h->n_ranks());
		return KRY_ERR_ARG;
	}
	Slab &s = h->slabs[0];
	if (!s.mail_box) { set_err("kry_mail_connect: call kry_mail_export first"); return KRY_ERR_STATE; }
	if (s.mail_dev) { set_err("kry_mail_connect: already connected"); return KRY_ERR_STATE; }
	HIPCHK(hipSetDevice(s.device));
	KryMail m;
	memset(&m, 0, sizeof m);
	m.rank = rank;
	m.nranks = nranks;
	int khz = 0;
	if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s.device) != hipSuccess || khz <= 0) khz = 100000;
	const char *e = getenv("KRY_MAIL_TIMEOUT_MS");
	const long ms = e ? std::max(1L, atol(e)) : 30000L;
	m.timeout = (long long)khz * ms;
	m.box = s.mail_box;
	for (int r = 0; r < nranks; ++r) {
		if (r == rank) { m.peer[r] = s.mail_box; continue; }
		hipIpcMemHandle_t hd;
		memcpy(&hd, (const char *)handles + (size_t)r * sizeof hd, sizeof hd);
		void *p = nullptr;
		hipError_t err = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
		if (err != hipSuccess) {
			set_err("kry_mail_connect: cannot map the mailbox of rank %d: %s", r, hipGetErrorString(err));
			for (void *q : s.mail_mapped) hipIpcCloseMemHandle(q);
			s.mail_mapped.clear();
			return KRY_ERR_COMM;
		}
		s.mail_mapped.push_back(p);
		m.peer[r] = (double *)p;
	}
	HIPCHK(hipMalloc((void **)&s.mail_dev, sizeof m));
	HIPCHK(hipMemcpy(s.mail_dev, &m, sizeof m, hipMemcpyHostToDevice));
	std::vector<char *> bases(nranks);
	for (int r = 0; r < nranks; ++r) bases[r] = (char *)m.peer[r];
	if (h->comm.set_link(rank, (char *)s.mail_box, bases, s.link_cap, m.timeout, s.S) < 0) {
		set_err("%s", KryComm::last_error());
		return KRY_ERR_HIP;
	}
	// without a transport everything has to travel through the mapped memory
	return kry_mail_enable(h, h->comm.has_transport() ? KRY_MAIL_SUMS : (KRY_MAIL_SUMS | KRY_MAIL_PLANES));
}

int kry_mail_enable(kry_handle h, int on)
{
	if (!h) { set_err("null argument"); return KRY_ERR_ARG; }
	Slab &s = h->slabs[0];
	if (on && !s.mail_dev) { set_err("kry_mail_enable: not connected"); return KRY_ERR_STATE; }
	HIPCHK(hipSetDevice(s.device));
	HIPCHK(hipStreamSynchronize(s.st));
	drop_graph(h);
	if (!h->comm.has_transport() && on != (KRY_MAIL_SUMS | KRY_MAIL_PLANES)) {
		set_err("kry_mail_enable: no transport is attached that could carry the rest");
		return KRY_ERR_STATE;
	}
	s.mail_on = (on & KRY_MAIL_SUMS) != 0;
	h->comm.link_planes((on & KRY_MAIL_PLANES) != 0);
	const int ls = local_scalars(h);
	const KryMail *mp = s.mail_on ? s.mail_dev : nullptr;
	HIPCHK(hipMemcpy(&s.S->local_scalars, &ls, sizeof ls, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(&s.S->mail, &mp, sizeof mp, hipMemcpyHostToDevice));
	return KRY_OK;
}

int kry_set_profile(kry_handle h, int on)
{
	if (!h) { set_err("null argument"); return KRY_ERR_ARG; }
	CHK(sync_all(h));
	drop_graph(h);
	h->profile = on != 0;
	return KRY_OK;
}

int kry_load_const(kry_handle h, const double *phi, const double *rhs, const double *lambda, double Dxx,
                   double Dyy, double Dzz)
{
	if (!h) return KRY_ERR_ARG;
	const double Dc[3] = {Dxx, Dyy, Dzz};
	return load_common(h, false, false, Dc, phi, rhs, lambda, nullptr, nullptr, nullptr, nullptr);
}

int kry_load_var(kry_handle h, const double *phi, const double *rhs, const double *lambda, const double *Dxx,
                 const double *Dyy, const double *Dzz, const double *Dxy, int use_offdiag)
{
	if (!h) return KRY_ERR_ARG;
	const double Dc[3] = {1., 1., 1.};
	return load_common(h, true, use_offdiag != 0, Dc, phi, rhs, lambda, Dxx, Dyy, Dzz, Dxy);
}

static int solve_tail(kry_handle h, double *phi, int debug);

// After its loop the reference recomputes the true residual (solveLin_BICGStab.C:512-518); the
// value is unused, but the constant-coefficient operator passes its input -- phi -- through
// do_BCs (Krylov.C:516): the caller's phi comes back with its boundary cells re-extrapolated.
// Applied once per finished solve, by whichever call hands phi back first.
static int finish_phi_bcs(kry_handle h)
{
	const bool finished = h->done && (h->status == KRY_OK || h->status == KRY_MAX_ITER);
	if (h->var || !finished || h->it_base <= 0 || h->tail_bc_done) return 0;
	CHK(g_do_bcs(h, &Slab::phi));
	h->tail_bc_done = true;
	return 0;
}

// ---- resident re-solves: the coefficients stay on the device -------------------

static int update_field(kry_handle h, const char *who, Field Slab::*f, const double *src, bool exchange)
{
	if (!h || !src) { set_err("%s: null argument", who); return KRY_ERR_ARG; }
	if (!h->loaded) { set_err("%s: no problem loaded", who); return KRY_ERR_STATE; }
	const double t0 = now_ms();
	if (h->done || h->it_base > 0) { h->staged_fields = 0; h->done = false; h->it_base = 0; }   // first update after a solve
	h->reset_needed = true;
	for (size_t i = 0; i < h->slabs.size(); ++i) CHK(upload(h, h->slabs[i], (int)i, h->slabs[i].*f, src));
	if (exchange && (h->slabs.size() > 1 || h->comm.active())) {
		auto v = ptrs(h, f);
		CHK(exchange_halo(h, v));
	}
	CHK(sync_all(h));
	for (Slab &s : h->slabs) drop_stage(s);
	h->pending_stage_ms += now_ms() - t0;
	return KRY_OK;
}

int kry_update_rhs(kry_handle h, const double *rhs) { return update_field(h, "kry_update_rhs", &Slab::rhs, rhs, false); }
int kry_update_phi(kry_handle h, const double *phi) { return update_field(h, "kry_update_phi", &Slab::phi, phi, true); }
int kry_update_lambda(kry_handle h, const double *lambda)
{
	return update_field(h, "kry_update_lambda", &Slab::lam, lambda, false);
}

int kry_reset(kry_handle h)
{
	if (!h) return KRY_ERR_ARG;
	if (!h->loaded) { set_err("kry_reset: no problem loaded"); return KRY_ERR_STATE; }
	const double t0 = now_ms();
	h->solve_ms = 0;
	h->enqueue_ms = 0;
	for (int k = 8; k < 10; ++k) { h->kernel_ms[k] = 0; h->kernel_calls[k] = 0; }
	CHK(prologue(h));
	CHK(sync_all(h));
	h->reset_needed = false;
	h->stage_ms = h->pending_stage_ms + (now_ms() - t0);
	h->pending_stage_ms = 0;
	return h->status;
}

int kry_resolve(kry_handle h, double *phi, int debug)
{
	if (!h) return KRY_ERR_ARG;
	h->debug = debug;
	int st = kry_reset(h);
	if (st < 0) return st;
	if (!phi) {
		if (st == 0 && !h->done) st = run_iterations(h, 0);
		return st;
	}
	return solve_tail(h, phi, debug);
}

int kry_iterate(kry_handle h, int n_outer)
{
	if (!h) return KRY_ERR_ARG;
	if (n_outer <= 0) { set_err("n_outer must be positive"); return KRY_ERR_ARG; }
	return run_iterations(h, n_outer);
}

int kry_fetch_phi(kry_handle h, double *phi)
{
	if (!h || !phi) return KRY_ERR_ARG;
	if (!h->loaded) { set_err("no problem loaded"); return KRY_ERR_STATE; }
	CHK(finish_phi_bcs(h));      // a finished solve hands back what kry_solve_* would; an unfinished one the iterate
	return finish_solve(h, phi);
}

int kry_fetch_residual(kry_handle h, double *r0)
{
	if (!h || !r0) return KRY_ERR_ARG;
	if (!h->loaded) { set_err("no problem loaded"); return KRY_ERR_STATE; }
	// residuals[0] alternates between its two copies with the pass on the fused path
	const int copy = h->fused ? (h->it_base & 1) : 0;
	double t0 = now_ms();
	for (size_t i = 0; i < h->slabs.size(); ++i) CHK(download(h, h->slabs[i], (int)i, h->slabs[i].rb[0][copy], r0));
	CHK(sync_all(h));
	for (Slab &s : h->slabs) drop_stage(s);
	h->stage_ms += now_ms() - t0;
	return KRY_OK;
}

static int solve_tail(kry_handle h, double *phi, int debug)
{
	h->debug = debug;
	int st = h->status;
	if (st == 0 && !h->done) st = run_iterations(h, 0);
	if (st < 0) return st;
	if (debug > 2) {
		printf(" Final error: %g after %d iterations \n", h->norm_final, h->it_base);
		printf(" Norm: %g %g %g %g %d\n", h->norm_final, h->norm_final / h->norm_rhs, h->norm_rhs,
		       h->d.eps * h->norm_rhs, h->it_base);
	}
	CHK(finish_phi_bcs(h));
	int st2 = finish_solve(h, phi);
	return st2 < 0 ? st2 : st;
}

int kry_solve_const(kry_handle h, double *phi, const double *rhs, const double *lambda, double Dxx, double Dyy,
                    double Dzz, int debug)
{
	if (!h) return KRY_ERR_ARG;
	h->debug = debug;
	int st = kry_load_const(h, phi, rhs, lambda, Dxx, Dyy, Dzz);
	if (st < 0) return st;
	return solve_tail(h, phi, debug);
}

int kry_solve_var(kry_handle h, double *phi, const double *rhs, const double *lambda, const double *Dxx,
                  const double *Dyy, const double *Dzz, const double *Dxy, int use_offdiag, int debug)
{
	if (!h) return KRY_ERR_ARG;
	h->debug = debug;
	int st = kry_load_var(h, phi, rhs, lambda, Dxx, Dyy, Dzz, Dxy, use_offdiag);
	if (st < 0) return st;
	return solve_tail(h, phi, debug);
}

int kry_get_stats(kry_handle h, kry_stats *o)
{
	if (!h || !o) return KRY_ERR_ARG;
	memset(o, 0, sizeof *o);
	o->status = h->status;
	o->iterations = h->it_base;
	o->fused = h->fused ? (h->gen ? 2 : 1) : 0;
	o->norm_rhs = h->norm_rhs;
	o->norm_init = h->norm_init;
	o->norm_final = h->norm_final;
	o->solve_ms = h->solve_ms;
	o->stage_ms = h->stage_ms;
	o->enqueue_ms = h->enqueue_ms;
	// algorithmic doubles moved per grid point and outer pass under the fully fused schedule
	int W;
	if (h->L == 1) W = h->var ? (h->offd ? 28 : 26) : 20;
	else if (h->L == 2) W = h->var ? (h->offd ? 66 : 62) : 50;
	else {
		// 2l stencil kernels, the pointwise sweeps (2 l^2 + l - 3), the Gram-Schmidt kernels (2 l^2), the end of the pass (2l + 7)
		const int per_stencil = h->var ? (h->offd ? 10 : 9) : 6;
		W = 2 * h->L * per_stencil + 4 * h->L * h->L + 3 * h->L + 4;
	}
	const double npts = ((double)h->d.mx[0] * h->np[0] + 1) * ((double)h->d.mx[1] * h->np[1] + 1) *
	                    ((double)h->mzl * h->d.nslabs + 1);
	o->bytes_per_iter = 8.0 * W * npts;
	for (int k = 0; k < 8; ++k) { o->kernel_ms[k] = h->kernel_ms[k]; o->kernel_calls[k] = h->kernel_calls[k]; }
	for (int k = 0; k < 2; ++k) { o->comm_wait_ms[k] = h->kernel_ms[8 + k]; o->comm_wait_calls[k] = h->kernel_calls[8 + k]; }
	if (h->loaded && h->fused && !h->slabs.empty() && h->slabs[0].Sh) {
		// the mapped-memory routes: waiting time counted by the kernels (head of KryState, fetched by poll_state)
		o->comm_wait_ms[0] += h->slabs[0].Sh->wait_sum_ticks * h->tick_ms;
		o->comm_wait_ms[1] += h->slabs[0].Sh->wait_halo_ticks * h->tick_ms;
	}
	o->rccl_ranks = h->comm.rccl_ranks();
	o->staged_fields = h->staged_fields;
	o->overlapped_exchanges = h->overlapped_exchanges;
	o->slab_link = h->slab_link ? 1 : 0;
	o->tile_rows = h->gen ? (h->gen_y0 ? -8 : 8) : h->ty;     // (negative: general-boundary tiles that start at row 0)
	o->chunk_planes = h->zc;
	o->stream_blocks = h->stream_blocks;
	return 0;
}

const char *kry_slab_link_note(kry_handle h) { return h ? h->slab_link_note.c_str() : ""; }

int kry_get_history(kry_handle h, double *norms, int cap)
{
	if (!h || (!norms && cap > 0)) return KRY_ERR_ARG;
	int n = h->it_base;
	if (h->fused) {
		n = std::min(n, KRY_HIST_CAP);
		int m = std::min(n, cap);
		if (m > 0) {
			Slab &s = h->slabs[0];
			hipSetDevice(s.device);
			if (hipMemcpy(norms, s.S->hist, (size_t)m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
				set_err("history copy failed");
				return KRY_ERR_HIP;
			}
		}
		return n;
	}
	n = (int)h->hist.size();
	for (int i = 0; i < std::min(n, cap); ++i) norms[i] = h->hist[i];
	return n;
}

int kry_get_step_times(kry_handle h, double *ms, int cap)
{
	if (!h || (!ms && cap > 0)) return KRY_ERR_ARG;
	const int n = (int)h->step_ms.size();
	for (int i = 0; i < std::min(n, cap); ++i) ms[i] = h->step_ms[i];
	return n;
}

// ------------------------------------------------------ single operators

static int op_prepare(kry_handle h, bool var, bool offd, const double Dc[3])
{
	// the single operators borrow the solver's fields (phi, rhs, lam, r0) and its operator
	// description: whatever problem was loaded is gone
	h->loaded = false;
	drop_graph(h);
	h->gen = false;
	h->inbc = false;
	h->var = var;
	h->offd = var && offd;
	make_op(h, var, Dc);
	return 0;
}

int kry_op_do_bcs(kry_handle h, double *field)
{
	if (!h || !field) return KRY_ERR_ARG;
	h->loaded = false;      // phi is overwritten
	for (size_t i = 0; i < h->slabs.size(); ++i) CHK(upload(h, h->slabs[i], (int)i, h->slabs[i].phi, field));
	CHK(g_do_bcs(h, &Slab::phi));
	for (size_t i = 0; i < h->slabs.size(); ++i) CHK(download(h, h->slabs[i], (int)i, h->slabs[i].phi, field));
	return sync_all(h);
}

static int op_spmv(kry_handle h, double *psi, bool psi_mutable, const double *lambda, double *out, int apply_bcs,
                   int fused)
{
	bool dirichlet = true;
	for (int f = 0; f < 6; ++f) dirichlet = dirichlet && (h->d.bc_type[f] == 0);
	if (fused && !dirichlet) { set_err("fused operator needs Dirichlet faces"); return KRY_ERR_ARG; }
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		CHK(upload(h, s, (int)i, s.phi, psi));
		CHK(upload(h, s, (int)i, s.lam, lambda));
		CHK(zero_field(s, s.rb[0][0]));
	}
	auto in = ptrs(h, &Slab::phi), o = ptrs_r(h, 0);
	if (fused) {
		if (!h->var) CHK(g_do_bcs(h, &Slab::phi));
		for (size_t i = 0; i < h->slabs.size(); ++i) {
			Slab &s = h->slabs[i];
			KryFused a = fused_base(h, s, PH_NONE);
			a.X = s.phi.p;
			a.O = s.rb[0][0].p;
			CHK(launch_fused(h, s, KF_SPMV, a));
		}
		CHK(exchange_halo(h, o));
		(void)apply_bcs;   // Dirichlet output faces stay zero by construction
	} else {
		CHK(g_spmv(h, in, o, apply_bcs != 0));
	}
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		CHK(download(h, s, (int)i, s.rb[0][0], out));
		if (psi_mutable) CHK(download(h, s, (int)i, s.phi, psi));
	}
	return sync_all(h);
}

int kry_op_spmv_const(kry_handle h, double *psi, const double *lambda, double Dxx, double Dyy, double Dzz,
                      double *out, int apply_bcs, int fused)
{
	if (!h || !psi || !lambda || !out) return KRY_ERR_ARG;
	const double Dc[3] = {Dxx, Dyy, Dzz};
	op_prepare(h, false, false, Dc);
	return op_spmv(h, psi, true, lambda, out, apply_bcs, fused);
}

int kry_op_spmv_var(kry_handle h, const double *psi, const double *lambda, const double *Dxx, const double *Dyy,
                    const double *Dzz, const double *Dxy, int use_offdiag, double *out, int apply_bcs, int fused)
{
	if (!h || !psi || !lambda || !out || !Dxx || !Dyy || !Dzz || (use_offdiag && !Dxy)) return KRY_ERR_ARG;
	const double Dc[3] = {1., 1., 1.};
	op_prepare(h, true, use_offdiag != 0, Dc);
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		CHK(upload(h, s, (int)i, s.Dxx, Dxx));
		CHK(upload(h, s, (int)i, s.Dyy, Dyy));
		CHK(upload(h, s, (int)i, s.Dzz, Dzz));
		if (use_offdiag) CHK(upload(h, s, (int)i, s.Dxy, Dxy));
	}
	return op_spmv(h, const_cast<double *>(psi), false, lambda, out, apply_bcs, fused);
}

static int op_dot(kry_handle h, const double *a, const double *b, int norm_mode, double *result)
{
	h->loaded = false;      // phi and rhs are overwritten
	for (size_t i = 0; i < h->slabs.size(); ++i) {
		Slab &s = h->slabs[i];
		CHK(upload(h, s, (int)i, s.phi, a));
		CHK(upload(h, s, (int)i, s.rhs, b));
	}
	auto x = ptrs(h, &Slab::phi), y = ptrs(h, &Slab::rhs);
	double v;
	CHK(g_dot(h, x, y, norm_mode, &v));
	*result = norm_mode ? std::sqrt(v) : v;
	return 0;
}

int kry_op_dot(kry_handle h, const double *a, const double *b, double *result)
{
	if (!h || !a || !b || !result) return KRY_ERR_ARG;
	return op_dot(h, a, b, 0, result);
}

int kry_op_l2norm(kry_handle h, const double *v, double *result)
{
	if (!h || !v || !result) return KRY_ERR_ARG;
	return op_dot(h, v, v, 1, result);
}

int kry_op_dot_static(int device, const double *a, const double *b, const int mx[3], int norm_mode, double *result,
                      void *stream)
{
	if (!a || !b || !mx || !result) { set_err("kry_op_dot_static: null argument"); return KRY_ERR_ARG; }
	for (int k = 0; k < 3; ++k)
		if (mx[k] < 1) { set_err("kry_op_dot_static: mx[%d]=%d, need >= 1", k, mx[k]); return KRY_ERR_ARG; }
	const double n = ((double)mx[0] + 3) * ((double)mx[1] + 3) * ((double)mx[2] + 3);
	if (n > (double)KRY_STATIC_DOT_MAX_DOUBLES) {
		set_err("kry_op_dot_static: a field of %d x %d x %d points (%.0f doubles with ghosts) exceeds the limit of %ld doubles",
		        mx[0] + 1, mx[1] + 1, mx[2] + 1, n, (long)KRY_STATIC_DOT_MAX_DOUBLES);
		return KRY_ERR_ARG;
	}
	int cur = -1;
	HIPCHK(hipGetDevice(&cur));
	if (cur != device) HIPCHK(hipSetDevice(device));
	// the caller's unpadded layout seen through the kernels' index function: row pitch mx+3, grid
	// index i in column i+1, hence a base pointer one double in front of the field
	KryGeo g;
	memset(&g, 0, sizeof g);
	g.mx = mx[0]; g.my = mx[1]; g.mz = mx[2];
	g.px = mx[0] + 3;
	g.ps = (long)g.px * (mx[1] + 3);
	g.n = g.ps * (mx[2] + 3);
	hipError_t e = kry_launch_wdot_static(device, g, a - 1, b - 1, norm_mode, result, (hipStream_t)stream);
	if (cur != device) (void)hipSetDevice(cur);
	if (e != hipSuccess) {
		set_err("kry_op_dot_static: %s", hipGetErrorString(e));
		return KRY_ERR_HIP;
	}
	return KRY_OK;
}

} // extern "C"
