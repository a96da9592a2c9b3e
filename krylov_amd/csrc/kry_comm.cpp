#include "kry_comm.h"
#include "kry_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

namespace {

struct Api {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Api g_api;
std::mutex g_mu;
thread_local std::string g_cerr;

template <class F> bool sym(F &f, const char *name)
{
	f = reinterpret_cast<F>(dlsym(g_api.lib, name));
	return f != nullptr;
}

// The ROCm installation's library is preferred over one a Python package may
// have mapped already under the same soname.
bool load_api()
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (g_api.lib) return true;
	const char *cands[] = {getenv("KRY_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
	for (const char *c : cands) {
		if (!c || !*c) continue;
		g_api.lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
		if (g_api.lib) break;
	}
	if (!g_api.lib) { g_cerr = std::string("cannot open librccl: ") + dlerror(); return false; }
	bool ok = sym(g_api.GetUniqueId, "ncclGetUniqueId") && sym(g_api.CommInitRank, "ncclCommInitRank") &&
	          sym(g_api.CommDestroy, "ncclCommDestroy") && sym(g_api.AllReduce, "ncclAllReduce") &&
	          sym(g_api.Send, "ncclSend") && sym(g_api.Recv, "ncclRecv") && sym(g_api.GroupStart, "ncclGroupStart") &&
	          sym(g_api.GroupEnd, "ncclGroupEnd") && sym(g_api.GetErrorString, "ncclGetErrorString");
	if (!ok) { g_cerr = "librccl lacks a required symbol"; dlclose(g_api.lib); g_api.lib = nullptr; }
	else sym(g_api.CommCount, "ncclCommCount");
	return ok;
}

int fail(const char *what, ncclResult_t r)
{
	g_cerr = std::string(what) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
	return -3;   // KRY_ERR_COMM
}

#define NCHK(call)                                 \
	do {                                           \
		ncclResult_t r_ = (call);                  \
		if (r_ != ncclSuccess) return fail(#call, r_); \
	} while (0)

} // namespace

const char *KryComm::last_error() { return g_cerr.c_str(); }

int KryComm::unique_id(void *uid)
{
	if (!uid) return -1;
	if (!load_api()) return -3;
	ncclUniqueId id;
	NCHK(g_api.GetUniqueId(&id));
	static_assert(sizeof(id) == 128, "unique id size");
	memcpy(uid, &id, sizeof id);
	return 0;
}

int KryComm::init(const void *uid, int rank, int nranks)
{
	if (comm_) return 0;
	if (!load_api()) return -3;
	ncclUniqueId id;
	memcpy(&id, uid, sizeof id);
	ncclComm_t c = nullptr;
	NCHK(g_api.CommInitRank(&c, nranks, id, rank));
	comm_ = c;
	rank_ = rank;
	nranks_ = nranks;
	return 0;
}

int KryComm::rccl_ranks() const
{
	int n = 0;
	if (!comm_ || !g_api.CommCount) return 0;
	if (g_api.CommCount(static_cast<ncclComm_t>(comm_), &n) != ncclSuccess) return 0;
	return n;
}

int KryComm::init_callbacks(int rank, int nranks, allreduce_fn ar, halo_fn hl, void *ctx)
{
	if (!ar || !hl) { g_cerr = "null callback"; return -1; }
	cb_allreduce_ = ar;
	cb_halo_ = hl;
	cb_ctx_ = ctx;
	rank_ = rank;
	nranks_ = nranks;
	return 0;
}

int KryComm::init_host_callbacks(int rank, int nranks, allreduce_fn ar, halo_fn hl, void *ctx)
{
	int st = init_callbacks(rank, nranks, ar, hl, ctx);
	if (st == 0) cb_host_ = true;
	return st;
}

int KryComm::ensure_stage(size_t count)
{
	if (count <= stage_cap_) return 0;
	if (stage_) hipHostFree(stage_);
	stage_ = nullptr;
	stage_cap_ = 0;
	if (hipHostMalloc((void **)&stage_, 4 * count * sizeof(double), hipHostMallocDefault) != hipSuccess) {
		g_cerr = "cannot allocate pinned staging buffers";
		return -2;
	}
	stage_cap_ = count;
	return 0;
}

void KryComm::destroy()
{
	cb_allreduce_ = nullptr;
	cb_halo_ = nullptr;
	cb_host_ = false;
	if (stage_) hipHostFree(stage_);
	stage_ = nullptr;
	stage_cap_ = 0;
	if (comm_ && g_api.CommDestroy) g_api.CommDestroy(static_cast<ncclComm_t>(comm_));
	comm_ = nullptr;
	if (link_ticket_) hipFree(link_ticket_);
	link_ticket_ = nullptr;
	link_on_ = false;
	link_self_ = nullptr;
}

int KryComm::allreduce_sum(double *buf, int n, hipStream_t st)
{
	if (cb_allreduce_ && cb_host_) {
		if (ensure_stage(64) != 0) return -2;
		if (hipMemcpyAsync(stage_, buf, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess) { g_cerr = "staging copy failed"; return -2; }
		if (cb_allreduce_(cb_ctx_, stage_, n) != 0) { g_cerr = "allreduce callback failed"; return -3; }
		if (hipMemcpyAsync(buf, stage_, n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess) { g_cerr = "staging copy failed"; return -2; }
		return 0;
	}
	if (cb_allreduce_) {
		if (hipStreamSynchronize(st) != hipSuccess) { g_cerr = "stream synchronize failed"; return -2; }
		if (cb_allreduce_(cb_ctx_, buf, n) != 0) { g_cerr = "allreduce callback failed"; return -3; }
		return 0;
	}
	if (!comm_) { g_cerr = "no transport attached for the all-reduce"; return -3; }
	NCHK(g_api.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm_), st));
	return 0;
}

// Link area of a rank: [mailbox of the all-reduce | sequence words [3 dirs][2 sides] | landing
// slots [3 dirs][2 sides][2 parities][cap doubles]].  Side 0 receives from the lower
// neighbour, side 1 from the upper one.
size_t KryComm::link_flag_offset()
{
	const size_t mail = sizeof(double) * KRY_MAIL_SLOT * (1 + 2 * (size_t)KRY_MAIL_MAX);
	return (mail + 255) / 256 * 256;
}
static size_t link_slot_offset(size_t cap, int dir, int side, int par)
{
	return KryComm::link_flag_offset() + 256 + (size_t)((dir * 2 + side) * 2 + par) * cap * sizeof(double);
}
size_t KryComm::link_bytes(size_t cap) { return link_slot_offset(cap, 3, 0, 0); }

int KryComm::set_link(int rank, char *self, const std::vector<char *> &peer, size_t cap, long long timeout, KryState *S)
{
	if (!link_ticket_) {
		if (hipMalloc((void **)&link_ticket_, 64) != hipSuccess || hipMemset(link_ticket_, 0, 64) != hipSuccess) {
			g_cerr = "cannot allocate the link's ticket";
			return -2;
		}
	}
	rank_ = rank;
	nranks_ = (int)peer.size();
	link_self_ = self;
	link_peer_ = peer;
	link_cap_ = cap;
	link_timeout_ = timeout;
	link_S_ = S;
	return 0;
}

int KryComm::halo(int peer_lo, const double *send_lo, double *recv_lo, int peer_hi, const double *send_hi,
                  double *recv_hi, size_t count, hipStream_t st, int dir)
{
	if (peer_lo < 0 && peer_hi < 0) return 0;
	if (link_on_) {
		if (count > link_cap_ || dir < 0 || dir > 2) { g_cerr = "halo larger than the link's landing slots"; return -1; }
		const unsigned long long seq = ++link_seq_[dir];
		const int par = (int)(seq & 1ull);
		auto slot = [&](char *base, int side) { return (double *)(base + link_slot_offset(link_cap_, dir, side, par)); };
		auto flag = [&](char *base, int side) {
			return (unsigned long long *)(base + link_flag_offset()) + (dir * 2 + side);
		};
		// towards the lower neighbour this rank is the upper one (side 1 there), and vice versa
		char *plo = peer_lo >= 0 ? link_peer_[peer_lo] : nullptr, *phi = peer_hi >= 0 ? link_peer_[peer_hi] : nullptr;
		if (kry_launch_halo_push(plo ? send_lo : nullptr, plo ? slot(plo, 1) : nullptr, plo ? flag(plo, 1) : nullptr,
		                         phi ? send_hi : nullptr, phi ? slot(phi, 0) : nullptr, phi ? flag(phi, 0) : nullptr, count,
		                         seq, link_ticket_, link_S_, st) != hipSuccess ||
		    kry_launch_halo_pull(plo ? recv_lo : nullptr, plo ? slot(link_self_, 0) : nullptr, flag(link_self_, 0),
		                         phi ? recv_hi : nullptr, phi ? slot(link_self_, 1) : nullptr, flag(link_self_, 1), count,
		                         seq, link_timeout_, link_S_, st) != hipSuccess) {
			g_cerr = "launch of the ghost-plane kernels failed";
			return -2;
		}
		return 0;
	}
	if (cb_halo_ && cb_host_) {
		if (ensure_stage(count) != 0) return -2;
		double *slo = stage_, *shi = stage_ + count, *rlo = stage_ + 2 * count, *rhi = stage_ + 3 * count;
		const size_t bytes = count * sizeof(double);
		bool ok = true;
		if (peer_lo >= 0) ok = ok && hipMemcpyAsync(slo, send_lo, bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
		if (peer_hi >= 0) ok = ok && hipMemcpyAsync(shi, send_hi, bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
		ok = ok && hipStreamSynchronize(st) == hipSuccess;
		if (!ok) { g_cerr = "staging copy failed"; return -2; }
		if (cb_halo_(cb_ctx_, peer_lo, slo, rlo, peer_hi, shi, rhi, (unsigned long)count) != 0) {
			g_cerr = "halo callback failed";
			return -3;
		}
		if (peer_lo >= 0) ok = ok && hipMemcpyAsync(recv_lo, rlo, bytes, hipMemcpyHostToDevice, st) == hipSuccess;
		if (peer_hi >= 0) ok = ok && hipMemcpyAsync(recv_hi, rhi, bytes, hipMemcpyHostToDevice, st) == hipSuccess;
		ok = ok && hipStreamSynchronize(st) == hipSuccess;
		if (!ok) { g_cerr = "staging copy failed"; return -2; }
		return 0;
	}
	if (cb_halo_) {
		if (hipStreamSynchronize(st) != hipSuccess) { g_cerr = "stream synchronize failed"; return -2; }
		if (cb_halo_(cb_ctx_, peer_lo, send_lo, recv_lo, peer_hi, send_hi, recv_hi, (unsigned long)count) != 0) {
			g_cerr = "halo callback failed";
			return -3;
		}
		return 0;
	}
	if (!comm_) { g_cerr = "no transport attached for the ghost planes"; return -3; }
	ncclComm_t c = static_cast<ncclComm_t>(comm_);
	NCHK(g_api.GroupStart());
	if (peer_lo >= 0) {
		NCHK(g_api.Send(send_lo, count, ncclDouble, peer_lo, c, st));
		NCHK(g_api.Recv(recv_lo, count, ncclDouble, peer_lo, c, st));
	}
	if (peer_hi >= 0) {
		NCHK(g_api.Send(send_hi, count, ncclDouble, peer_hi, c, st));
		NCHK(g_api.Recv(recv_hi, count, ncclDouble, peer_hi, c, st));
	}
	NCHK(g_api.GroupEnd());
	return 0;
}

int KryComm::selftest()
{
	ncclUniqueId id;
	if (unique_id(&id) < 0) return -3;
	KryComm c;
	if (c.init(&id, 0, 1) < 0) return -3;
	const int n = 4096;
	double *d = nullptr;
	std::string err;
	auto bad = [&](const char *m) { err = m; };
	if (hipMalloc((void **)&d, 3 * n * sizeof(double)) != hipSuccess) { g_cerr = "hipMalloc failed"; c.destroy(); return -2; }
	double *host = new double[3 * n];
	for (int i = 0; i < n; ++i) { host[i] = 1.0 + i; host[n + i] = 1000.0 + i; host[2 * n + i] = -1.0; }
	hipStream_t st;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	hipMemcpyAsync(d, host, 3 * n * sizeof(double), hipMemcpyHostToDevice, st);
	int rc = c.allreduce_sum(d, 8, st);                       // one rank: values unchanged
	if (rc == 0) rc = c.halo(0, d + n, d + 2 * n, -1, nullptr, nullptr, (size_t)n, st);   // to itself
	if (rc == 0) {
		hipMemcpyAsync(host, d, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st);
		if (hipStreamSynchronize(st) != hipSuccess) bad("stream synchronize failed");
		for (int i = 0; i < 8 && err.empty(); ++i)
			if (host[i] != 1.0 + i) bad("all-reduce changed the data of a one-rank communicator");
		for (int i = 0; i < n && err.empty(); ++i)
			if (host[2 * n + i] != 1000.0 + i) bad("send/receive to self delivered wrong data");
	}
	hipStreamDestroy(st);
	hipFree(d);
	delete[] host;
	c.destroy();
	if (rc < 0) return rc;
	if (!err.empty()) { g_cerr = err; return -3; }
	return 0;
}
