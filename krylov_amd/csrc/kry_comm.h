// Inter-process exchange for z slabs owned by other processes: RCCL over xGMI.
// librccl is opened at run time (kry_comm_init), so a single-process solve
// needs no communication library at all.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <vector>

struct KryState;

class KryComm {
public:
	// 128-byte id created on one rank and handed to every rank's init()
	static int unique_id(void *uid);
	static const char *last_error();
	// one-rank communicator on the current device: all-reduce + self send/recv, verified
	static int selftest();

	int init(const void *uid, int rank, int nranks);
	// caller-provided transport (device pointers, blocking); see include/krylov_amd.h
	typedef int (*allreduce_fn)(void *, double *, int);
	typedef int (*halo_fn)(void *, int, const double *, double *, int, const double *, double *, unsigned long);
	int init_callbacks(int rank, int nranks, allreduce_fn ar, halo_fn hl, void *ctx);
	// the same with HOST pointers: the data is staged through pinned buffers here
	int init_host_callbacks(int rank, int nranks, allreduce_fn ar, halo_fn hl, void *ctx);
	void destroy();
	bool active() const { return comm_ != nullptr || cb_allreduce_ != nullptr || link_self_ != nullptr; }
	bool has_transport() const { return comm_ != nullptr || cb_allreduce_ != nullptr; }
	// ranks the RCCL communicator reports (ncclCommCount); 0 without one
	int rccl_ranks() const;
	int rank() const { return rank_; }
	int nranks() const { return nranks_; }

	// in-place sum of n doubles over all ranks, queued on `st`
	int allreduce_sum(double *buf, int n, hipStream_t st);
	// send `count` doubles to / receive from the lower and upper neighbour
	// (peer < 0: no neighbour on that side), one grouped call queued on `st`
	// `dir`: 0, 1, 2 = the exchange runs along x, y, z (landing slots are kept per direction)
	int halo(int peer_lo, const double *send_lo, double *recv_lo, int peer_hi, const double *send_hi,
	         double *recv_hi, size_t count, hipStream_t st, int dir = 2);

	// Ghost planes through device memory the ranks have mapped from each other instead of
	// the transport: `self` / `peer[r]` are the ranks' link areas (this rank's own and as
	// mapped here), laid out as link_bytes() describes; `cap` doubles per landing slot.
	static size_t link_flag_offset();
	static size_t link_bytes(size_t cap);
	int set_link(int rank, char *self, const std::vector<char *> &peer, size_t cap, long long timeout, KryState *S);
	void link_planes(bool on) { link_on_ = on && link_self_ != nullptr; }
	bool link_planes() const { return link_on_; }
	// how halo() moves the planes: through library calls that involve the communicator (RCCL), or
	// through caller-supplied callbacks, which return only when the planes have arrived
	bool halo_by_rccl() const { return !link_on_ && !cb_halo_ && comm_ != nullptr; }
	bool halo_blocks_host() const { return !link_on_ && cb_halo_ != nullptr; }

private:
	void *comm_ = nullptr;
	allreduce_fn cb_allreduce_ = nullptr;
	halo_fn cb_halo_ = nullptr;
	void *cb_ctx_ = nullptr;
	bool cb_host_ = false;          // callbacks take host pointers
	double *stage_ = nullptr;       // pinned: [send_lo | send_hi | recv_lo | recv_hi], each stage_cap_ doubles
	size_t stage_cap_ = 0;
	int ensure_stage(size_t count);
	int rank_ = 0, nranks_ = 1;
	// mapped-memory link
	bool link_on_ = false;
	char *link_self_ = nullptr;
	std::vector<char *> link_peer_;
	size_t link_cap_ = 0;
	unsigned long long link_seq_[3] = {0, 0, 0};
	unsigned *link_ticket_ = nullptr;
	long long link_timeout_ = 0;
	KryState *link_S_ = nullptr;
};
