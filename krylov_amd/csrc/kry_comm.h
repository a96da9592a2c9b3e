// Inter-process exchange for z slabs owned by other processes: RCCL over xGMI.
// librccl is opened at run time (kry_comm_init), so a single-process solve
// needs no communication library at all.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>

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
	bool active() const { return comm_ != nullptr || cb_allreduce_ != nullptr; }
	int rank() const { return rank_; }
	int nranks() const { return nranks_; }

	// in-place sum of n doubles over all ranks, queued on `st`
	int allreduce_sum(double *buf, int n, hipStream_t st);
	// send `count` doubles to / receive from the lower and upper neighbour
	// (peer < 0: no neighbour on that side), one grouped call queued on `st`
	int halo(int peer_lo, const double *send_lo, double *recv_lo, int peer_hi, const double *send_hi,
	         double *recv_hi, size_t count, hipStream_t st);

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
};
