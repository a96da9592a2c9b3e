// HBM streaming ceilings on this device for the access shapes the solver uses.
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline double2 ld(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ inline void st(double* p, double2 v) { *reinterpret_cast<double2*>(p) = v; }
__device__ inline double2 ldnt(const double* p) { double2 v; v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1); return v; }
__device__ inline void stnt(double* p, double2 v) { __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1); }

template <int NR, int NW, bool NT_, int UN>
__global__ __launch_bounds__(256) void k_stream(double* const* rp, double* const* wp, long n2, double c)
{
	const double* r[NR]; double* w[NW + 1];
#pragma unroll
	for (int i = 0; i < NR; ++i) r[i] = rp[i];
#pragma unroll
	for (int i = 0; i < NW; ++i) w[i] = wp[i];
	const long stride = (long)gridDim.x * 256 * UN;
	for (long q0 = (long)blockIdx.x * 256 * UN + threadIdx.x; q0 < n2; q0 += stride) {
		double2 acc[UN];
#pragma unroll
		for (int u = 0; u < UN; ++u) {
			const long q = q0 + u * 256;
			acc[u] = {0., 0.};
			if (q < n2) {
#pragma unroll
				for (int i = 0; i < NR; ++i) {
					double2 v = NT_ ? ldnt(r[i] + 2 * q) : ld(r[i] + 2 * q);
					acc[u].x += v.x * c; acc[u].y += v.y * c;
				}
			}
		}
#pragma unroll
		for (int u = 0; u < UN; ++u) {
			const long q = q0 + u * 256;
			if (q < n2) {
#pragma unroll
				for (int i = 0; i < NW; ++i) { if (NT_) stnt(w[i] + 2 * q, acc[u]); else st(w[i] + 2 * q, acc[u]); }
				if (NW == 0 && acc[u].x == 123.456) ((double*)r[0])[0] = 1.;
			}
		}
	}
}

template <int NR, int NW, bool NT_, int UN>
double run(std::vector<double*>& bufs, long n, int nblk, const char* name, bool inplace = false)
{
	double **dr, **dw;
	CHECK(hipMalloc(&dr, 16 * sizeof(double*))); CHECK(hipMalloc(&dw, 16 * sizeof(double*)));
	std::vector<double*> r(bufs.begin(), bufs.begin() + NR), w(bufs.begin() + NR, bufs.begin() + NR + (NW ? NW : 1));
	if (inplace) w.assign(bufs.begin(), bufs.begin() + (NW ? NW : 1));
	CHECK(hipMemcpy(dr, r.data(), NR * sizeof(double*), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(dw, w.data(), (NW ? NW : 1) * sizeof(double*), hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const long n2 = n / 2;
	for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_stream<NR, NW, NT_, UN>), dim3(nblk), dim3(256), 0, 0, dr, dw, n2, 1.0001);
	CHECK(hipEventRecord(e0, 0));
	const int reps = 10;
	for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<NR, NW, NT_, UN>), dim3(nblk), dim3(256), 0, 0, dr, dw, n2, 1.0001);
	CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	double gbs = (double)(NR + NW) * n * 8 * reps / (ms * 1e-3) / 1e9;
	printf("%-28s R%d W%d nt=%d un=%d blocks=%6d : %8.1f GB/s\n", name, NR, NW, (int)NT_, UN, nblk, gbs);
	hipFree(dr); hipFree(dw);
	return gbs;
}

int main(int argc, char** argv)
{
	long n = (argc > 1) ? atol(argv[1]) : 136000000L;   // ~512^3 doubles
	n &= ~1023L;
	std::vector<double*> bufs(12);
	// MEMBENCH_SKEW=<bytes>: carve the arrays out of one allocation, array i starting at a
	// multiple of 2 MiB plus i*<bytes> (0: all arrays equally aligned); unset: one hipMalloc each
	const char* sk = getenv("MEMBENCH_SKEW");
	if (sk) {
		const size_t skew = (size_t)atol(sk), two_mb = 2u << 20;
		const size_t slot = ((n * sizeof(double) + 12 * skew + two_mb) / two_mb + 1) * two_mb;
		char* base = nullptr;
		CHECK(hipMalloc(&base, slot * 12 + two_mb));
		CHECK(hipMemset(base, 0, slot * 12 + two_mb));
		base = (char*)(((size_t)base + two_mb - 1) / two_mb * two_mb);
		for (size_t i = 0; i < bufs.size(); ++i) bufs[i] = (double*)(base + i * slot + i * skew);
		printf("arrays carved from one allocation, skew %zu bytes\n", skew);
	} else {
		for (auto& b : bufs) { CHECK(hipMalloc(&b, n * sizeof(double) + 4096)); CHECK(hipMemset(b, 0, n * sizeof(double))); }
		for (auto& b : bufs) printf("%p ", (void*)b);
		printf("\n");
	}
	for (int nblk : {16384, 65536}) {
		run<1, 1, false, 4>(bufs, n, nblk, "copy");
		run<1, 1, true, 4>(bufs, n, nblk, "copy nontemporal");
		run<1, 0, false, 4>(bufs, n, nblk, "read");
		run<4, 0, false, 4>(bufs, n, nblk, "read4");
		run<3, 1, false, 4>(bufs, n, nblk, "3R1W (mgs2 shape)");
		run<3, 1, true, 4>(bufs, n, nblk, "3R1W nontemporal");
		run<8, 3, false, 4>(bufs, n, nblk, "8R3W (final shape)");
		run<8, 3, true, 4>(bufs, n, nblk, "8R3W nontemporal");
		run<8, 3, false, 4>(bufs, n, nblk, "8R3W in place", true);
		run<8, 3, true, 4>(bufs, n, nblk, "8R3W in place nontemporal", true);
		run<8, 3, true, 2>(bufs, n, nblk, "8R3W in place nt un2", true);
	}
	return 0;
}
