// What the memory system gives to the ACCESS PATTERN of the fused stencil kernels, without the stencil:
// 256-thread blocks march a tile (TW doubles x TY rows) over a chunk of planes; per plane every thread
// loads its double2 elements of NR arrays (the loads of plane z+PF are issued before plane z is used),
// adds them up and stores the sum to NW arrays; one barrier per plane as in the real kernel (BAR).
// Field layout as in the solver: rows padded to a multiple of 16 doubles, first point 128-byte aligned.
// Build: make -C tools tilebench      Run: tools/tilebench [N] [zc]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
__device__ inline d2 ldp(const double* p) { return *reinterpret_cast<const d2*>(p); }
__device__ inline d2 ldnt(const double* p) { return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p)); }
__device__ inline void stnt(double* p, d2 v) { __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p)); }

struct Args {
	const double* r[8];
	double* w[3];
	long px, ps;
	int ntx, nty, nchunk, zc, nz, nblk, pad_lds;
};

template <int NR, int NW, int TW, int TY, int PF, bool BAR>
__global__ __launch_bounds__(256) void k_tile(const Args a)
{
	constexpr int LPR = TW / 2;              // lanes per tile row
	constexpr int RPP = 256 / LPR;           // rows per sweep of the block
	constexpr int VD = TY / RPP;             // sweeps = double2 elements per thread, plane and array
	extern __shared__ double lds[];
	// XCD-aware block -> tile map (as k_fused)
	int bid;
	{
		const int nb = a.nblk, b = blockIdx.x;
		const int q = nb >> 3, r = nb & 7, x = b & 7;
		bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
	}
	const int tx = bid % a.ntx, ty = (bid / a.ntx) % a.nty, ch = bid / (a.ntx * a.nty);
	const int tid = threadIdx.x;
	const int row0 = tid / LPR, lane = tid % LPR;
	long off[VD];
#pragma unroll
	for (int e = 0; e < VD; ++e) off[e] = (long)(ty * TY + e * RPP + row0 + 1) * a.px + 2 + tx * TW + 2 * lane;
	const int z0 = 1 + ch * a.zc, z1 = min(z0 + a.zc, a.nz);
	d2 buf[PF + 1][NR][VD];
	auto issue = [&](int z, int slot) {
		const long pb = (long)z * a.ps;
#pragma unroll
		for (int i = 0; i < NR; ++i)
#pragma unroll
			for (int e = 0; e < VD; ++e) buf[slot][i][e] = (i < 2) ? ldp(a.r[i] + pb + off[e]) : ldnt(a.r[i] + pb + off[e]);
	};
#pragma unroll
	for (int k = 0; k < PF; ++k) issue(z0 + k, k);
	// plane z sits in register slot (z - z0) % (PF + 1); the loop is unrolled over one period of the ring
	for (int zb = z0; zb < z1; zb += PF + 1) {
#pragma unroll
		for (int s = 0; s <= PF; ++s) {
			const int z = zb + s;
			if (z < z1) {
				if (z + PF < z1) issue(z + PF, (s + PF) % (PF + 1));
				const long pb = (long)z * a.ps;
#pragma unroll
				for (int e = 0; e < VD; ++e) {
					d2 acc = buf[s][0][e];
#pragma unroll
					for (int i = 1; i < NR; ++i) acc += buf[s][i][e] * 1.0001;
					if (BAR) lds[(z & 1) * 1024 + tid * 2 + (e & 1)] = acc.x;
#pragma unroll
					for (int i = 0; i < NW; ++i) stnt(a.w[i] + pb + off[e], acc);
				}
				if (BAR) __syncthreads();
			}
		}
	}
}

template <int NR, int NW, int TW, int TY, int PF, bool BAR>
void run(std::vector<double*>& bufs, int n, int zc, int lds_bytes, const char* name)
{
	Args a;
	a.px = ((n + 4 + 15) / 16) * 16;
	a.ps = a.px * (n + 3);
	a.nz = n;   // planes 1 .. n-1
	for (int i = 0; i < 8; ++i) a.r[i] = bufs[i] + 14;
	for (int i = 0; i < 3; ++i) a.w[i] = bufs[8 + i] + 14;
	a.ntx = (n + TW - 1) / TW;
	a.nty = (n + TY - 1) / TY;
	a.zc = zc;
	a.nchunk = (n - 1 + zc - 1) / zc;
	a.nblk = a.ntx * a.nty * a.nchunk;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	auto kern = k_tile<NR, NW, TW, TY, PF, BAR>;
	for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(256), lds_bytes, 0, a);
	CHECK(hipEventRecord(e0, 0));
	const int reps = 10;
	for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(256), lds_bytes, 0, a);
	CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
	CHECK(hipGetLastError());
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	int occ = 0;
	hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds_bytes);
	const double pts = (double)(n - 1) * n * n;   // points the tiles cover (rows 0..n-1, planes 1..n-1)
	printf("%-22s R%d W%d tile %3dx%-2d pf=%d bar=%d lds=%6d blocks=%5d occ=%d : %7.3f ms %8.1f GB/s\n", name, NR, NW, TW, TY, PF,
	       (int)BAR, lds_bytes, a.nblk, occ, ms / reps, (NR + NW) * pts * 8 * reps / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv)
{
	const int n = (argc > 1) ? atoi(argv[1]) : 512;
	const int zc = (argc > 2) ? atoi(argv[2]) : 64;
	const long px = ((n + 4 + 15) / 16) * 16, total = px * (n + 3) * (n + 3) + 8192;
	std::vector<double*> bufs(11);
	for (auto& b : bufs) { CHECK(hipMalloc(&b, total * sizeof(double))); CHECK(hipMemset(b, 0, total * sizeof(double))); }
	const int two = 57344 - 16384, pad16k = 16384;   // dynamic LDS that holds a compute unit to 2 blocks
	printf("n = %d, planes per chunk = %d\n", n, zc);
	const int one = 90112 - 16384;
	// the shapes of the l = 1 kernels: var A0 (8R 2W), const A0 (5R 2W), const B (3R 2W)
	run<8, 2, 128, 8, 1, true>(bufs, n, zc, two, "var A0 shape");
	run<8, 2, 128, 8, 2, true>(bufs, n, zc, two, "  prefetch 2");
	run<8, 2, 256, 8, 1, true>(bufs, n, zc, two, "  tile 256x8");
	run<8, 2, 128, 16, 1, true>(bufs, n, zc, two, "  tile 128x16");
	run<8, 2, 512, 4, 1, true>(bufs, n, zc, two, "  tile 512x4");
	run<8, 2, 256, 16, 1, true>(bufs, n, zc, two, "  tile 256x16 (spills)");
	run<5, 2, 128, 8, 1, true>(bufs, n, zc, two, "const A0 shape");
	run<5, 2, 256, 8, 1, true>(bufs, n, zc, two, "  tile 256x8");
	run<5, 2, 256, 8, 1, true>(bufs, n, zc, one, "  tile 256x8, 1 per CU");
	run<5, 2, 128, 16, 1, true>(bufs, n, zc, two, "  tile 128x16");
	run<5, 2, 512, 4, 1, true>(bufs, n, zc, two, "  tile 512x4");
	run<5, 2, 256, 16, 1, true>(bufs, n, zc, two, "  tile 256x16");
	run<5, 2, 256, 8, 2, true>(bufs, n, zc, two, "  tile 256x8 pf2");
	run<3, 2, 128, 8, 1, true>(bufs, n, zc, two, "const B shape");
	run<3, 2, 256, 8, 1, true>(bufs, n, zc, two, "  tile 256x8");
	run<3, 2, 256, 8, 1, true>(bufs, n, zc, one, "  tile 256x8, 1 per CU");
	run<3, 2, 128, 16, 1, true>(bufs, n, zc, two, "  tile 128x16");
	run<3, 2, 256, 16, 1, true>(bufs, n, zc, two, "  tile 256x16");
	run<3, 2, 256, 16, 1, true>(bufs, n, zc, one, "  tile 256x16, 1 per CU");
	run<3, 2, 512, 8, 1, true>(bufs, n, zc, two, "  tile 512x8");
	return 0;
}
