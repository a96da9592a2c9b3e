// Check of the one-file-for-all-processes writer (Hdf5Stream with use_MPI_IO,
// Write3DMatrix_withMPI_IO) under a real MPI: the processes form the reference's Cartesian
// split (mpi_manager_3D), every one writes its block of a field whose value encodes the global
// position, and the first process reads the file back through Hdf5iStream.
// usage: mpiexec -n P compat_h5_mpi_selftest FILE px py pz      (P = px*py*pz, 16 points per direction)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "Hdf5File_cbase.H"
#include "grid_manager.H"
#include "mpi_manager.H"

static float code(int i, int j, int k) { return (float)(1 + i + 32 * j + 1024 * k); }

int main(int argc, char **argv)
{
	MPI_Init(&argc, &argv);
	int wrank;
	MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
	if (argc < 5) { if (wrank == 0) fprintf(stderr, "usage: %s FILE px py pz\n", argv[0]); MPI_Finalize(); return 2; }
	NumArray<int> nproc(3), mxg(3), mxl(3), shift(3);
	const int M = 16, rim = 1;
	for (int d = 0; d < 3; ++d) { nproc[d] = atoi(argv[2 + d]); mxg[d] = M; }
	mpi_manager_3D mpi(nproc, mxg);
	for (int d = 0; d < 3; ++d) { mxl[d] = M / nproc[d]; shift[d] = mpi.get_coord(d) * mxl[d]; }
	NumMatrix<float, 3> f(Index::set(-rim, -rim, -rim), Index::set(mxl[0] + rim, mxl[1] + rim, mxl[2] + rim));
	for (int k = -rim; k <= mxl[2] + rim; ++k)
		for (int j = -rim; j <= mxl[1] + rim; ++j)
			for (int i = -rim; i <= mxl[0] + rim; ++i) f(i, j, k) = code(shift[0] + i, shift[1] + j, shift[2] + k);
	NumArray<float> xb(3), dx(3);
	for (int d = 0; d < 3; ++d) { xb[d] = -1.f / M; dx[d] = 1.f / M; }
	int bad = 0;
	{
		Hdf5Stream out(argv[1], 4, wrank, true, MPI_COMM_WORLD);
		if (!out.AddGlobalAttr("time", 0.25)) ++bad;                       // every process may make every call
		if (!out.Write3DMatrix_withMPI_IO("rho", f, mxg, mxl, shift, xb, dx)) ++bad;
		if (!out.Write3DMatrix_withMPI_IO("rho2", f, mxg, mxl, shift, xb, dx, false)) ++bad;
		// one process announces extents its field does not have: every process gets `false`, none waits for another
		{
			int wsize;
			MPI_Comm_size(MPI_COMM_WORLD, &wsize);
			NumArray<int> mxw(3);
			for (int d = 0; d < 3; ++d) mxw[d] = mxl[d] + ((wrank == wsize - 1 && d == 0) ? 1 : 0);
			if (out.Write3DMatrix_withMPI_IO("rho3", f, mxg, mxw, shift, xb, dx)) ++bad;
		}
		out.close();
	}
	MPI_Barrier(MPI_COMM_WORLD);
	if (wrank == 0) {
		Hdf5iStream in(argv[1]);
		NumMatrix<float, 3> g;
		double o[3], d[3];
		if (!in.Read3DMatrix("rho", g, o, d)) ++bad;
		for (int q = 0; q < 3; ++q) {
			if (g.getHigh(q) - g.getLow(q) + 1 != M + 1 + 2 * rim) ++bad;
			if (std::fabs(o[q] + 1. / M) > 1e-7 || std::fabs(d[q] - 1. / M) > 1e-7) ++bad;
		}
		if (!bad)
			for (int k = 0; k <= M + 2 * rim; ++k)
				for (int j = 0; j <= M + 2 * rim; ++j)
					for (int i = 0; i <= M + 2 * rim; ++i)
						if (g(g.getLow(0) + i, g.getLow(1) + j, g.getLow(2) + k) != code(i - rim, j - rim, k - rim)) ++bad;
		double t = 0;
		if (!in.ReadGlobalAttr("time", t) || t != 0.25) ++bad;
		if (in.GetDatasetName(0) != "rho" || in.GetDatasetName(1) != "rho2") ++bad;
		in.close();
		printf(bad ? "selftest FAILED (%d)\n" : "selftest ok\n", bad);
	}
	MPI_Bcast(&bad, 1, MPI_INT, 0, MPI_COMM_WORLD);
	MPI_Finalize();
	return bad ? 1 : 0;
}
