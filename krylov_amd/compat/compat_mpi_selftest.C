// Host-only check of the `-D parallel` classes: process topology, local grids and
// the plane exchange of BoundaryHandler3D::do_BCs.  Run under mpiexec with any
// number of processes P that divides the grid; the split is {1, 1, P} unless three
// factors are given.  Prints "selftest ok" on the first process, exits non-zero on
// a mismatch.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "BoundaryHandler.H"
#include "grid_manager.H"
#include "mpi_manager.H"

static double code(double x, double y, double z) { return 1. + x + 10. * y + 100. * z; }

int main(int argc, char **argv)
{
	MPI_Init(&argc, &argv);
	int ntasks, wrank;
	MPI_Comm_size(MPI_COMM_WORLD, &ntasks);
	MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
	NumArray<int> nproc(3), mxg(3);
	nproc[0] = 1; nproc[1] = 1; nproc[2] = ntasks;
	if (argc >= 4) for (int d = 0; d < 3; ++d) nproc[d] = atoi(argv[1 + d]);
	const int M = 16;
	mxg[0] = mxg[1] = mxg[2] = M;
	mpi_manager_3D mpi(nproc, mxg);
	grid_manager global(0., 0., 0., 1., 1., 1., M + 1, M + 1, M + 1, 1);
	for (int b = 0; b < 6; ++b) global.set_bcType(b, 0);
	grid_manager local = mpi.make_LocalGrid(global);
	int bad = 0;

	// geometry: same spacing, shifted origin, shared interface planes
	for (int d = 0; d < 3; ++d) {
		if (local.get_mx(d) != M / nproc[d]) ++bad;
		if (std::fabs(local.get_delx(d) - global.get_delx(d)) > 1e-15) ++bad;
		const double want = global.get_Pos(d, mpi.get_coord(d) * local.get_mx(d));
		if (std::fabs(local.get_Pos(d, 0) - want) > 1e-14) ++bad;
		grid_1D g1(0, 1, M + 1, 2, 0, true);
		grid_1D l1 = mpi.make_LocalGrid1D(g1, d);
		if (l1.get_mx() != local.get_mx(d) || std::fabs(l1.get_del() - g1.get_del()) > 1e-15) ++bad;
		if (local.get_bcType(2 * d) != (mpi.is_OuterBoundary(2 * d) ? 0 : -1)) ++bad;
		if (local.get_bcType(2 * d + 1) != (mpi.is_OuterBoundary(2 * d + 1) ? 0 : -1)) ++bad;
	}
	// rank layout of a Cartesian communicator without reordering (row major, z fastest)
	const int want_rank = (mpi.get_coord(0) * nproc[1] + mpi.get_coord(1)) * nproc[2] + mpi.get_coord(2);
	if (mpi.get_rank() != want_rank) ++bad;
	if (mpi.get_OtherRankAbs(mpi.get_coord(0), mpi.get_coord(1), mpi.get_coord(2)) != mpi.get_rank()) ++bad;

	// exchange: a field that encodes the global position; after do_BCs every ghost cell
	// towards a neighbour must hold the neighbour's value, i.e. the code of its own position
	const int mx = local.get_mx(0), my = local.get_mx(1), mz = local.get_mx(2);
	NumMatrix<double, 3> f;
	f.resize(Index::set(-1, -1, -1), Index::set(mx + 1, my + 1, mz + 1));
	f.clear();
	for (int k = 0; k <= mz; ++k)
		for (int j = 0; j <= my; ++j)
			for (int i = 0; i <= mx; ++i) f(i, j, k) = code(local.get_Pos(0, i), local.get_Pos(1, j), local.get_Pos(2, k));
	BoundaryHandler3D bounds(mpi);
	bounds.set_bcType(1);            // outer faces: linear extrapolation reproduces the (linear) code exactly
	bounds.do_BCs(f, 1);
	for (int k = -1; k <= mz + 1; ++k)
		for (int j = -1; j <= my + 1; ++j)
			for (int i = -1; i <= mx + 1; ++i) {
				// cells of edges and corners towards two neighbours travel in two hops and
				// hold valid data as well; ghost edges that touch an OUTER face are not
				// defined (the y and z face fills skip the x ghosts, BoundaryHandler.C:479-480, 529-530)
				int outside = 0, ghost = 0;
				if (i < 0 || i > mx) ++ghost;
				if (j < 0 || j > my) ++ghost;
				if (k < 0 || k > mz) ++ghost;
				if ((i < 0 && mpi.is_OuterBoundary(0)) || (i > mx && mpi.is_OuterBoundary(1))) ++outside;
				if ((j < 0 && mpi.is_OuterBoundary(2)) || (j > my && mpi.is_OuterBoundary(3))) ++outside;
				if ((k < 0 && mpi.is_OuterBoundary(4)) || (k > mz && mpi.is_OuterBoundary(5))) ++outside;
				if (ghost > 1 && outside > 0) continue;
				const double want = code(local.get_Pos(0, i), local.get_Pos(1, j), local.get_Pos(2, k));
				if (std::fabs(f(i, j, k) - want) > 1e-12) {
					if (bad < 5) fprintf(stderr, "rank %d: (%d,%d,%d) = %.15g, want %.15g\n", wrank, i, j, k, f(i, j, k), want);
					++bad;
				}
			}
	int total = 0;
	MPI_Allreduce(&bad, &total, 1, MPI_INT, MPI_SUM, mpi.comm3d);
	if (mpi.get_rank() == 0) printf(total == 0 ? "selftest ok\n" : "selftest FAILED (%d)\n", total);
	MPI_Finalize();
	return total == 0 ? 0 : 1;
}
