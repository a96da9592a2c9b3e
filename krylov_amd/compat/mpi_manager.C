#include "mpi_manager.H"

#include <cassert>
#include <cmath>
#include <cstdlib>
#include <iostream>

mpi_manager_3D::mpi_manager_3D()
    : comm3d(MPI_COMM_NULL), comm_plane_xy(MPI_COMM_NULL), comm_plane_xz(MPI_COMM_NULL), comm_plane_yz(MPI_COMM_NULL),
      ntasks(1), left(MPI_PROC_NULL), right(MPI_PROC_NULL), front(MPI_PROC_NULL), back(MPI_PROC_NULL),
      top(MPI_PROC_NULL), bottom(MPI_PROC_NULL), rank(0), DIM(3)
{
	for (int d = 0; d < 3; ++d) { nproc[d] = 1; coords[d] = 0; }
}

mpi_manager_3D::mpi_manager_3D(NumArray<int> &nproc_, NumArray<int> &mx) : mpi_manager_3D() { setup(nproc_, mx); }

void mpi_manager_3D::setup(NumArray<int> &nproc_, NumArray<int> &mx)
{
	for (int d = 0; d < DIM; ++d) nproc[d] = nproc_[d];
	MPI_Comm_rank(MPI_COMM_WORLD, &rank);
	MPI_Comm_size(MPI_COMM_WORLD, &ntasks);

	// the checks of BICGStab3D/mpi_manager.C:71-109
	if (ntasks != nproc[0] * nproc[1] * nproc[2]) {
		std::cerr << " Wrong number of processes " << std::endl;
		std::cout << ntasks << " " << nproc[0] * nproc[1] * nproc[2] << std::endl;
		Finalise();
	}
	if (rank == 0) std::cout << " Number of tasks: " << ntasks << std::endl;
	for (int d = 0; d < DIM; ++d) {
		if (mx[d] < nproc[d] && nproc[d] > 1) {
			if (rank == 0) {
				std::cerr << " Wrong grid topology for dimension " << d << std::endl;
				std::cerr << "  mx[" << d << "]:" << mx[d] << std::endl;
				std::cerr << " nproc[" << d << "]:" << nproc[d] << std::endl;
			}
			Finalise();
		}
	}
	for (int d = 0; d < DIM; ++d) {
		int v = mx[d];
		if (v < 1 || (v & (v - 1)) != 0) {
			if (rank == 0) {
				std::cerr << " Error: grid must be of the form mx = 2^n " << std::endl;
				std::cerr << " Exiting " << std::endl;
			}
			Finalise();
		}
	}

	int periods[3] = {0, 0, 0};
	MPI_Cart_create(MPI_COMM_WORLD, DIM, nproc, periods, 0, &comm3d);
	MPI_Comm_rank(comm3d, &rank);
	MPI_Cart_coords(comm3d, rank, DIM, coords);
	MPI_Cart_shift(comm3d, 0, 1, &left, &right);
	MPI_Cart_shift(comm3d, 1, 1, &front, &back);
	MPI_Cart_shift(comm3d, 2, 1, &bottom, &top);
	if (rank == 0) std::cout << " Cart topology:  MPI_CART " << nproc[0] << "x" << nproc[1] << "x" << nproc[2] << std::endl;

	int keep_xy[3] = {1, 1, 0}, keep_xz[3] = {1, 0, 1}, keep_yz[3] = {0, 1, 1};
	MPI_Cart_sub(comm3d, keep_xy, &comm_plane_xy);
	MPI_Cart_sub(comm3d, keep_xz, &comm_plane_xz);
	MPI_Cart_sub(comm3d, keep_yz, &comm_plane_yz);
}

grid_1D mpi_manager_3D::make_LocalGrid1D(grid_1D &GlobalGrid1D, int dir)
{
	// BICGStab3D/mpi_manager.C:308-340.  (There the local origin is advanced twice;
	// only mx and the spacing of these 1D grids are ever used, and those agree.)
	const int mx = GlobalGrid1D.get_mx() / nproc[dir];
	const double Len = GlobalGrid1D.get_Len() / nproc[dir];
	const double xb = GlobalGrid1D.get_xb() + Len * coords[dir];
	return grid_1D(xb, xb + Len, mx + 1, 2, 0, true);
}

grid_manager mpi_manager_3D::make_LocalGrid(grid_manager &GlobalGrid)
{
	// BICGStab3D/mpi_manager.C:342-408
	int mx[3];
	double xb[3], xe[3];
	for (int d = 0; d < DIM; ++d) {
		mx[d] = GlobalGrid.get_mx(d) / nproc[d];
		const double Len = GlobalGrid.get_Len(d) / nproc[d];
		xb[d] = GlobalGrid.get_xb(d) + Len * coords[d];
		xe[d] = xb[d] + Len;
	}
	grid_manager LocalGrid(xb[0], xb[1], xb[2], xe[0], xe[1], xe[2], mx[0] + 1, mx[1] + 1, mx[2] + 1,
	                       GlobalGrid.get_rim());
	for (int b = 0; b < 6; ++b) LocalGrid.set_bcType(b, is_OuterBoundary(b) ? GlobalGrid.get_bcType(b) : -1);
	return LocalGrid;
}

int mpi_manager_3D::get_OtherRankAbs(int ix, int iy, int iz) const
{
	assert(ix >= 0 && ix < nproc[0] && iy >= 0 && iy < nproc[1] && iz >= 0 && iz < nproc[2]);
	int c[3] = {ix, iy, iz}, r;
	MPI_Cart_rank(comm3d, c, &r);
	return r;
}

bool mpi_manager_3D::is_OuterBoundary(int boundary) const
{
	// BICGStab3D/mpi_manager.C:467-515
	assert(boundary >= 0 && boundary < 6);
	const int nb[6] = {left, right, front, back, bottom, top};
	return nb[boundary] == MPI_PROC_NULL;
}

void mpi_manager_3D::Finalise()
{
	// BICGStab3D/mpi_manager.C:459-465
	if (rank == 0) std::cerr << " Ending the program " << std::endl;
	MPI_Finalize();
	exit(-5);
}
