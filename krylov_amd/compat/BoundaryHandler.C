#include "BoundaryHandler.H"

#include <vector>

BoundaryHandler::BoundaryHandler(int DIM_, int bcType, int test_case) : eps(1.e-42), DIM(DIM_), is_testRun(test_case > 0)
{
	bc_Type.resize(2 * DIM);
	for (int b = 0; b < 2 * DIM; ++b) bc_Type(b) = bcType;
}

void BoundaryHandler::set_bcType(int bcType)
{
	for (int b = 0; b < 2 * DIM; ++b) bc_Type(b) = bcType;
}

void BoundaryHandler::set_bcType(int bound, int bcType)
{
	assert(bound >= 0 && bound < 2 * DIM);
	bc_Type(bound) = bcType;
}

int BoundaryHandler::get_bcType(int bound) const
{
	assert(bound >= 0 && bound < 2 * DIM);
	return bc_Type(bound);
}

#ifdef parallel
BoundaryHandler3D::BoundaryHandler3D(int type) : BoundaryHandler(3, type), have_mpi(false) {}
BoundaryHandler3D::BoundaryHandler3D(mpi_manager_3D MyMPI_, int type) : BoundaryHandler(3, type), MyMPI(MyMPI_), have_mpi(true) {}
#else
BoundaryHandler3D::BoundaryHandler3D(int type) : BoundaryHandler(3, type) {}
#endif

namespace {

// One face pair along a line of the array: v(p) for p = -rim..m+rim with stride
// access through `at`.  lo/hi: face types.  zlow: the low-z face of the
// reference also rewrites the boundary point, after the ghost.
template <class At> void fill_line(At at, int m, int rim, int lo, int hi, bool low_rewrites_boundary)
{
	if (lo == 0) {
		for (int p = -rim; p <= 0; ++p) at(p) = 0.;
	} else if (lo == 1) {
		if (low_rewrites_boundary) {
			for (int p = -rim; p <= 0; ++p) at(p) = 2. * at(p + 1) - at(p + 2);   // ascending: ghost sees the old boundary value
		} else {
			for (int p = -1; p >= -rim; --p) at(p) = 2. * at(p + 1) - at(p + 2);
		}
	}
	if (hi == 0) {
		for (int p = m; p <= m + rim; ++p) at(p) = 0.;
	} else if (hi == 1) {
		for (int p = m; p <= m + rim; ++p) at(p) = 2. * at(p - 1) - at(p - 2);
	}
}

} // namespace

void BoundaryHandler3D::do_BCs(NumMatrix<double, 3> &d, int rim, int dir, bool keepBoundVals)
{
	const int r0 = -d.getLow(0);
	const int mx = d.getHigh(0) - r0, my = d.getHigh(1) - r0, mz = d.getHigh(2) - r0;
	// face types in effect: a face towards a neighbour process is left to the exchange,
	// keepBoundVals leaves the outer faces alone (BoundaryHandler.C:417, 443)
	int t[6];
	for (int f = 0; f < 6; ++f) {
		t[f] = keepBoundVals ? -1 : bc_Type[f];
#ifdef parallel
		if (have_mpi && !MyMPI.is_OuterBoundary(f)) t[f] = -1;
#endif
	}

	if (dir == -1 || dir == 0) {
		// x faces cover the ghost rows of y and z (BoundaryHandler.C:420-421)
		for (int j = -rim; j <= my + rim; ++j)
			for (int k = -rim; k <= mz + rim; ++k)
				fill_line([&](int p) -> double & { return d(p, j, k); }, mx, rim, t[0], t[1], false);
#ifdef parallel
		if (have_mpi) do_bc_MPI(d, 0, rim);
#endif
	}
	if (dir == -1 || dir == 1) {
		// y faces: x restricted to 0..mx (BoundaryHandler.C:479-480)
		for (int i = 0; i <= mx; ++i)
			for (int k = -rim; k <= mz + rim; ++k)
				fill_line([&](int p) -> double & { return d(i, p, k); }, my, rim, t[2], t[3], false);
#ifdef parallel
		if (have_mpi) do_bc_MPI(d, 1, rim);
#endif
	}
	if (dir == -1 || dir == 2) {
		// z faces: x and y restricted to 0..m (BoundaryHandler.C:529-530)
		for (int j = 0; j <= my; ++j)
			for (int i = 0; i <= mx; ++i)
				fill_line([&](int p) -> double & { return d(i, j, p); }, mz, rim, t[4], t[5], true);
#ifdef parallel
		if (have_mpi) do_bc_MPI(d, 2, rim);
#endif
	}
}

#ifdef parallel
// Planes 1..rim go to the lower neighbour's ghosts m+1..m+rim, planes m-rim..m-1 to
// the upper neighbour's ghosts -rim..-1, over the full tangential box
// (BoundaryHandler.C:579-1045; the interface plane itself is held by both sides).
void BoundaryHandler3D::do_bc_MPI(NumMatrix<double, 3> &d, int dir, int rim)
{
	const int lo_nb[3] = {MyMPI.get_left(), MyMPI.get_front(), MyMPI.get_bottom()};
	const int hi_nb[3] = {MyMPI.get_right(), MyMPI.get_back(), MyMPI.get_top()};
	const int lo = lo_nb[dir], hi = hi_nb[dir];
	if (lo == MPI_PROC_NULL && hi == MPI_PROC_NULL) return;
	const int r0 = -d.getLow(0);
	const int m[3] = {d.getHigh(0) - r0, d.getHigh(1) - r0, d.getHigh(2) - r0};
	const int a = (dir + 1) % 3, b = (dir + 2) % 3;
	const int na = m[a] + 2 * r0 + 1, nb = m[b] + 2 * r0 + 1;
	const size_t n = (size_t)na * nb * rim;
	std::vector<double> s_lo(n), s_hi(n), r_lo(n), r_hi(n);
	auto at = [&](int p, int ia, int ib) -> double & {
		int idx[3];
		idx[dir] = p; idx[a] = ia; idx[b] = ib;
		return d(idx[0], idx[1], idx[2]);
	};
	size_t q = 0;
	for (int l = 0; l < rim; ++l)
		for (int ib = -r0; ib <= m[b] + r0; ++ib)
			for (int ia = -r0; ia <= m[a] + r0; ++ia, ++q) {
				s_lo[q] = at(1 + l, ia, ib);
				s_hi[q] = at(m[dir] - rim + l, ia, ib);
			}
	MPI_Sendrecv(s_lo.data(), (int)n, MPI_DOUBLE, lo, 71, r_hi.data(), (int)n, MPI_DOUBLE, hi, 71, MyMPI.comm3d,
	             MPI_STATUS_IGNORE);
	MPI_Sendrecv(s_hi.data(), (int)n, MPI_DOUBLE, hi, 72, r_lo.data(), (int)n, MPI_DOUBLE, lo, 72, MyMPI.comm3d,
	             MPI_STATUS_IGNORE);
	q = 0;
	for (int l = 0; l < rim; ++l)
		for (int ib = -r0; ib <= m[b] + r0; ++ib)
			for (int ia = -r0; ia <= m[a] + r0; ++ia, ++q) {
				if (hi != MPI_PROC_NULL) at(m[dir] + 1 + l, ia, ib) = r_hi[q];
				if (lo != MPI_PROC_NULL) at(-rim + l, ia, ib) = r_lo[q];
			}
}
#endif
