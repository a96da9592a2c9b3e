#include "BoundaryHandler.H"

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

BoundaryHandler3D::BoundaryHandler3D(int type) : BoundaryHandler(3, type) {}

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
	if (keepBoundVals) return;   // serial build: only inter-process halos would move (BoundaryHandler.C:417,443)
	const int r0 = -d.getLow(0);
	const int mx = d.getHigh(0) - r0, my = d.getHigh(1) - r0, mz = d.getHigh(2) - r0;
	const int untouched = -1;

	if (dir == -1 || dir == 0) {
		// x faces cover the ghost rows of y and z (BoundaryHandler.C:420-421)
		for (int j = -rim; j <= my + rim; ++j)
			for (int k = -rim; k <= mz + rim; ++k)
				fill_line([&](int p) -> double & { return d(p, j, k); }, mx, rim, bc_Type[0], bc_Type[1], false);
	}
	if (dir == -1 || dir == 1) {
		// y faces: x restricted to 0..mx (BoundaryHandler.C:479-480)
		for (int i = 0; i <= mx; ++i)
			for (int k = -rim; k <= mz + rim; ++k)
				fill_line([&](int p) -> double & { return d(i, p, k); }, my, rim, bc_Type[2], bc_Type[3], false);
	}
	if (dir == -1 || dir == 2) {
		// z faces: x and y restricted to 0..m (BoundaryHandler.C:529-530)
		for (int j = 0; j <= my; ++j)
			for (int i = 0; i <= mx; ++i)
				fill_line([&](int p) -> double & { return d(i, j, p); }, mz, rim, bc_Type[4], bc_Type[5], true);
	}
	(void)untouched;
}
