// C entry points over Hdf5Stream / Hdf5iStream for the tests (ctypes).
#include <cstring>
#include <string>

#include "Hdf5File_cbase.H"

namespace {
void fill(NumMatrix<double, 3> &m, const int mx[3], int rim, const double *src)
{
	m.resize(Index::set(-rim, -rim, -rim), Index::set(mx[0] + rim, mx[1] + rim, mx[2] + rim));
	if (src) memcpy(m.data(), src, m.size() * sizeof(double));
}
} // namespace

extern "C" {

// Writes the fields of a solve the way a driver of this code family would: "phi" and "rhs"
// as double fields, "phi_f" as a float field, "phi_T" transposed, two plain arrays and a
// few attributes.  Fields are ghost-inclusive, x fastest, extents mx[d] + 1 + 2*rim.
int compat_h5_write(const char *path, const int mx[3], int rim, const double *phi, const double *rhs,
                    const double xb[3], const double dx[3], int iterations, double residual)
{
	NumMatrix<double, 3> a, b;
	fill(a, mx, rim, phi);
	fill(b, mx, rim, rhs);
	NumMatrix<float, 3> f;
	f.resize(Index::set(-rim, -rim, -rim), Index::set(mx[0] + rim, mx[1] + rim, mx[2] + rim));
	for (size_t i = 0; i < f.size(); ++i) f.data()[i] = (float)phi[i];
	double o[3] = {xb[0], xb[1], xb[2]}, d[3] = {dx[0], dx[1], dx[2]};
	Hdf5Stream out(path, 6);
	bool ok = out.AddGlobalAttr("iterations", iterations);
	ok = out.AddGlobalAttr("residual", residual) && ok;
	ok = out.AddGlobalAttr("tolerance", 1.e-6f) && ok;
	ok = out.AddGlobalAttr("solver", std::string("BiCGStab(l)")) && ok;
	int m3[3] = {mx[0], mx[1], mx[2]};
	ok = out.AddGlobalAttr("mx", m3, 3) && ok;
	ok = out.AddGlobalAttr("delx", d, 3) && ok;
	ok = out.Write3DMatrix("phi", a, o, d) && ok;
	ok = out.Write3DMatrix("rhs", b, o, d) && ok;
	ok = out.Write3DMatrix("phi_f", f, o, d) && ok;
	ok = out.Write3DMatrixSwap("phi_T", a, o, d) && ok;
	ok = out.AddAttributeToArray("phi", "rim", rim) && ok;
	ok = out.AddAttributeToArray("phi", "scale", 2.5) && ok;
	double hist[3] = {1., 0.5, residual};
	ok = out.WriteArray("history", hist, 3) && ok;
	int flags[2] = {7, 9};
	ok = out.WriteArray(flags, 2) && ok;                // sixth dataset, named data006
	bool over = out.WriteArray("one_too_many", hist, 3);   // seventh: refused by the entry cap
	ok = out.ChangeGlobalAttr("iterations", iterations + 1) && ok;
	out.close();
	out.Reopen();
	ok = out.ChangeGlobalAttr("residual", residual * 2.) && ok;
	out.close();
	return (ok ? 0 : 1) | (over ? 2 : 0);
}

// Reads back what compat_h5_write stored.  phi/rhs: ghost-inclusive doubles; phi_f: floats.
int compat_h5_read(const char *path, const int mx[3], int rim, double *phi, double *rhs, float *phi_f,
                   double xb_f[3], double dx_f[3], int *iterations, double *residual, int extent[3],
                   int size_rev[3], float *point, int *entries_ok, char *names, int names_cap, float *grid_x1)
{
	Hdf5iStream in(path, 1);
	NumMatrix<double, 3> a, b;
	fill(a, mx, rim, nullptr);
	fill(b, mx, rim, nullptr);
	bool ok = in.Read3DMatrix("phi", a);
	ok = in.Read3DMatrix("rhs", b) && ok;
	memcpy(phi, a.data(), a.size() * sizeof(double));
	memcpy(rhs, b.data(), b.size() * sizeof(double));
	NumMatrix<float, 3> f;
	ok = in.Read3DMatrix("phi_f", f, xb_f, dx_f) && ok;
	if (f.getLow(0) != 0 || f.getHigh(0) != mx[0] + 2 * rim) ok = false;
	memcpy(phi_f, f.data(), f.size() * sizeof(float));
	ok = in.ReadGlobalAttr("iterations", *iterations) && ok;
	ok = in.ReadGlobalAttr("residual", *residual) && ok;
	NumArray<int> e = in.GetDatasetExtent("phi");
	for (int q = 0; q < 3; ++q) extent[q] = e[q];
	in.getSize("phi", size_rev, 3);
	*point = in.ReadPointFrom3DMatrix("phi_f", 1, 2, 3);
	*entries_ok = in.doesAttrExist("Entries") && !in.doesAttrExist("nonsense") && in.GetDatasetDimension("history") == 1;
	std::string all = in.GetDatasetName(0) + "," + in.GetDatasetName(1) + "," + in.GetDatasetName(2);
	strncpy(names, all.c_str(), names_cap - 1);
	names[names_cap - 1] = 0;
	NumArray<float> gx, gy, gz;
	ok = in.ReadDatasetGrid("phi_f", gx, gy, gz) && ok;
	*grid_x1 = gx[1];
	if (in.ReadDatasetAttr("phi", "scale") != 2.5f) ok = false;
	return ok ? 0 : 1;
}

} // extern "C"
