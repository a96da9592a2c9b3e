// Field files (HDF5, group "/Data"); layout described in Hdf5File_cbase.H.
// Written against the serial HDF5 C API (1.8+: the *2 calls).
#include "Hdf5File_cbase.H"

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

namespace {

// HDF5 handles closed on scope exit
template <herr_t (*Close)(hid_t)> struct Handle {
	hid_t id;
	explicit Handle(hid_t i) : id(i) {}
	~Handle() { if (id >= 0) Close(id); }
	operator hid_t() const { return id; }
	Handle(const Handle &) = delete;
	Handle &operator=(const Handle &) = delete;
};
typedef Handle<H5Sclose> Space;
typedef Handle<H5Tclose> Type;
typedef Handle<H5Aclose> Attr;
typedef Handle<H5Dclose> Dset;

// little-endian copy of a native type (what every writer of the format stores)
hid_t le_type(hid_t native)
{
	hid_t t = H5Tcopy(native);
	H5Tset_order(t, H5T_ORDER_LE);
	return t;
}

template <class T> hid_t native_of();
template <> hid_t native_of<double>() { return H5T_NATIVE_DOUBLE; }
template <> hid_t native_of<float>() { return H5T_NATIVE_FLOAT; }
template <> hid_t native_of<int>() { return H5T_NATIVE_INT; }
template <> hid_t native_of<unsigned long>() { return H5T_NATIVE_ULONG; }

// attribute with `count` elements (0: scalar dataspace) on `loc`
bool put_attr(hid_t loc, const std::string &name, hid_t native, const void *buf, int count)
{
	Type t(le_type(native));
	hsize_t n = (hsize_t)count;
	Space sp(count > 0 ? H5Screate_simple(1, &n, NULL) : H5Screate(H5S_SCALAR));
	Attr a(H5Acreate2(loc, name.c_str(), t, sp, H5P_DEFAULT, H5P_DEFAULT));
	return a >= 0 && H5Awrite(a, t, buf) >= 0;
}

bool set_attr(hid_t loc, const std::string &name, hid_t native, const void *buf)
{
	Attr a(H5Aopen(loc, name.c_str(), H5P_DEFAULT));
	return a >= 0 && H5Awrite(a, native, buf) >= 0;
}

bool get_attr(hid_t loc, const std::string &name, hid_t native, void *buf)
{
	Attr a(H5Aopen(loc, name.c_str(), H5P_DEFAULT));
	return a >= 0 && H5Aread(a, native, buf) >= 0;
}

// dataset of `rank` dimensions written from contiguous memory
bool put_dataset(hid_t group, const std::string &name, hid_t native, int rank, const hsize_t *dims, const void *buf,
                 hid_t *keep = nullptr)
{
	Type t(le_type(native));
	Space sp(H5Screate_simple(rank, dims, NULL));
	hid_t d = H5Dcreate2(group, name.c_str(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
	if (d < 0) return false;
	bool ok = H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) >= 0;
	if (keep) *keep = d;
	else H5Dclose(d);
	return ok;
}

// storage extents of a 3-D matrix in file order: slowest index first
template <class T> void file_dims(const NumMatrix<T, 3> &m, hsize_t dims[3])
{
	for (int q = 0; q < 3; ++q) dims[q] = (hsize_t)(m.getHigh(2 - q) - m.getLow(2 - q) + 1);
}

// a field with its origin/delta attributes in the element type A
template <class T, class A>
bool put_field(hid_t group, const std::string &name, NumMatrix<T, 3> &data, const double *xb, const double *dx,
               bool with_grid)
{
	hsize_t dims[3];
	file_dims(data, dims);
	hid_t d = -1;
	bool ok = put_dataset(group, name, native_of<T>(), 3, dims, data.data(), &d);
	if (d < 0) return false;
	Dset ds(d);
	if (with_grid) {
		A origin[3], delta[3];
		for (int q = 0; q < 3; ++q) {
			origin[q] = (A)xb[q];
			delta[q] = (A)dx[q];
		}
		ok = put_attr(ds, "origin", native_of<A>(), origin, 3) && ok;
		ok = put_attr(ds, "delta", native_of<A>(), delta, 3) && ok;
	}
	return ok;
}

// extents of a dataset as stored (file order); exits like the reference on a rank mismatch
std::vector<hsize_t> stored_dims(hid_t dataset, int want_rank)
{
	Space sp(H5Dget_space(dataset));
	int rank = H5Sget_simple_extent_ndims(sp);
	if (want_rank > 0 && rank != want_rank) {
		std::cerr << " Wrong dimensionality of input data: " << std::endl;      // Hdf5File_cbase.C:1108-1111
		std::cerr << rank << " " << want_rank << std::endl;
		exit(-2);
	}
	std::vector<hsize_t> d(rank > 0 ? rank : 0);
	if (rank > 0) H5Sget_simple_extent_dims(sp, d.data(), NULL);
	return d;
}

std::string entry_attr_name(int n)
{
	char buf[32];
	snprintf(buf, sizeof buf, "Name_om%2.2d", n);
	return buf;
}

} // namespace

// ------------------------------------------------------------------ writer

Hdf5Stream::Hdf5Stream(std::string filename, int NumEntries, int, bool use_MPI_IO_
#ifdef parallel
                       , MPI_Comm comm_
#endif
                       )
    : filename(filename), NumEntries(NumEntries), num(0), open(true)
{
#ifdef parallel
	comm = comm_;
	use_MPI_IO = use_MPI_IO_;
	comm_rank = 0;
	comm_size = 1;
	if (use_MPI_IO) {
		MPI_Comm_rank(comm, &comm_rank);
		MPI_Comm_size(comm, &comm_size);
	}
	if (!on_disk()) {
		// not the owner of the shared file: the same calls work on a file that lives in memory only
		hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
		H5Pset_fapl_core(fapl, 1 << 20, 0);
		hdf5file = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
		H5Pclose(fapl);
	} else
#else
	(void)use_MPI_IO_;
#endif
	hdf5file = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
	group = H5Gcreate2(hdf5file, "/Data", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
	const int one = 1;
	// stored with the native types, not the little-endian copies (Hdf5File_cbase.C:58, 76)
	Space sp(H5Screate(H5S_SCALAR));
	{
		Attr a(H5Acreate2(group, "Entries", H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT));
		H5Awrite(a, H5T_NATIVE_INT, &NumEntries);
	}
	{
		Attr a(H5Acreate2(group, "using_cbase", H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT));
		H5Awrite(a, H5T_NATIVE_INT, &one);
	}
}

Hdf5Stream::~Hdf5Stream()
{
	if (open) close();
}

bool Hdf5Stream::close()
{
	if (!open) return true;
	set_attr(group, "Entries", H5T_NATIVE_INT, &num);      // the count actually written (:2066-2070)
	H5Gclose(group);
	H5Fclose(hdf5file);
	open = false;
	return true;
}

void Hdf5Stream::Reopen()
{
	if (open) return;
#ifdef parallel
	if (!on_disk()) {
		// the memory-only file of a non-owner is gone with close(): start a fresh one
		hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
		H5Pset_fapl_core(fapl, 1 << 20, 0);
		hdf5file = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
		H5Pclose(fapl);
		group = H5Gcreate2(hdf5file, "/Data", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
		Space sp(H5Screate(H5S_SCALAR));
		Attr a(H5Acreate2(group, "Entries", H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT));
		H5Awrite(a, H5T_NATIVE_INT, &num);
		open = true;
		return;
	}
#endif
	hdf5file = H5Fopen(filename.c_str(), H5F_ACC_RDWR, H5P_DEFAULT);
	group = H5Gopen2(hdf5file, "Data", H5P_DEFAULT);
	open = true;
}

bool Hdf5Stream::AddGlobalAttr(std::string n, double v) { return put_attr(group, n, H5T_NATIVE_DOUBLE, &v, 0); }
bool Hdf5Stream::AddGlobalAttr(std::string n, float v) { return put_attr(group, n, H5T_NATIVE_FLOAT, &v, 0); }
bool Hdf5Stream::AddGlobalAttr(std::string n, int v) { return put_attr(group, n, H5T_NATIVE_INT, &v, 0); }
bool Hdf5Stream::AddGlobalAttr(std::string n, double *v, int c) { return put_attr(group, n, H5T_NATIVE_DOUBLE, v, c); }
bool Hdf5Stream::AddGlobalAttr(std::string n, float *v, int c) { return put_attr(group, n, H5T_NATIVE_FLOAT, v, c); }
bool Hdf5Stream::AddGlobalAttr(std::string n, unsigned long *v, int c) { return put_attr(group, n, native_of<unsigned long>(), v, c); }
bool Hdf5Stream::AddGlobalAttr(std::string n, int *v, int c) { return put_attr(group, n, H5T_NATIVE_INT, v, c); }

bool Hdf5Stream::AddGlobalAttr(std::string AttrName, std::string AttrData)
{
	// variable-length C string (:306-320)
	Type t(H5Tcopy(H5T_C_S1));
	H5Tset_size(t, H5T_VARIABLE);
	Space sp(H5Screate(H5S_SCALAR));
	Attr a(H5Acreate2(group, AttrName.c_str(), t, sp, H5P_DEFAULT, H5P_DEFAULT));
	const char *p = AttrData.c_str();
	return a >= 0 && H5Awrite(a, t, &p) >= 0;
}

bool Hdf5Stream::ChangeGlobalAttr(std::string n, double v) { return set_attr(group, n, H5T_NATIVE_DOUBLE, &v); }
bool Hdf5Stream::ChangeGlobalAttr(std::string n, float v) { return set_attr(group, n, H5T_NATIVE_FLOAT, &v); }
bool Hdf5Stream::ChangeGlobalAttr(std::string n, int v) { return set_attr(group, n, H5T_NATIVE_INT, &v); }

template <class T> static bool dataset_attr(hid_t group, const std::string &array, const std::string &name, T v)
{
	Dset d(H5Dopen2(group, array.c_str(), H5P_DEFAULT));
	return d >= 0 && put_attr(d, name, native_of<T>(), &v, 0);
}
bool Hdf5Stream::AddAttributeToArray(std::string a, const std::string &n, int v) { return dataset_attr(group, a, n, v); }
bool Hdf5Stream::AddAttributeToArray(std::string a, const std::string &n, float v) { return dataset_attr(group, a, n, v); }
bool Hdf5Stream::AddAttributeToArray(std::string a, const std::string &n, double v) { return dataset_attr(group, a, n, v); }

bool Hdf5Stream::AddDatasetName(std::string &DatasetName)
{
	// fixed-length string that includes the terminating NUL (:394-399)
	Type t(H5Tcopy(H5T_C_S1));
	H5Tset_size(t, DatasetName.size() + 1);
	Space sp(H5Screate(H5S_SCALAR));
	Attr a(H5Acreate2(group, entry_attr_name(num).c_str(), t, sp, H5P_DEFAULT, H5P_DEFAULT));
	return a >= 0 && H5Awrite(a, t, DatasetName.c_str()) >= 0;
}

// The field writers register the name and count the dataset, without testing the count
// against NumEntries (:645-647, 1179-1180).
bool Hdf5Stream::Write3DMatrix(std::string ArrayName, NumMatrix<double, 3> &data, double *xb, double *dx)
{
	AddDatasetName(ArrayName);
	num += 1;
	return put_field<double, double>(group, ArrayName, data, xb, dx, true);
}

// (The reference forms the float extents with the lower bound of direction 2 throughout,
// :1175-1177; all fields of this code family have equal lower bounds, for which that is the
// true extent used here.)
bool Hdf5Stream::Write3DMatrix(std::string ArrayName, NumMatrix<float, 3> &data, double *xb, double *dx,
                               bool with_opendxinfo)
{
	AddDatasetName(ArrayName);
	num += 1;
	return put_field<float, float>(group, ArrayName, data, xb, dx, with_opendxinfo);
}

bool Hdf5Stream::Write3DMatrix(std::string ArrayName, NumMatrix<float, 3> &data)
{
	return Write3DMatrix(ArrayName, data, nullptr, nullptr, false);
}

#ifdef parallel
bool Hdf5Stream::Write3DMatrix_withMPI_IO(std::string ArrayName, NumMatrix<float, 3> &data, NumArray<int> &mx_global,
                                          NumArray<int> &mx_local, NumArray<int> &rank_shift, NumArray<float> &xb,
                                          NumArray<float> &dx, bool with_opendxinfo)
{
	// BICGStab3D/Hdf5File_cbase.C:1241-1365
	const int rim = data.getLow(0) < 0 ? -data.getLow(0) : 0;
	AddDatasetName(ArrayName);
	num += 1;
	if (!use_MPI_IO) {
		std::cerr << " Write3DMatrix_withMPI_IO: the stream was opened without use_MPI_IO" << std::endl;
		return false;
	}
	// block geometry in file order (slowest index first), gathered on the owner
	int geo[6];
	for (int q = 0; q < 3; ++q) {
		geo[q] = mx_local[2 - q] + 1 + 2 * rim;
		geo[3 + q] = rank_shift[2 - q];
	}
	// A process whose field does not have the extents it announces still takes part in every
	// collective step: it announces a block of -1 (nothing is sent for it) and all return false together.
	const long count = (long)geo[0] * geo[1] * geo[2];
	const bool fits = (long)data.size() == count;
	if (!fits) {
		std::cerr << " Write3DMatrix_withMPI_IO: the field's extents do not match mx_local and its rim" << std::endl;
		geo[0] = -1;
	}
	std::vector<int> all(on_disk() ? 6 * comm_size : 0);
	MPI_Gather(geo, 6, MPI_INT, all.data(), 6, MPI_INT, 0, comm);
	bool ok = fits;
	if (!on_disk()) {
		if (fits) MPI_Send(data.data(), (int)count, MPI_FLOAT, 0, 4711, comm);
	} else {
		hsize_t dims[3];
		for (int q = 0; q < 3; ++q) dims[q] = (hsize_t)(mx_global[2 - q] + 1 + 2 * rim);
		Type t(le_type(H5T_NATIVE_FLOAT));
		Space file_space(H5Screate_simple(3, dims, NULL));
		for (int r = 0; r < comm_size; ++r) ok = ok && all[6 * r] >= 0;   // no dataset unless every block is valid
		Dset ds(ok ? H5Dcreate2(group, ArrayName.c_str(), t, file_space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT) : -1);
		if (ds < 0) ok = false;
		if (ok && with_opendxinfo) {
			float origin[3], delta[3];
			for (int q = 0; q < 3; ++q) { origin[q] = xb[q]; delta[q] = dx[q]; }
			ok = put_attr(ds, "origin", H5T_NATIVE_FLOAT, origin, 3) && put_attr(ds, "delta", H5T_NATIVE_FLOAT, delta, 3);
		}
		std::vector<float> buf;
		for (int r = 0; r < comm_size; ++r) {
			const int *g = &all[6 * r];
			if (g[0] < 0) continue;                  // that process holds no valid block and sends none
			const long n = (long)g[0] * g[1] * g[2];
			const float *src = data.data();
			if (r != 0) {
				buf.resize((size_t)n);
				MPI_Recv(buf.data(), (int)n, MPI_FLOAT, r, 4711, comm, MPI_STATUS_IGNORE);
				src = buf.data();
			}
			hsize_t size[3], off[3];
			for (int q = 0; q < 3; ++q) { size[q] = (hsize_t)g[q]; off[q] = (hsize_t)g[3 + q]; }
			if (!ok) continue;                      // keep receiving so that no sender is left waiting
			for (int q = 0; q < 3; ++q) ok = ok && off[q] + size[q] <= dims[q];
			if (!ok) continue;
			Space mem(H5Screate_simple(3, size, NULL));
			ok = H5Sselect_hyperslab(file_space, H5S_SELECT_SET, off, NULL, size, NULL) >= 0 &&
			     H5Dwrite(ds, t, mem, file_space, H5P_DEFAULT, src) >= 0;
		}
	}
	int mine = ok ? 1 : 0, every = 0;
	MPI_Allreduce(&mine, &every, 1, MPI_INT, MPI_MIN, comm);
	return every == 1;
}
#endif

bool Hdf5Stream::Write3DMatrixSwap(std::string ArrayName, NumMatrix<double, 3> &data, double *xb, double *dx)
{
	// transposed copy: element (k, j, i) of the copy is element (i, j, k) of the field (:1756-1775)
	NumMatrix<double, 3> t(Index::set(data.getLow(2), data.getLow(1), data.getLow(0)),
	                       Index::set(data.getHigh(2), data.getHigh(1), data.getHigh(0)));
	for (int k = data.getLow(2); k <= data.getHigh(2); ++k)
		for (int j = data.getLow(1); j <= data.getHigh(1); ++j)
			for (int i = data.getLow(0); i <= data.getHigh(0); ++i) t(k, j, i) = data(i, j, k);
	num += 1;
	if (num > NumEntries) return false;
	return put_field<double, double>(group, ArrayName, t, xb, dx, true);      // no name entry (:1783-1786)
}

template <class T> static bool plain_array(hid_t group, int &num, int cap, const std::string &name, const T *data, int max)
{
	num += 1;
	if (num > cap) return false;
	hsize_t n = (hsize_t)max;
	return put_dataset(group, name, native_of<T>(), 1, &n, data);
}
static std::string numbered(int n)
{
	char buf[32];
	snprintf(buf, sizeof buf, "data%3.3d", n);
	return buf;
}
bool Hdf5Stream::WriteArray(std::string n, int *d, int max) { return plain_array(group, num, NumEntries, n, d, max); }
bool Hdf5Stream::WriteArray(std::string n, float *d, int max) { return plain_array(group, num, NumEntries, n, d, max); }
bool Hdf5Stream::WriteArray(std::string n, double *d, int max) { return plain_array(group, num, NumEntries, n, d, max); }
bool Hdf5Stream::WriteArray(int *d, int max) { return WriteArray(numbered(num + 1), d, max); }
bool Hdf5Stream::WriteArray(float *d, int max) { return WriteArray(numbered(num + 1), d, max); }
bool Hdf5Stream::WriteArray(double *d, int max) { return WriteArray(numbered(num + 1), d, max); }

bool Hdf5Stream::WriteNDArray(std::string ArrayName, float *data, int mx[], int dim)
{
	num += 1;
	if (num > NumEntries) return false;
	std::vector<hsize_t> dims(mx, mx + dim);      // as given, no reversal (:2043-2046)
	return put_dataset(group, ArrayName, H5T_NATIVE_FLOAT, dim, dims.data(), data);
}

// ------------------------------------------------------------------ reader

Hdf5iStream::Hdf5iStream(std::string filename, int rank)
{
	if (rank == 0) std::cout << " Opening file: " << filename;
	hdf5file = H5Fopen(filename.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
	group = H5Gopen2(hdf5file, "Data", H5P_DEFAULT);
	if (rank == 0) std::cout << " ...done " << std::endl;
	NumEntries = 0;
	get_attr(group, "Entries", H5T_NATIVE_INT, &NumEntries);
}

Hdf5iStream::~Hdf5iStream()
{
	H5Gclose(group);
	H5Fclose(hdf5file);
}

bool Hdf5iStream::close() { return true; }      // the handles live until destruction (:2080-2086)

bool Hdf5iStream::Read3DMatrix(std::string ArrayName, NumMatrix<double, 3> &data)
{
	Dset d(H5Dopen2(group, ArrayName.c_str(), H5P_DEFAULT));
	std::vector<hsize_t> dims = stored_dims(d, 3);
	hsize_t want[3];
	file_dims(data, want);
	for (int q = 0; q < 3; ++q)
		if (dims[q] != want[q]) {      // reported, not fatal (:1120-1125)
			std::cerr << " Wrong size of dimension " << q << ":" << std::endl;
			std::cerr << (int)dims[q] << " " << (int)want[q] << std::endl;
		}
	return H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.data()) >= 0;
}

static bool read_float_field(hid_t group, const std::string &name, NumMatrix<float, 3> &data, double *xb, double *dx)
{
	Dset d(H5Dopen2(group, name.c_str(), H5P_DEFAULT));
	if (xb && dx) {
		float v[3];
		get_attr(d, "origin", H5T_NATIVE_FLOAT, v);
		for (int q = 0; q < 3; ++q) xb[q] = (double)v[q];
		get_attr(d, "delta", H5T_NATIVE_FLOAT, v);
		for (int q = 0; q < 3; ++q) dx[q] = (double)v[q];
	}
	std::vector<hsize_t> dims = stored_dims(d, 3);
	int lo[3] = {0, 0, 0}, hi[3];
	for (int q = 0; q < 3; ++q) hi[q] = (int)dims[2 - q] - 1;
	data.resize(lo, hi);
	return H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.data()) >= 0;
}

bool Hdf5iStream::Read3DMatrix(std::string ArrayName, NumMatrix<float, 3> &data)
{
	return read_float_field(group, ArrayName, data, nullptr, nullptr);
}

bool Hdf5iStream::Read3DMatrix(std::string ArrayName, NumMatrix<float, 3> &data, double *xb, double *dx)
{
	return read_float_field(group, ArrayName, data, xb, dx);
}

float Hdf5iStream::ReadPointFrom3DMatrix(std::string ArrayName, int ix, int iy, int iz)
{
	Dset d(H5Dopen2(group, ArrayName.c_str(), H5P_DEFAULT));
	stored_dims(d, 3);
	Space file(H5Dget_space(d));
	hsize_t one = 1, coord[3] = {(hsize_t)iz, (hsize_t)iy, (hsize_t)ix};
	Space mem(H5Screate_simple(1, &one, NULL));
	H5Sselect_elements(file, H5S_SELECT_SET, 1, coord);
	float v = 0.f;
	H5Dread(d, H5T_NATIVE_FLOAT, mem, file, H5P_DEFAULT, &v);
	std::cout << " Data: " << v << std::endl;      // :1676
	return v;
}

void Hdf5iStream::getSize(std::string ArrayName, int mx[], int DIM)
{
	Dset d(H5Dopen2(group, ArrayName.c_str(), H5P_DEFAULT));
	std::vector<hsize_t> dims = stored_dims(d, DIM);
	for (int i = 0; i < DIM; ++i) mx[DIM - i - 1] = (int)dims[i];
}

bool Hdf5iStream::ReadGlobalAttr(std::string n, double &v) { return get_attr(group, n, H5T_NATIVE_DOUBLE, &v); }
bool Hdf5iStream::ReadGlobalAttr(std::string n, float &v) { return get_attr(group, n, H5T_NATIVE_FLOAT, &v); }
bool Hdf5iStream::ReadGlobalAttr(std::string n, unsigned long &v) { return get_attr(group, n, H5T_NATIVE_ULONG, &v); }
bool Hdf5iStream::ReadGlobalAttr(std::string n, int &v) { return get_attr(group, n, H5T_NATIVE_INT, &v); }

float Hdf5iStream::ReadDatasetAttr(std::string DataSetName, std::string AttrName)
{
	Dset d(H5Dopen2(group, DataSetName.c_str(), H5P_DEFAULT));
	float v = 0.f;
	get_attr(d, AttrName, H5T_NATIVE_FLOAT, &v);
	return v;
}

bool Hdf5iStream::ReadDatasetGrid(std::string DataSetName, NumArray<float> &xPos, NumArray<float> &yPos,
                                  NumArray<float> &zPos)
{
	Dset d(H5Dopen2(group, DataSetName.c_str(), H5P_DEFAULT));
	float xb[3], dx[3];
	if (!get_attr(d, "origin", H5T_NATIVE_FLOAT, xb) || !get_attr(d, "delta", H5T_NATIVE_FLOAT, dx)) return false;
	std::vector<hsize_t> dims = stored_dims(d, 0);
	if (dims.size() != 3) return false;
	NumArray<float> *pos[3] = {&xPos, &yPos, &zPos};
	for (int q = 0; q < 3; ++q) {
		const int n = (int)dims[2 - q];
		pos[q]->resize(n);
		for (int i = 0; i < n; ++i) (*pos[q])[i] = xb[q] + dx[q] * i;
	}
	return true;
}

int Hdf5iStream::GetDatasetDimension(std::string DataSetName)
{
	Dset d(H5Dopen2(group, DataSetName.c_str(), H5P_DEFAULT));
	return (int)stored_dims(d, 0).size();
}

NumArray<int> Hdf5iStream::GetDatasetExtent(std::string DataSetName)
{
	Dset d(H5Dopen2(group, DataSetName.c_str(), H5P_DEFAULT));
	std::vector<hsize_t> dims = stored_dims(d, 0);
	const int n = (int)dims.size();
	NumArray<int> Nx(n);
	for (int i = 0; i < n; ++i) Nx[i] = (int)dims[n - (i + 1)];      // fastest index first (:904-907)
	return Nx;
}

std::string Hdf5iStream::GetDatasetName(int n)
{
	Attr a(H5Aopen(group, entry_attr_name(n).c_str(), H5P_DEFAULT));
	if (a < 0) return "";
	Type ft(H5Aget_type(a));
	if (H5Tis_variable_str(ft) > 0) {
		Type mt(H5Tget_native_type(ft, H5T_DIR_ASCEND));
		char *s = nullptr;
		H5Aread(a, mt, &s);
		std::string r = s ? s : "";
		if (s) H5free_memory(s);
		return r;
	}
	std::vector<char> buf(H5Tget_size(ft) + 1, '\0');
	H5Aread(a, ft, buf.data());
	return buf.data();
}

bool Hdf5iStream::doesAttrExist(std::string name) const { return H5Aexists(group, name.c_str()) > 0; }
