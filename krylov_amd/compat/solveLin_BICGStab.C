#include "solveLin_BICGStab.H"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "../../include/krylov_amd.h"

using std::cerr;
using std::cout;
using std::endl;

BICGStab::BICGStab()
    : handle(NULL), eps(1e-3), LValue(2), max_outer(0), use_spatialDiffusion(false), use_offDiagDiffusion(false),
      verbose(false), n_iter(0), last_status(0), fused(false), norm_final(0.), norm_rhs(0.), solve_ms(0.), stage_ms(0.)
{
	my_type = 20;
	for (int d = 0; d < 3; ++d) { delx[d] = 1.; DiffDiag[d] = 1.; mx[d] = 0; }
	for (int f = 0; f < 6; ++f) handle_bc[f] = -99;
#ifdef parallel
	comm3d = MPI_COMM_NULL;
	for (int d = 0; d < 3; ++d) { nproc_mpi[d] = 1; coord_mpi[d] = 0; }
#endif
	devices.push_back(0);
	// KRY_DEVICES=0,1,2,... : one z slab per listed device
	if (const char *e = getenv("KRY_DEVICES")) {
		std::vector<int> v;
		for (const char *p = e; *p;) {
			char *end;
			long d = strtol(p, &end, 10);
			if (end == p) break;
			v.push_back((int)d);
			p = (*end == ',') ? end + 1 : end;
		}
		if (!v.empty()) devices = v;
	}
}

#ifdef parallel
BICGStab::BICGStab(grid_manager &TheGrid, double tol, int LValue_, mpi_manager_3D &MyMPI, bool, bool) : BICGStab()
{
	use_MPI(MyMPI);
#else
BICGStab::BICGStab(grid_manager &TheGrid, double tol, int LValue_, bool, bool) : BICGStab()
{
#endif
	// BICGStab3D/solveLin_BICGStab.C:104-131
	LValue = LValue_;
	for (int d = 0; d < 3; ++d) {
		delx[d] = TheGrid.get_delx(d);
		mx[d] = TheGrid.get_mx(d);
	}
	eps = tol;
}

BICGStab::~BICGStab() { release(); }

void BICGStab::release()
{
	if (handle) kry_destroy(handle);
	handle = NULL;
	// nothing is resident on a handle that does not exist: the next solve stages every field
	// (a handle created afterwards may sit at the very address this one had)
	last_mode = -1;
	for (int k = 0; k < 5; ++k) last_coef[k] = NULL;
}

#ifdef parallel
// BICGStab3D/solveLin_BICGStab.C:142-147 takes comm3d and the rank from the manager;
// here its split and this process's place in it are kept as well.
void BICGStab::use_MPI(mpi_manager_3D &MyMPI)
{
	comm3d = MyMPI.comm3d;
	rank = MyMPI.get_rank();
	for (int d = 0; d < 3; ++d) { nproc_mpi[d] = MyMPI.get_nproc(d); coord_mpi[d] = MyMPI.get_coord(d); }
}

int BICGStab::mpi_allreduce(void *ctx, double *buf, int n)
{
	// Krylov.C:237-242
	BICGStab *self = static_cast<BICGStab *>(ctx);
	return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, MPI_SUM, self->comm3d) == MPI_SUCCESS ? 0 : 1;
}

int BICGStab::mpi_halo(void *ctx, int peer_lo, const double *send_lo, double *recv_lo, int peer_hi,
                       const double *send_hi, double *recv_hi, unsigned long count)
{
	// BoundaryHandler.C:1633-1677; the library names the neighbours by their rank in the
	// Cartesian communicator (no reordering), a negative rank meaning "none"
	BICGStab *self = static_cast<BICGStab *>(ctx);
	const int lo = peer_lo < 0 ? MPI_PROC_NULL : peer_lo, hi = peer_hi < 0 ? MPI_PROC_NULL : peer_hi;
	int e1 = MPI_Sendrecv(send_lo, (int)count, MPI_DOUBLE, lo, 81, recv_hi, (int)count, MPI_DOUBLE, hi, 81, self->comm3d,
	                      MPI_STATUS_IGNORE);
	int e2 = MPI_Sendrecv(send_hi, (int)count, MPI_DOUBLE, hi, 82, recv_lo, (int)count, MPI_DOUBLE, lo, 82, self->comm3d,
	                      MPI_STATUS_IGNORE);
	return (e1 == MPI_SUCCESS && e2 == MPI_SUCCESS) ? 0 : 1;
}

void BICGStab::setup(grid_1D &xGrid, grid_1D &yGrid, grid_1D &zGrid, double epsilon_, NumArray<int> &solverPars, bool,
                     bool, mpi_manager_3D &MyMPI, int maxIter)
{
	use_MPI(MyMPI);
#else
void BICGStab::setup(grid_1D &xGrid, grid_1D &yGrid, grid_1D &zGrid, double epsilon_, NumArray<int> &solverPars, bool,
                     bool, int maxIter)
{
	// BICGStab3D/solveLin_BICGStab.C:134-157.  maxIter is accepted and ignored
	// there; here a positive value caps the outer passes.
	rank = 0;
#endif
	eps = epsilon_;
	LValue = solverPars(0);
	dim = 3;
	set_Grid(xGrid, yGrid, zGrid);
	if (maxIter > 0) max_outer = maxIter;
	release();
}

void BICGStab::set_Grid(grid_1D &xGrid, grid_1D &yGrid, grid_1D &zGrid)
{
	delx[0] = xGrid.get_del(); delx[1] = yGrid.get_del(); delx[2] = zGrid.get_del();
	mx[0] = xGrid.get_mx(); mx[1] = yGrid.get_mx(); mx[2] = zGrid.get_mx();
}

void BICGStab::set_Devices(const std::vector<int> &d)
{
	if (!d.empty()) devices = d;
	release();
}

void BICGStab::ensure_handle(BoundaryHandler3D &bounds, const NumMatrix<double, 3> &phi)
{
	// fields must be the ghost-inclusive boxes [-1..m+1]^3 of make_Arrays (:175-191)
	for (int d = 0; d < 3; ++d) {
		if (phi.getLow(d) != -1 || phi.getHigh(d) != mx[d] + 1) {
			cerr << " BICGStab: field extent [" << phi.getLow(d) << ".." << phi.getHigh(d) << "] in direction " << d
			     << " does not match the grid (mx = " << mx[d] << ", one ghost layer)" << endl;
			exit(2);
		}
	}
	int bc[6];
	bool same = handle != NULL;
	for (int f = 0; f < 6; ++f) {
		bc[f] = bounds.get_bcType(f);
		same = same && bc[f] == handle_bc[f];
	}
	if (same) return;
	release();
	kry_desc d;
	memset(&d, 0, sizeof d);
	for (int k = 0; k < 3; ++k) { d.mx[k] = mx[k]; d.delx[k] = delx[k]; }
	for (int f = 0; f < 6; ++f) d.bc_type[f] = handle_bc[f] = bc[f];
	d.L = LValue;
	d.eps = eps;
	d.max_outer = max_outer;
#ifdef parallel
	// one sub-domain per process; process r uses the (r mod listed)-th listed device
	for (int k = 0; k < 3; ++k) { d.nproc[k] = nproc_mpi[k]; d.coord[k] = coord_mpi[k]; }
	d.nslabs = nproc_mpi[2];
	d.slab_begin = coord_mpi[2];
	d.slab_count = 1;
	d.devices[0] = devices[rank % devices.size()];
#else
	int ns = (int)devices.size();
	while (ns > 1 && (mx[2] % ns != 0 || mx[2] / ns < 4)) --ns;
	d.nslabs = d.slab_count = ns;
	d.slab_begin = 0;
	for (int i = 0; i < ns; ++i) d.devices[i] = devices[i];
#endif
	int st = 0;
	handle = kry_create(&d, &st);
	if (!handle) {
		cerr << " BICGStab: cannot set up the device solver: " << kry_status_string(st) << " (" << kry_last_error()
		     << ")" << endl;
		exit(2);
	}
#ifdef parallel
	const int ntasks_mpi = nproc_mpi[0] * nproc_mpi[1] * nproc_mpi[2];
	if (ntasks_mpi > 1 &&
	    kry_comm_init_host_callbacks(handle, rank, ntasks_mpi, &BICGStab::mpi_allreduce, &BICGStab::mpi_halo, this) != 0) {
		cerr << " BICGStab: cannot attach the MPI transport (" << kry_last_error() << ")" << endl;
		exit(2);
	}
	// Processes that share a node let their GPUs exchange the dot products and ghost planes among
	// themselves through mapped device memory; MPI stays attached as the fallback and carries the
	// set-up (KRY_MAIL=0 keeps everything on MPI).
	const char *mail_env = getenv("KRY_MAIL");
	if (ntasks_mpi > 1 && !(mail_env && atoi(mail_env) == 0)) {
		MPI_Comm node;
		int node_size = 0;
		MPI_Comm_split_type(comm3d, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
		MPI_Comm_size(node, &node_size);
		MPI_Comm_free(&node);
		int ok = (node_size == ntasks_mpi && ntasks_mpi <= 64) ? 1 : 0, all_ok = 0;
		std::vector<char> mine(KRY_MAIL_HANDLE_BYTES, 0), all((size_t)KRY_MAIL_HANDLE_BYTES * ntasks_mpi, 0);
		if (ok && kry_mail_export(handle, mine.data()) != 0) ok = 0;
		MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm3d);
		if (all_ok) {
			MPI_Allgather(mine.data(), KRY_MAIL_HANDLE_BYTES, MPI_BYTE, all.data(), KRY_MAIL_HANDLE_BYTES, MPI_BYTE, comm3d);
			ok = kry_mail_connect(handle, rank, ntasks_mpi, all.data()) == 0 ? 1 : 0;
			MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, comm3d);
			if (ok) kry_mail_enable(handle, all_ok ? (KRY_MAIL_SUMS | KRY_MAIL_PLANES) : 0);
		}
		device_link = all_ok != 0;
	}
#endif
}

void BICGStab::finish(int st)
{
	kry_stats s;
	kry_get_stats(handle, &s);
	n_iter = s.iterations;
	last_status = st;
	fused = s.fused != 0;
	staged_fields = s.staged_fields;
	norm_final = s.norm_final;
	norm_rhs = s.norm_rhs;
	solve_ms = s.solve_ms;
	stage_ms = s.stage_ms;
	if (st == KRY_NAN_ALPHA) {
		cout << " alpha is nan " << endl;
		exit(1);
	}
	if (st == KRY_NAN_NORM) {
		cout << " Norm is nan at iteration " << n_iter << endl;
		exit(3);
	}
	if (st < 0) {
		cerr << " BICGStab: " << kry_status_string(st) << " (" << kry_last_error() << ")" << endl;
		exit(2);
	}
	if (verbose) cout << " BiCGStab(" << LValue << "): " << n_iter << " passes, " << solve_ms << " ms on the device" << endl;
}

std::vector<double> BICGStab::residualHistory() const
{
	std::vector<double> h;
	if (!handle) return h;
	int n = kry_get_history(handle, NULL, 0);
	if (n <= 0) return h;
	h.resize(n);
	kry_get_history(handle, h.data(), n);
	return h;
}

std::vector<double> BICGStab::step_times() const
{
	std::vector<double> t;
	if (!handle) return t;
	int n = kry_get_step_times(handle, NULL, 0);
	if (n <= 0) return t;
	t.resize(n);
	kry_get_step_times(handle, t.data(), n);
	return t;
}

void BICGStab::solve(BoundaryHandler3D &bounds, NumMatrix<double, 3> &phi, NumMatrix<double, 3> &rhs,
                     NumMatrix<double, 3> &lambda, double D_xx, double D_yy, double D_zz, int debug_, double, bool)
{
	// BICGStab3D/solveLin_BICGStab.C:200-215
	debug = debug_;
	use_spatialDiffusion = false;
	DiffDiag[0] = D_xx; DiffDiag[1] = D_yy; DiffDiag[2] = D_zz;
	solve_int(bounds, phi, rhs, lambda, rhs, rhs, rhs, rhs, debug_);
}

void BICGStab::solve(BoundaryHandler3D &bounds, NumMatrix<double, 3> &phi, NumMatrix<double, 3> &rhs,
                     NumMatrix<double, 3> &lambda, NumMatrix<double, 3> &Dxx, NumMatrix<double, 3> &Dyy,
                     NumMatrix<double, 3> &Dzz, NumMatrix<double, 3> &Dxy, int debug_, bool use_offDiagDiffusion_, double,
                     bool)
{
	// BICGStab3D/solveLin_BICGStab.C:218-236: Dxy is cleared when it is not used
	use_spatialDiffusion = true;
	debug = debug_;
	use_offDiagDiffusion = use_offDiagDiffusion_;
	if (!use_offDiagDiffusion) Dxy.clear();
	solve_int(bounds, phi, rhs, lambda, Dxx, Dyy, Dzz, Dxy, debug_);
}

void BICGStab::solve(BoundaryHandler3D &bounds, NumMatrix<double, 3> &phi, NumMatrix<double, 3> &rhs,
                     NumMatrix<double, 3> &lambda, NumMatrix<double, 3> &Dxx, NumMatrix<double, 3> &Dyy,
                     NumMatrix<double, 3> &Dzz, int debug_, double, bool)
{
	// The reference forwards Dzz in the place of Dxy, which the four-field
	// overload then clears (BICGStab3D/solveLin_BICGStab.C:238-248, 231-233): the
	// caller's Dzz is zeroed before the solve.  Here the off-diagonal element is
	// simply absent and Dzz is left intact, unless the reference's behaviour was asked for.
	if (three_field_quirk) {
		solve(bounds, phi, rhs, lambda, Dxx, Dyy, Dzz, Dzz, debug_, false);
		return;
	}
	use_spatialDiffusion = true;
	debug = debug_;
	use_offDiagDiffusion = false;
	NumMatrix<double, 3> none;
	solve_int(bounds, phi, rhs, lambda, Dxx, Dyy, Dzz, none, debug_);
}

unsigned long long BICGStab::hash(const NumMatrix<double, 3> &m)
{
	// BICGStab3D/solveLin_BICGStab.C:13-47
	unsigned long long h = 797003437ULL;
	const double *p = m.data();
	for (size_t q = 0; q < m.size(); ++q) {
		unsigned long long bits;
		memcpy(&bits, p + q, sizeof bits);
		h ^= bits;
	}
	return h;
}

void BICGStab::solve_int(BoundaryHandler3D &bounds, NumMatrix<double, 3> &phi, NumMatrix<double, 3> &rhs,
                         NumMatrix<double, 3> &lambda, NumMatrix<double, 3> &Dxx, NumMatrix<double, 3> &Dyy,
                         NumMatrix<double, 3> &Dzz, NumMatrix<double, 3> &Dxy, int debug_)
{
	// BICGStab3D/solveLin_BICGStab.C:250-527 runs on the device
	ensure_handle(bounds, phi);      // (a new handle forgets what was resident: release())
	int st;
	const double *dxy = (use_spatialDiffusion && use_offDiagDiffusion && Dxy.size() > 0) ? Dxy.data() : NULL;
	const int mode = use_spatialDiffusion ? (dxy ? 2 : 1) : 0;
	const void *coef[5] = {lambda.data(), use_spatialDiffusion ? Dxx.data() : NULL, use_spatialDiffusion ? Dyy.data() : NULL,
	                       use_spatialDiffusion ? Dzz.data() : NULL, dxy};
	bool same = keep_coefficients && last_mode >= 0 && mode == last_mode;
	for (int k = 0; k < 5; ++k) same = same && coef[k] == last_coef[k];
	if (mode == 0) for (int k = 0; k < 3; ++k) same = same && DiffDiag[k] == last_diag[k];
	for (int k = 0; k < 5; ++k) last_coef[k] = coef[k];
	for (int k = 0; k < 3; ++k) last_diag[k] = DiffDiag[k];
	last_mode = keep_coefficients ? mode : -1;   // arrays may change freely while the switch is off
	if (hash_trace) {
		// the inputs as the reference would print them (:284-287), then pass by pass (:472-477: its line converts
		// the 64-bit value to double and prints 6 digits; the exact integers follow on a line of their own)
		if (rank == 0) {
			cout << "Input variables:" << endl;
			cout << "\thash(phi)     = " << hash(phi) << endl;
			cout << "\thash(rhs)     = " << hash(rhs) << endl;
			cout << "\thash(lambda)  = " << hash(lambda) << endl;
		}
		st = use_spatialDiffusion
		         ? kry_load_var(handle, phi.data(), rhs.data(), lambda.data(), Dxx.data(), Dyy.data(), Dzz.data(), dxy, dxy != NULL)
		         : kry_load_const(handle, phi.data(), rhs.data(), lambda.data(), DiffDiag[0], DiffDiag[1], DiffDiag[2]);
		NumMatrix<double, 3> res = phi;
		kry_stats ks;
		int before = -1;
		while (st == KRY_OK) {
			st = kry_iterate(handle, 1);
			kry_get_stats(handle, &ks);
			if (st < 0 || ks.iterations == before) break;      // converged (or stopped) before this call: no pass ran
			before = ks.iterations;
			if (kry_fetch_phi(handle, phi.data()) < 0 || kry_fetch_residual(handle, res.data()) < 0) { st = KRY_ERR_HIP; break; }
			if (rank == 0) {
				const unsigned long long hp = hash(phi), hr = hash(res);
				cout << " hash(phi) = " << (double)hp << ", hash(residual) " << (double)hr << endl;
				cout << " [exact, pass " << ks.iterations << "] hash(phi) = " << hp << ", hash(residual) = " << hr << endl;
			}
			if (ks.norm_final <= eps * ks.norm_rhs) break;     // :496
		}
		if (st >= 0) {
			int st2 = kry_fetch_phi(handle, phi.data());
			if (st2 < 0) st = st2;
		}
		last_mode = -1;
	} else if (same) {
		// the operator is on the device already: stage phi and rhs, re-run the prologue, solve
		st = kry_update_phi(handle, phi.data());
		if (st >= 0) st = kry_update_rhs(handle, rhs.data());
		if (st >= 0) st = kry_resolve(handle, phi.data(), debug_);
	} else if (use_spatialDiffusion) {
		st = kry_solve_var(handle, phi.data(), rhs.data(), lambda.data(), Dxx.data(), Dyy.data(), Dzz.data(), dxy,
		                   dxy != NULL, debug_);
	} else {
		st = kry_solve_const(handle, phi.data(), rhs.data(), lambda.data(), DiffDiag[0], DiffDiag[1], DiffDiag[2],
		                     debug_);
	}
	finish(st);
}

void BICGStab::set_Advection(NumMatrix<double, 3> &, BoundaryHandler3D &, int)
{
	// BICGStab3D/Krylov.C:544-549
	cerr << " Advection not implemented yet for Krylov solvers -- Exiting " << endl;
	exit(3);
}
