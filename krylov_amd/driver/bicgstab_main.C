// bicgstab_main.C -- command-line driver for the MI355X BiCGStab(l) solver.
//
// Same job, options and printed results as the reference's drivers
// (BICGStab3D/main.C:57-581 and, for --stats / -i / -p, BiCGStab3D_CL/main.cpp:
// 102-144, 532-608): set up one of five manufactured problems with the analytic
// solution sin(pi x) sin(pi y) sin(pi z) on [0,1]^3, solve it through the
// Linsolver3D interface and report the l2 error against the analytic solution.
//
//   bicgstab [-n MX] [-t 1..5] [-r] [-l L] [-e TOL] [-i MAXITER] [--stats] [-q]
//
// The spellings of the OpenCL driver are understood as well (--np, -p/--precision, --random,
// -s/--stats/--statistical, -v/--verbose, and --gpu/--cpu/--acc, which choose an OpenCL context
// there and are accepted without effect here: the device is a HIP GPU, KRY_DEVICES picks it),
// so that its sweep script (BiCGStab3D_CL/statistics.sh: `--gpu --stats -t T -n N` over a list
// of sizes, rows appended to one file) runs against this binary; --stats prints the same
// tab-separated row under the same header (BiCGStab3D_CL/main.cpp:576, 599-602).
//
// Built with -D parallel (and an MPI installation) it is the reference's parallel
// driver: `mpiexec -n P bicgstab_mpi ...` splits the domain over P processes, one GPU
// each (KRY_DEVICES lists the devices, default all on 0): 2 x 2 x 2 for P = 8 as the
// reference hard-codes (BICGStab3D/main.C:145-150), P z slabs otherwise, or
// `--nproc NX NY NZ`.
//
// Built with -D WITH_HDF5 (bicgstab_h5) it also writes and reads field files in the
// format of the reference's Hdf5Stream classes (BICGStab3D/Hdf5File_cbase.H:18-144):
// `-o FILE` stores phi, phi_ana and rhs after the solve, `--restart FILE` takes the
// initial guess from the phi of such a file.  Under MPI every process uses FILE.<rank>.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../compat/BoundaryHandler.H"
#include "../compat/LinSolver3D.hpp"
#include "../compat/grid_manager.H"
#include "../compat/solveLin_BICGStab.H"
#ifdef WITH_HDF5
#include "../compat/Hdf5File_cbase.H"
#endif

using namespace std;

namespace {

const double default_tolerance = 1e-6;      // BICGStab3D/main.C:16

double randomf(double lo, double hi)
{
	// BICGStab3D/main.C:47-50
	const double diff = hi - lo;
	return lo + (double)rand() / (double)(RAND_MAX / diff);
}

struct Options {
	int mx = 32;            // BICGStab3D/main.C:71
	int test = 1;           // 1..5
	bool randomize = false;
	int L = 2;              // BICGStab3D/main.C:458
	double tol = default_tolerance;
	int max_iter = 0;
	bool stats = false;
	bool hashes = false;    // the reference's disabled hash prints (solveLin_BICGStab.C:284-287, 472-477)
	bool quiet = false;
	unsigned seed = 12345;
	int nproc[3] = {0, 0, 0};   // -D parallel: process grid (0 = choose from the process count)
	string output, restart;     // field files (WITH_HDF5)
};

void usage(const char *argv0)
{
	cout << "BiCGStab solver (MI355X)" << endl;
	cout << "Usage: " << argv0 << " [OPTIONS]" << endl;
	cout << "OPTIONS:" << endl;
	cout << "    -h   --help                   Print program help" << endl;
	cout << "    -n   --size MX                " << endl;
	cout << "         --mx   MX                Define problem size" << endl;
	cout << "    -t   --test TEST              Define test case (1-5)" << endl;
	cout << "    -r   --randomize              Randomize input" << endl;
	cout << "    -l   --lvalue L               BiCGStab(L), default 2" << endl;
	cout << "    -e   --tolerance TOL          Relative tolerance, default 1e-6" << endl;
	cout << "    -i   --iterations N           Stop after N outer iterations" << endl;
	cout << "    -s   --seed SEED              Seed for --randomize" << endl;
	cout << "         --stats                  Print one tab-separated row instead of the report" << endl;
	cout << "    -q   --quiet                  No per-iteration output" << endl;
	cout << "         --hashes                 Print the reference's hash() of the inputs and, after every pass, of phi" << endl;
	cout << "                                  and the residual (its disabled prints, solveLin_BICGStab.C:284-287, 472-477)" << endl;
#ifdef WITH_HDF5
	cout << "    -o   --output FILE            Write phi, phi_ana and rhs to an HDF5 field file" << endl;
	cout << "         --restart FILE           Initial guess: phi of an HDF5 field file" << endl;
#endif
}

bool parse(int argc, char **argv, Options &o)
{
	for (int i = 1; i < argc; ++i) {
		const string a = argv[i];
		auto need = [&](const char *what) -> const char * {
			if (i >= argc - 1) { cerr << "Missing argument: " << what << endl; exit(EXIT_FAILURE); }
			return argv[++i];
		};
		if (a == "-h" || a == "--help") { usage(argv[0]); exit(EXIT_SUCCESS); }
		else if (a == "-n" || a == "--size" || a == "--mx" || a == "--np") o.mx = atoi(need("Size"));
		else if (a == "-r" || a == "--randomize" || a == "--random") o.randomize = true;
		else if (a == "--gpu" || a == "--cpu" || a == "--acc" || a == "--accelerator") {}   // OpenCL context choices
		else if (a == "-v" || a == "--verbose") o.quiet = false;
		else if (a == "-p" || a == "--precision") o.tol = atof(need("Tolerance"));
		else if (a == "--statistical") o.stats = true;
		// -s: the seed when a number follows, else the OpenCL driver's short form of --stats
		else if (a == "-s" && (i >= argc - 1 || argv[i + 1][0] == '-')) o.stats = true;
		else if (a == "-t" || a == "--test") {
			o.test = atoi(need("Test"));
			if (o.test < 1 || o.test > 5) { cerr << "Illegal test number. Must be [1-5] " << endl; return false; }
		} else if (a == "-l" || a == "--lvalue") o.L = atoi(need("L"));
		else if (a == "-e" || a == "--tolerance") o.tol = atof(need("Tolerance"));
		else if (a == "-i" || a == "--iterations") o.max_iter = atoi(need("Iterations"));
		else if (a == "-s" || a == "--seed") o.seed = (unsigned)atol(need("Seed"));
		else if (a == "--nproc") { for (int d = 0; d < 3; ++d) o.nproc[d] = atoi(need("Process grid")); }
		else if (a == "--stats") o.stats = true;
		else if (a == "--hashes") o.hashes = true;
		else if (a == "-q" || a == "--quiet") o.quiet = true;
#ifdef WITH_HDF5
		else if (a == "-o" || a == "--output") o.output = need("Output file");
		else if (a == "--restart") o.restart = need("Restart file");
#endif
		else if (!a.empty() && a[0] != '-') o.mx = atoi(a.c_str());
		else { cerr << "Illegal argument: " << a << endl; return false; }
	}
	return true;
}

} // namespace

int main(int argc, char *argv[])
{
#ifdef parallel
	int ntasks = 1, rank = 0;                 // BICGStab3D/main.C:58-63
	MPI_Init(&argc, &argv);
	MPI_Comm_size(MPI_COMM_WORLD, &ntasks);
	MPI_Comm_rank(MPI_COMM_WORLD, &rank);
#else
	const int rank = 0;
#endif
	Options opt;
	if (!parse(argc, argv, opt)) return EXIT_FAILURE;
	if (rank != 0) opt.quiet = true;
	const bool talk = !opt.stats && rank == 0;
	if (talk) cout << "Problem size: " << opt.mx << ", running test " << opt.test << endl;

	const int N = opt.mx + 1;
	grid_manager MyGrid(0., 0., 0., 1., 1., 1., N, N, N, 1);
	grid_1D grid_x(0, 1, N, 2, 0, true), grid_y(0, 1, N, 2, 0, true), grid_z(0, 1, N, 2, 0, true);
	for (int b = 0; b < 6; ++b) MyGrid.set_bcType(b, 0);

#ifdef parallel
	// BICGStab3D/main.C:145-186
	NumArray<int> nproc(3), mx_global(3);
	if (opt.nproc[0] > 0) { nproc[0] = opt.nproc[0]; nproc[1] = opt.nproc[1]; nproc[2] = opt.nproc[2]; }
	else if (ntasks == 8) { nproc[0] = nproc[1] = nproc[2] = 2; }
	else { nproc[0] = 1; nproc[1] = 1; nproc[2] = ntasks; }
	mx_global[0] = mx_global[1] = mx_global[2] = opt.mx;
	mpi_manager_3D MyMPI(nproc, mx_global);
	grid_manager LocalGrid = MyMPI.make_LocalGrid(MyGrid);
	grid_1D grid_x_local = MyMPI.make_LocalGrid1D(grid_x, 0);
	grid_1D grid_y_local = MyMPI.make_LocalGrid1D(grid_y, 1);
	grid_1D grid_z_local = MyMPI.make_LocalGrid1D(grid_z, 2);
	BoundaryHandler3D MyBounds(MyMPI);
	grid_manager &PosGrid = LocalGrid;
	const int m = LocalGrid.get_mx(0), my_ = LocalGrid.get_mx(1), mz_ = LocalGrid.get_mx(2);
#else
	BoundaryHandler3D MyBounds;
	grid_manager &PosGrid = MyGrid;
	const int m = opt.mx, my_ = opt.mx, mz_ = opt.mx;
#endif
	MyBounds.set_bcType(0);                   // all faces Dirichlet (BICGStab3D/main.C:187)

	NumMatrix<double, 3> phi, phi_ana, rhs, lambda, D[4];
	const Index lo = Index::set(-1, -1, -1), hi = Index::set(m + 1, my_ + 1, mz_ + 1);
	phi.resize(lo, hi); phi_ana.resize(lo, hi); rhs.resize(lo, hi); lambda.resize(lo, hi);
	for (int k = 0; k < 4; ++k) D[k].resize(lo, hi);
	D[0].set_constVal(1.); D[1].set_constVal(1.); D[2].set_constVal(1.); D[3].clear();

	const double pi = M_PI;
	double Diff[3] = {1., 1., 10.};           // BICGStab3D/main.C:223-229
	double lambda_factor = 0.2, dfac[4] = {1., 1., 1., 1.};
	if (opt.randomize) {                      // BICGStab3D/main.C:237-241
		srand(opt.seed);
		lambda_factor = randomf(0.1, 100.0);
		for (int k = 0; k < 4; ++k) dfac[k] = randomf(0.1, 10.0);
	}
	const double DPar = 1., DPerp = 0.1;

	for (int iz = 0; iz <= mz_; ++iz)
		for (int iy = 0; iy <= my_; ++iy)
			for (int ix = 0; ix <= m; ++ix) {
				const double x = PosGrid.get_Pos(0, ix), y = PosGrid.get_Pos(1, iy), z = PosGrid.get_Pos(2, iz);
				// faces of the global domain (the unit-tensor test zeroes its solution there)
				const bool top_face = (x >= 1. - 1e-12) || (y >= 1. - 1e-12) || (z >= 1. - 1e-12);
				const double sx = sin(pi * x), sy = sin(pi * y), sz = sin(pi * z);
				const double cx = cos(pi * x), cy = cos(pi * y), cz = cos(pi * z);
				const double ana = sx * sy * sz;
				phi_ana(ix, iy, iz) = ana;
				const double lam = lambda_factor * x * sqr(y) * z;
				lambda(ix, iy, iz) = lam;
				switch (opt.test) {
				case 1:   // constant diagonal tensor (main.C:282-284)
					rhs(ix, iy, iz) = -(sqr(pi) * (Diff[0] + Diff[1] + Diff[2]) + lam) * ana;
					break;
				case 2:   // Dxx = y, Dyy = x, Dzz = z (main.C:299-311)
					D[0](ix, iy, iz) = dfac[0] * y; D[1](ix, iy, iz) = dfac[1] * x; D[2](ix, iy, iz) = dfac[2] * z;
					rhs(ix, iy, iz) = -(sqr(pi) * (x + y + z) + lam) * ana + pi * sx * sy * cz;
					break;
				case 3: { // test 2 plus Dxy = A x^2 y z (main.C:344-357)
					const double A = 0.1;
					D[0](ix, iy, iz) = dfac[0] * y; D[1](ix, iy, iz) = dfac[1] * x; D[2](ix, iy, iz) = dfac[2] * z;
					const double dxy = dfac[3] * A * sqr(x) * y * z;
					D[3](ix, iy, iz) = dxy;
					rhs(ix, iy, iz) = -(sqr(pi) * (x + y + z) + lam) * ana + pi * sx * sy * cz +
					                  2. * dxy * sqr(pi) * cx * cy * sz + 2. * A * x * y * z * pi * sx * cy * sz +
					                  A * sqr(x) * z * pi * cx * sy * sz;
					break;
				}
				case 4: { // rotated anisotropic tensor (main.C:360-394).  As there, the right-hand side is 0/0 on
				          // the edge x = y = 0 (r2 below): its norm is NaN, the stopping test `norm > eps*normRHS`
				          // is false at once and the run ends after one pass with phi near zero.
					const double ang = atan2(y, x), sa = sin(ang), ca = cos(ang);
					const double dxx = dfac[0] * (DPar * sqr(sa) + DPerp * sqr(ca));
					const double dyy = dfac[1] * (DPar * sqr(ca) + DPerp * sqr(sa));
					const double dzz = dfac[2] * DPerp;
					const double dxy = dfac[3] * (DPerp - DPar) * sa * ca;
					D[0](ix, iy, iz) = dxx; D[1](ix, iy, iz) = dyy; D[2](ix, iy, iz) = dzz; D[3](ix, iy, iz) = dxy;
					const double r2 = sqr(x) + sqr(y);
					const double dadx = -y / r2, dady = x / r2;
					const double dDxxDx = 2. * (DPar - DPerp) * sa * ca * dadx;
					const double dDyyDy = 2. * (DPerp - DPar) * sa * ca * dady;
					const double dDxyDx = (DPerp - DPar) * (sqr(ca) - sqr(sa)) * dadx;
					const double dDxyDy = (DPerp - DPar) * (sqr(ca) - sqr(sa)) * dady;
					rhs(ix, iy, iz) = (dDyyDy + dDxyDx) * pi * sx * cy * sz + (dDxxDx + dDxyDy) * pi * cx * sy * sz +
					                  2. * dxy * sqr(pi) * cx * cy * sz - ((dxx + dyy + dzz) * sqr(pi) + lam) * ana;
					break;
				}
				default:  // test 5: unit tensor fields through the field interface (main.C:313-336)
					if (top_face) phi_ana(ix, iy, iz) = 0.;
					D[0](ix, iy, iz) = dfac[0]; D[1](ix, iy, iz) = dfac[1]; D[2](ix, iy, iz) = dfac[2]; D[3](ix, iy, iz) = 0.;
					rhs(ix, iy, iz) = -(sqr(pi) * (1. + 1. + 1.) + lam) * phi_ana(ix, iy, iz);
					break;
				}
			}

#ifdef WITH_HDF5
	const string suffix = [&] {
#ifdef parallel
		return "." + to_string(rank);
#else
		return string();
#endif
	}();
	if (!opt.restart.empty()) {
		Hdf5iStream in(opt.restart + suffix, talk ? 0 : 1);
		in.Read3DMatrix("phi", phi);
	}
#endif

	if (talk) cout << " bauen " << endl;
	Linsolver3D *lin_solver = new BICGStab();
	NumArray<int> pars(1);
	pars(0) = opt.L;
#ifdef parallel
	lin_solver->setup(grid_x_local, grid_y_local, grid_z_local, opt.tol, pars, false, false, MyMPI, opt.max_iter);
#else
	lin_solver->setup(grid_x, grid_y, grid_z, opt.tol, pars, false, false, opt.max_iter);
#endif

	if (opt.hashes) static_cast<BICGStab *>(lin_solver)->set_HashTrace(true);
	if (talk) cout << " lösen " << endl;
	const int dbg = (opt.quiet || opt.stats) ? 0 : 8;
	const auto t0 = chrono::steady_clock::now();
	if (opt.test == 1) lin_solver->solve(MyBounds, phi, rhs, lambda, Diff[0], Diff[1], Diff[2], dbg);
	else if (opt.test == 2 || opt.test == 5) lin_solver->solve(MyBounds, phi, rhs, lambda, D[0], D[1], D[2], D[3], dbg);
	else lin_solver->solve(MyBounds, phi, rhs, lambda, D[0], D[1], D[2], D[3], dbg, true);
	const double runtime = chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count();
	if (talk) cout << "Solver fertig in " << (long long)runtime << " ms" << endl;

	// l2 and max error against the analytic solution (main.C:503-567)
	double error = 0., num = 0., emax = 0.;
	for (int iz = 0; iz <= mz_; ++iz)
		for (int iy = 0; iy <= my_; ++iy)
			for (int ix = 0; ix <= m; ++ix) {
				const double d = phi_ana(ix, iy, iz) - phi(ix, iy, iz);
				error += sqr(d);
				num += 1.;
				emax = max(emax, fabs(d));
			}
#ifdef parallel
	// main.C:550-560 (interface planes are counted by both owners there as well)
	double g[2] = {error, num}, gm = emax;
	MPI_Allreduce(MPI_IN_PLACE, g, 2, MPI_DOUBLE, MPI_SUM, MyMPI.comm3d);
	MPI_Allreduce(MPI_IN_PLACE, &gm, 1, MPI_DOUBLE, MPI_MAX, MyMPI.comm3d);
	error = g[0]; num = g[1]; emax = gm;
#endif
	const double l2err = sqrt(error / num);

	BICGStab *bs = static_cast<BICGStab *>(lin_solver);
#ifdef WITH_HDF5
	if (!opt.output.empty()) {
		double xb[3], dx[3];
		for (int d = 0; d < 3; ++d) {
			dx[d] = PosGrid.get_delx(d);
			xb[d] = PosGrid.get_Pos(d, -1);       // position of the first stored element (the ghost layer)
		}
		Hdf5Stream out(opt.output + suffix, 3, rank);
		out.AddGlobalAttr("test", opt.test);
		out.AddGlobalAttr("mx", opt.mx);
		out.AddGlobalAttr("lvalue", opt.L);
		out.AddGlobalAttr("tolerance", opt.tol);
		out.AddGlobalAttr("iterations", bs->iterations());
		out.AddGlobalAttr("l2_error", l2err);
		out.Write3DMatrix("phi", phi, xb, dx);
		out.Write3DMatrix("phi_ana", phi_ana, xb, dx);
		out.Write3DMatrix("rhs", rhs, xb, dx);
		out.close();
	}
#endif
	if (rank != 0) {
		// only the first process reports
	} else if (opt.stats) {
		// columns of the reference's --stats row (BiCGStab3D_CL/main.cpp:576,600-602)
		const int it = bs->iterations();
		const double calc = bs->solveRuntime();
		const std::vector<double> steps = bs->step_times();
		double tmin = 0., tmax = 0., avg = it > 0 ? calc / it : 0., geo = avg;
		if (!steps.empty()) {
			tmin = tmax = steps[0];
			double sum = 0., logsum = 0.;
			for (size_t i = 0; i < steps.size(); ++i) {
				tmin = min(tmin, steps[i]); tmax = max(tmax, steps[i]);
				sum += steps[i]; logsum += log(max(steps[i], 1e-300));
			}
			avg = sum / steps.size();
			geo = exp(logsum / steps.size());
		}
		cout << "# Test\tSize\tIterations\tTotal runtime\tCalculation time\tMin step time\tMax step time\t"
		        "Average (arithmetic)\tAverage (geometric)\tTolerance\tL2 error\tMax error" << endl;
		const char sep = '\t';
		cout << opt.test << sep << opt.mx << sep << it << sep << runtime << sep << calc << sep << tmin << sep << tmax << sep
		     << avg << sep << geo << sep << opt.tol << sep << l2err << sep << emax << endl;
	} else {
		cout << " l2 error for " << opt.mx << " is " << l2err << endl;
		cout << "Runtime (solver) : " << (long long)runtime << " ms" << endl;
		cout << "Iterations       : " << bs->iterations() << " (" << bs->solveRuntime() << " ms on the device, "
		     << (bs->usedFusedKernels() ? "fused" : "generic") << " kernels)" << endl;
#ifdef parallel
		cout << "Exchange         : " << (bs->usesDeviceLink() ? "device memory mapped between the processes" : "MPI") << endl;
#endif
	}
	delete lin_solver;
#ifdef parallel
	MPI_Finalize();                           // main.C:577-579
#endif
	return EXIT_SUCCESS;
}
