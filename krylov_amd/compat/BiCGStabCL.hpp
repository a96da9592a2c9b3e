// BiCGStabCL.hpp -- the construction style and accessors of the reference's
// OpenCL solver class (BiCGStab3D_CL/BiCGStabCL.hpp:126-143, 196-200), on the
// MI355X solver.
//
//   BiCGStabSolver solver(grid, tolerance, lValue, device);   // a HIP device ordinal
//   solver.setVerbose(true); solver.setupContext();           // in place of a flexCL::Context*
//   solver.solve(bounds, phi, rhs, lambda, Dxx, Dyy, Dzz, Dxy, debug, useOffDiag);
//   solver.iterations(); solver.steptimeMin(); solver.steptimeMax(); solver.stepRuntimes();
//
// (BiCGStab3D_CL/main.cpp:410-412, 532-540 is the calling sequence.)  Times are
// milliseconds as in the reference; the fractional values are available too.
#ifndef KRYLOV_AMD_BICGSTABCL_HPP
#define KRYLOV_AMD_BICGSTABCL_HPP

#include <vector>

#include "solveLin_BICGStab.H"

class BiCGStabSolver : public BICGStab {
public:
#ifdef parallel
	BiCGStabSolver(grid_manager &grid, double tolerance, int lValue, mpi_manager_3D &MyMPI, int hip_device = 0);
#else
	BiCGStabSolver(grid_manager &grid, double tolerance, int lValue, int hip_device = 0);
#endif
	virtual ~BiCGStabSolver();

	void setupContext(void);      // checks that the device exists; the solver itself is set up on the first solve
	void cleanupContext(void);    // releases the device memory
	bool isInitialized(void);

	long iterations(void);
	std::vector<long> stepRuntimes(void);       // ms per outer iteration, rounded
	std::vector<double> stepRuntimesMs(void);   // the same, not rounded
	long steptimeMin(void);
	long steptimeMax(void);

private:
	bool initialized;
};

#endif
