#include "BiCGStabCL.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "../../include/krylov_amd.h"

#ifdef parallel
BiCGStabSolver::BiCGStabSolver(grid_manager &grid, double tolerance, int lValue, mpi_manager_3D &MyMPI, int hip_device)
    : BICGStab(grid, tolerance, lValue, MyMPI), initialized(false)
#else
BiCGStabSolver::BiCGStabSolver(grid_manager &grid, double tolerance, int lValue, int hip_device)
    : BICGStab(grid, tolerance, lValue), initialized(false)
#endif
{
	set_Devices(std::vector<int>(1, hip_device));
}

BiCGStabSolver::~BiCGStabSolver() {}

void BiCGStabSolver::setupContext(void)
{
	// BiCGStab3D_CL/BiCGStabCL.cpp: the reference compiles its kernels here and throws
	// when the device is unusable; the same failure is reported here
	if (kry_device_check(device_list()[0]) != 0) {
		std::cerr << " BiCGStabSolver: " << kry_last_error() << std::endl;
		exit(2);
	}
	initialized = true;
}

void BiCGStabSolver::cleanupContext(void)
{
	drop_handle();
	initialized = false;
}

bool BiCGStabSolver::isInitialized(void) { return initialized; }

long BiCGStabSolver::iterations(void) { return BICGStab::iterations(); }

std::vector<double> BiCGStabSolver::stepRuntimesMs(void) { return step_times(); }

std::vector<long> BiCGStabSolver::stepRuntimes(void)
{
	std::vector<double> t = step_times();
	std::vector<long> r(t.size());
	for (size_t i = 0; i < t.size(); ++i) r[i] = std::lround(t[i]);
	return r;
}

long BiCGStabSolver::steptimeMin(void)
{
	std::vector<double> t = step_times();
	return t.empty() ? 0 : std::lround(*std::min_element(t.begin(), t.end()));
}

long BiCGStabSolver::steptimeMax(void)
{
	std::vector<double> t = step_times();
	return t.empty() ? 0 : std::lround(*std::max_element(t.begin(), t.end()));
}
