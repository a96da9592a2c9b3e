#include "grid_manager.H"

#include <cmath>

// ---- grid_manager ----------------------------------------------------------

void grid_manager::init(int dim, const double *xb, const double *xe, const int *N, int rim)
{
	dim_ = dim;
	rim_ = rim;
	for (int d = 0; d < 3; ++d) { mx_[d] = 0; xb_[d] = xe_[d] = len_[d] = 0.; delx_[d] = 1.; }
	for (int b = 0; b < 6; ++b) bc_[b] = 1;           // default face type (grid_manager.C:216-218)
	for (int d = 0; d < dim; ++d) {
		mx_[d] = N[d] - 1;                            // grid_manager.C:203-205
		xb_[d] = xb[d];
		xe_[d] = xe[d];
		len_[d] = xe_[d] - xb_[d];                    // grid_manager.C:372
		delx_[d] = len_[d] / (1. * mx_[d]);           // grid_manager.C:373
		centres_[d].resize(Index::set(-rim), Index::set(mx_[d] + rim));
		for (int i = -rim; i <= mx_[d] + rim; ++i) centres_[d](i) = xb_[d] + delx_[d] * i;   // :306-311
	}
}

grid_manager::grid_manager(double xb, double yb, double zb, double xe, double ye, double ze, int Nx, int Ny, int Nz,
                           int rim)
{
	const double b[3] = {xb, yb, zb}, e[3] = {xe, ye, ze};
	const int n[3] = {Nx, Ny, Nz};
	init(3, b, e, n, rim);
}

grid_manager::grid_manager(double xb, double yb, double xe, double ye, int Nx, int Ny, int rim)
{
	const double b[3] = {xb, yb, 0.}, e[3] = {xe, ye, 0.};
	const int n[3] = {Nx, Ny, 1};
	init(2, b, e, n, rim);
}

grid_manager::grid_manager(double xb, double xe, int Nx, int rim)
{
	const double b[3] = {xb, 0., 0.}, e[3] = {xe, 0., 0.};
	const int n[3] = {Nx, 1, 1};
	init(1, b, e, n, rim);
}

grid_manager::grid_manager(const grid_manager &gm) { *this = gm; }

grid_manager &grid_manager::operator=(const grid_manager &gm)
{
	dim_ = gm.dim_;
	rim_ = gm.rim_;
	for (int d = 0; d < 3; ++d) {
		mx_[d] = gm.mx_[d]; xb_[d] = gm.xb_[d]; xe_[d] = gm.xe_[d]; len_[d] = gm.len_[d]; delx_[d] = gm.delx_[d];
		centres_[d] = gm.centres_[d];
	}
	for (int b = 0; b < 6; ++b) bc_[b] = gm.bc_[b];   // all 2*DIM faces
	return *this;
}

void grid_manager::set_bcType(int boundary, int type)
{
	assert(boundary >= 0 && boundary < 2 * dim_);
	bc_[boundary] = type;
}
int grid_manager::get_bcType(int boundary) const { assert(boundary >= 0 && boundary < 2 * dim_); return bc_[boundary]; }
int grid_manager::get_mx(int dir) const { assert(dir >= 0 && dir < dim_); return mx_[dir]; }
double grid_manager::get_delx(int dir) const { assert(dir >= 0 && dir < dim_); return delx_[dir]; }
double grid_manager::get_xb(int dir) const { assert(dir >= 0 && dir < dim_); return xb_[dir]; }
double grid_manager::get_xe(int dir) const { assert(dir >= 0 && dir < dim_); return xe_[dir]; }
double grid_manager::get_Len(int dir) const { assert(dir >= 0 && dir < dim_); return len_[dir]; }
double grid_manager::get_Pos(int dir, int ix) const
{
	assert(dir >= 0 && dir < dim_ && ix >= -rim_ && ix <= mx_[dir] + rim_);
	return centres_[dir](ix);
}
NumMatrix<double, 1> grid_manager::get_Grid(int dir) const { assert(dir >= 0 && dir < dim_); return centres_[dir]; }

// ---- grid_1D ---------------------------------------------------------------

grid_1D::grid_1D() : set_(false), mx_(0), nx_(0), rim_(0), type_(-99), centred_(false), xb_(0.), xe_(0.), len_(0.), del_(0.) {}

grid_1D::grid_1D(double xb, double xe, int Nx, int rim, int type, bool centred)
    : set_(true), mx_(Nx - 1), nx_(Nx), rim_(rim), type_(type), centred_(centred), xb_(xb), xe_(xe)
{
	len_ = xe_ - xb_;
	del_ = centred_ ? len_ / (1. * (nx_ - 1)) : len_ / (1. * nx_);      // grid_manager.C:665-671
	edges_.resize(Index::set(-rim_), Index::set(mx_ + rim_ + 1));
	centres_.resize(Index::set(-rim_), Index::set(mx_ + rim_));
	widths_.resize(Index::set(-rim_), Index::set(mx_ + rim_));
	for (int i = -rim_; i <= mx_ + rim_ + 1; ++i) {
		if (type_ == 0) edges_(i) = centred_ ? xb_ + del_ * (i - 0.5) : xb_ + del_ * i;      // :715-722
		else edges_(i) = xb_ + del_ * (i + 2. * std::sin(2. * M_PI * i / (1. * nx_)));      // :729
	}
	for (int i = -rim_; i <= mx_ + rim_; ++i) {                                           // :735-738
		centres_(i) = 0.5 * (edges_(i) + edges_(i + 1));
		widths_(i) = edges_(i + 1) - edges_(i);
	}
}

double grid_1D::get_del() const { assert(set_ && type_ == 0); return widths_(0); }       // :758-761
double grid_1D::get_del(int ipos) const { assert(set_); return widths_(ipos); }
double grid_1D::get_xCen(int ix) const { assert(set_); return centres_(ix); }
double grid_1D::get_xL(int ix) const { assert(set_); return edges_(ix); }
