// Structures shared by the host solver and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#define KRY_LMAX 8          // largest supported l
#define KRY_HIST_CAP 65536  // residual-history entries kept on the device
#define KRY_NSUMS 8         // dot-product slots per reduction

// ---------------------------------------------------------------------------
// Device field layout.
//
// A field is the ghost-inclusive box [-1..mx+1] x [-1..my+1] x [-1..mz+1]
// (mz is slab-local).  Rows are padded: column c = i + 2 holds grid index i, so
// that i = 0 sits on an even column and, with the allocation shifted by 14
// doubles, on a 128-byte boundary.  px (row pitch) is a multiple of 16 doubles,
// ps = px*(my+3) is the plane pitch.  Field pointers handed to kernels point at
// (c=0, j=-1, k=-1):  idx(i,j,k) = (k+1)*ps + (j+1)*px + (i+2).
// Guard zones of zeros surround every allocation, so tile overshoot and halo
// reads never leave it.
// ---------------------------------------------------------------------------
struct KryGeo {
	int mx, my, mz;     // last grid-point index per direction (slab-local in z)
	int px;             // row pitch in doubles
	long ps;            // plane pitch in doubles
	long n;             // ps * (mz + 3)
	// position of this slab in the global domain (z slabs)
	int has_lo;         // 1: plane -1 is a neighbour's plane (not a domain face)
	int has_hi;         // 1: plane mz+1 is a neighbour's plane
	int nb[6];          // per face xlo,xhi,ylo,yhi,zlo,zhi: 1 = a neighbouring sub-domain lies behind it
	                    // (nb[4] == has_lo, nb[5] == has_hi).  Interface planes are held by both
	                    // sides; in dot products they count once: z and y interfaces on the lower
	                    // side, x interfaces on the upper side (which also evaluates the operator
	                    // there; the lower side receives plane mx with the halo exchange).
	int bc[6];          // face types of the GLOBAL domain
};

__host__ __device__ inline long kry_idx(const KryGeo &g, int i, int j, int k)
{
	return (long)(k + 1) * g.ps + (long)(j + 1) * g.px + (i + 2);
}

// Constant / precomputed operator coefficients.
struct KryOp {
	double c[3];        // const: D_d/delx_d^2 ; var: 1/delx_d^2       (Krylov.C:511-514 / 422-425)
	double cd2;         // const: 2*((c0+c1)+c2)                          (Krylov.C:530)
	double cxy;         // var: 1/(2 dx dy)                               (Krylov.C:427)
	double two_d[3];    // 2*delx_d                                       (Krylov.C:449-456)
	double inv_two_d[3];// 1/(2*delx_d), used only when exact (power of two)
	int exact_rcp;      // 1: every 2*delx_d is a power of two, x/(2d) == x*(1/(2d)) bit for bit
};

// Mailbox all-reduce between the processes of a split domain (one GPU each).
// Every rank owns a mailbox in uncached device memory that all peers have mapped
// (HIP IPC).  The block that finishes a grid reduction stores its sums into slot
// [parity][own rank] of every rank's mailbox, followed by the sequence number of
// the reduction, waits until all slots of its own mailbox carry that number and
// adds them in rank order: every rank obtains the same bits, with no further
// kernel launch.  Two parities suffice, since a rank can be at most one reduction
// ahead of a peer (it needs the peer's message to finish the current one).
#define KRY_MAIL_MAX 64       // ranks
#define KRY_MAIL_SLOT 16      // doubles per message: KRY_NSUMS values, sequence word, padding
struct KryMail {
	int rank, nranks;
	long long timeout;        // wall_clock64 ticks a rank waits for a peer before it gives up
	double *box;              // own mailbox: KRY_MAIL_SLOT header doubles (word 0: sequence counter),
	                          // then [2][nranks][KRY_MAIL_SLOT]
	double *peer[KRY_MAIL_MAX];   // every rank's mailbox as mapped into this process (peer[rank] == box)
};

// Scalar state of the BiCGStab(l) recurrences; lives in device memory so that an
// outer iteration needs no host round trip (solveLin_BICGStab.C:317-324).
struct KryState {
	double rho0, rho1, alpha, beta, omega;
	double alpha0;               // alpha of jj = 0, kept for the phi update that k_final<2> carries
	double sigma[KRY_LMAX + 1], gammap[KRY_LMAX + 1], gamma[KRY_LMAX + 1], gammapp[KRY_LMAX + 1];
	double tau[KRY_LMAX + 1][KRY_LMAX + 1];
	double norm, norm_rhs, norm_init, eps;
	double sums[KRY_NSUMS];      // reduced dot products of the last reducing kernel
	int iter;                    // outer iterations completed
	int max_outer;               // 0 = unlimited
	int done;                    // 1: converged / failed / capped; later kernels exit at once
	int status;                  // KRY_OK, KRY_NAN_ALPHA, KRY_NAN_NORM, KRY_MAX_ITER
	int L;
	int local_scalars;           // 1: the reducing kernel also advances the recurrences
	int pad_[2];
	const KryMail *mail;         // not null: sums are all-reduced over the ranks' mailboxes by the reducing kernel
	long long wait_sum_ticks;    // wall_clock64 ticks the reducing kernels spent in the mailbox all-reduce since the load
	long long wait_halo_ticks;   // ... and the ghost-plane kernels waiting for the neighbours' planes
	double hist[KRY_HIST_CAP];
};

// Phases of the scalar recurrences (which dots just became available).
enum KryPhase {
	PH_NONE = 0,
	PH_INIT,      // sums[0] = |r0|^2 of the first residual
	PH_A,         // sums[0] = (u[jj+1], r~)                     -> alpha
	PH_B,         // sums[0] = (r[jj+1], r~)                     -> beta for the next jj
	PH_BL1,       // l=1 last B: sums = (r0,r1), (r1,r1)          -> omega
	PH_BL2,       // l=2 last B: sums = (r1,r1), (r2,r1)          -> sigma1, tau12
	PH_M2,        // l=2 MR:     sums = (r2,r2), (r0,r2), (r0,r1) -> gammap1, gammas, omega
	PH_F,         // sums = |r0|^2, (r0, r~)                      -> norm, convergence, next rho/beta
	// general l (the index jj, and ii for PH_TAU, ride in the phase word: see KRY_PHASE)
	PH_BN,        // last B kernel of the BiCG part: only the alpha check, its dot is not used
	PH_TAU,       // sums[0] = (r_jj, r_ii)                       -> tau[ii][jj]
	PH_SIG        // sums = (r_jj, r_jj), (r0, r_jj)              -> sigma[jj], gammap[jj]; jj = l: omega, gamma, gamma''
};
#define KRY_PHASE(ph, jj, ii) ((ph) | ((jj) << 8) | ((ii) << 16))
