/*
 * krylov_amd.h -- C ABI of the MI355X BiCGStab(l) stencil solver.
 *
 * Plain C types only.  Every entry point names the reference interface it
 * stands in for (paths relative to the reference tree).
 *
 * Host field layout (what the reference's NumMatrix<double,3> fields hold):
 * ghost-inclusive box [-1 .. m+1] per direction, x fastest, no padding:
 *   index(i,j,k) = ((k+1)*(my+3) + (j+1))*(mx+3) + (i+1).
 * The library copies the fields into its own padded device layout and never
 * keeps a caller's pointer after a call returns.  Every field argument may be a
 * host pointer or a device pointer (same layout): a caller whose fields already
 * live on the GPU pays no PCIe transfer.
 *
 * Domain decomposition: sub-domains that share their interface planes, by default
 * z slabs (a Cartesian split in all three directions: kry_desc.nproc), as
 * mpi_manager_3D::make_LocalGrid does for direction 2
 * (BICGStab3D/mpi_manager.C:342-408): with `nslabs` slabs every slab has
 * mz_local = mz_global / nslabs and plane mz_local of slab s is plane 0 of
 * slab s+1.  A process owns `slab_count` consecutive slabs starting at
 * `slab_begin` (one per device in `devices`), and the host fields it passes
 * cover exactly those slabs: mz_host = slab_count * mz_local.
 */
#ifndef KRYLOV_AMD_H
#define KRYLOV_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define KRY_MAX_L        8
#define KRY_MAX_SLABS    16
#define KRY_UID_BYTES    128

/* status codes (the reference prints and calls exit(); see kry_solve_*) */
enum {
	KRY_OK            = 0,
	KRY_NAN_ALPHA     = 1,   /* BICGStab3D/solveLin_BICGStab.C:406-409, exit(1) */
	KRY_NAN_NORM      = 3,   /* BICGStab3D/solveLin_BICGStab.C:491-494, exit(3) */
	KRY_MAX_ITER      = 4,   /* stopped by max_outer (no counterpart: the reference has no cap) */
	KRY_ERR_ARG       = -1,
	KRY_ERR_HIP       = -2,
	KRY_ERR_COMM      = -3,
	KRY_ERR_STATE     = -4
};

/* flags */
enum {
	KRY_FORCE_GENERIC = 1,   /* run the one-kernel-per-primitive path even when the fused path applies */
	KRY_NO_GRAPH      = 2    /* never replay pairs of passes as a HIP graph (the default; KRY_GRAPH=1 in the
	                            environment turns the replay on for single-slab runs) */
};

typedef struct kry_desc {
	int    mx[3];          /* last grid-point index of the host fields, per direction
	                          (grid_1D::get_mx, BICGStab3D/solveLin_BICGStab.C:165-167); at least 4,
	                          per slab too (smaller grids: KRY_ERR_ARG) */
	double delx[3];        /* grid spacing (grid_1D::get_del, solveLin_BICGStab.C:161-163) */
	int    bc_type[6];     /* xlo,xhi,ylo,yhi,zlo,zhi of the GLOBAL domain: 0 Dirichlet,
	                          1 linear extrapolation ("Neumann"), other: untouched
	                          (BoundaryHandler::bc_Type, BICGStab3D/BoundaryHandler.C:49-91) */
	int    L;              /* solverPars(0), solveLin_BICGStab.C:150 */
	double eps;            /* epsilon_, solveLin_BICGStab.C:149 */
	int    max_outer;      /* 0 = unlimited */
	int    flags;
	int    nslabs;         /* global number of z slabs (>= 1) */
	int    slab_begin;     /* first slab owned by this process */
	int    slab_count;     /* slabs owned by this process (>= 1) */
	int    devices[KRY_MAX_SLABS]; /* HIP device ordinal per owned slab */
	/* Optional Cartesian split over processes, the layout of mpi_manager_3D
	 * (BICGStab3D/mpi_manager.C:56-305, 342-408): nproc sub-domains per direction, this
	 * process at coord, rank = (coord[0]*nproc[1] + coord[1])*nproc[2] + coord[2]
	 * (MPI_Cart_create without reordering).  All zero: plain z slabs as described by
	 * nslabs / slab_begin / slab_count.  Otherwise one sub-domain per process
	 * (slab_count = 1), mx[] are the LOCAL extents (global / nproc) and a communicator
	 * (kry_comm_init*) over nproc[0]*nproc[1]*nproc[2] ranks is required. */
	int    nproc[3];
	int    coord[3];
} kry_desc;

typedef struct kry_stats {
	int    status;
	int    iterations;       /* outer iterations executed by the last solve/iterate */
	int    fused;            /* 1: fused kernels, all faces Dirichlet; 2: fused kernels, general
	                            faces; 0: one kernel per primitive */
	double norm_rhs;
	double norm_init;
	double norm_final;
	double solve_ms;         /* device time of the iteration loop (HIP events) */
	double stage_ms;         /* host<->device staging + prologue */
	double bytes_per_iter;   /* algorithmic bytes: 8 * W * N_pts of the global grid */
	double kernel_ms[8];     /* HIP-event time per fused kernel kind, summed over calls */
	int    kernel_calls[8];
	int    tile_rows;        /* launch geometry, which fixes the summation order of the dots: */
	int    chunk_planes;     /*   rows per fused tile (negative: general-boundary tiles that  */
	int    stream_blocks;    /*   start at grid row 0 instead of -1), planes per fused chunk, */
	                         /*   grid target of the streaming kernels                        */
	double enqueue_ms;       /* host time spent queueing kernels and transfers of the passes  */
	/* Exchange between the sub-domains of a split, device time on this process's first slab since
	 * the last load: [0] the all-reduces of the dot products, [1] waiting for ghost planes.
	 * Through the transport they are timed by HIP events while per-kernel timing is on
	 * (kry_set_profile); through mapped device memory the kernels count their own waiting time. */
	double comm_wait_ms[2];
	int    comm_wait_calls[2];   /* timed transport calls (0 on the mapped-memory routes) */
	int    rccl_ranks;       /* ranks of the RCCL communicator (ncclCommCount), 0: none in use */
	int    staged_fields;    /* fields copied into the library for the last solve: 3 or 6-7 after a load /
	                            full solve, only the replaced ones after kry_update_* + kry_resolve */
	int    overlapped_exchanges; /* Cartesian split: stencil launches since the last load that ran their inner tiles
	                            while the ghost cells of their input were still being exchanged */
	int    slab_link;        /* 1: the slabs of this process exchange through each other's device memory
	                            (kry_slab_link_note), 0: by copies and events, or there is one slab */
} kry_stats;

typedef struct kry_solver *kry_handle;

/* Replaces BICGStab::BICGStab / BICGStab::setup + make_Arrays
 * (BICGStab3D/solveLin_BICGStab.C:104-191): fixes geometry, l and eps and
 * allocates the work vectors on the device(s). */
kry_handle kry_create(const kry_desc *desc, int *status);
void       kry_destroy(kry_handle h);

/* Inter-process communicator for slabs owned by other processes; the analogue
 * of handing mpi_manager_3D::comm3d to the solver
 * (BICGStab3D/solveLin_BICGStab.C:142-147).  Rank r must own slab r.  The id
 * comes from kry_comm_unique_id on one rank and is distributed by the caller
 * (MPI_Bcast, a torch.distributed store, ...). */
int kry_comm_unique_id(void *uid /* KRY_UID_BYTES */);
int kry_comm_init(kry_handle h, const void *uid, int rank, int nranks);

/* Exercises the RCCL transport on a single device: a one-rank communicator, an
 * in-place all-reduce and a grouped send/receive to itself, each checked against
 * the expected data.  Returns KRY_OK, or KRY_ERR_COMM with kry_last_error() set.
 * (No counterpart in the reference; a deployment check.) */
int kry_comm_selftest(int device);

/* KRY_OK when HIP device `device` exists and accepts work (a small allocation and
 * copy are tried), else KRY_ERR_HIP with kry_last_error() set.  Stands in for the
 * context set-up of the reference's OpenCL solver (BiCGStabSolver::setupContext,
 * BiCGStab3D_CL/BiCGStabCL.hpp:130-131), which fails there when the device is unusable. */
int kry_device_check(int device);

/* The same role with the caller's own transport instead of RCCL, e.g. the
 * MPI communicator of a `-D parallel` build (the reference exchanges faces with
 * MPI_Isend/MPI_Irecv, BICGStab3D/BoundaryHandler.C:1633-1677, and reduces with
 * MPI_Allreduce, BICGStab3D/Krylov.C:239-240).  Both callbacks get DEVICE
 * pointers, are called with the slab's stream idle, must complete before they
 * return and return 0 on success.
 *   allreduce: in-place sum of n doubles over all ranks;
 *   halo:      send `count` doubles to / receive `count` doubles from the lower
 *              and upper z neighbour (a peer < 0 means no neighbour). */
typedef int (*kry_allreduce_cb)(void *ctx, double *device_buf, int n);
typedef int (*kry_halo_cb)(void *ctx, int peer_lo, const double *send_lo, double *recv_lo, int peer_hi,
                           const double *send_hi, double *recv_hi, unsigned long count);
int kry_comm_init_callbacks(kry_handle h, int rank, int nranks, kry_allreduce_cb allreduce, kry_halo_cb halo,
                            void *ctx);

/* The same callbacks with HOST pointers, for a transport that cannot read device
 * memory (an MPI library without GPU support): the library stages the 8 sums and
 * the two interface planes through pinned host buffers around each call. */
int kry_comm_init_host_callbacks(kry_handle h, int rank, int nranks, kry_allreduce_cb allreduce, kry_halo_cb halo,
                                 void *ctx);

/* All-reduce of the dot products through device mailboxes instead of the transport's
 * all-reduce (the reference: MPI_Allreduce, BICGStab3D/Krylov.C:149-150, 239-240), for
 * processes whose GPUs can address each other (one node, xGMI or the same device).
 * The block that finishes a reduction kernel stores its sums into every rank's mailbox
 * and adds the ranks' contributions in rank order itself: no extra kernel launch per
 * dot product, and the same bits on every rank.  The transport set with kry_comm_init*
 * carries the ghost planes unless KRY_MAIL_PLANES is chosen too; with both chosen no
 * transport is needed at all (kry_mail_connect on a handle without one chooses both).
 *   kry_mail_export:  creates this rank's mailbox and landing slots, writes their HIP IPC handle
 *                     (KRY_MAIL_HANDLE_BYTES bytes) to `handle`;
 *   kry_mail_connect: `handles` = the nranks handles in rank order (all-gathered by the
 *                     caller); maps the peers' mailboxes and switches the path on;
 *   kry_mail_enable:  chooses what travels this way (between solves): KRY_MAIL_SUMS, the dot
 *                     products; KRY_MAIL_PLANES, also the ghost planes (the reference:
 *                     BICGStab3D/BoundaryHandler.C:1633-1677) -- the sender's kernel stores its
 *                     planes into landing slots of the neighbour's memory and raises a sequence
 *                     word, the receiver's kernel waits for the word and fills its ghost cells;
 *                     0 returns to the transport.  kry_mail_connect leaves KRY_MAIL_SUMS on.
 * A rank that waits longer than KRY_MAIL_TIMEOUT_MS (default 30000) for a peer stops
 * the solve with KRY_ERR_COMM.  At most 64 ranks. */
#define KRY_MAIL_HANDLE_BYTES 64
#define KRY_MAIL_SUMS   1
#define KRY_MAIL_PLANES 2
int kry_mail_export(kry_handle h, void *handle);
int kry_mail_connect(kry_handle h, int rank, int nranks, const void *handles);
int kry_mail_enable(kry_handle h, int what);

/* Replaces BICGStab::solve(bounds, phi, rhs, lambda, D_xx, D_yy, D_zz, debug)
 * (BICGStab3D/solveLin_BICGStab.C:200-215).  phi: initial guess in, solution
 * out.  Returns a status instead of calling exit(). */
int kry_solve_const(kry_handle h, double *phi, const double *rhs, const double *lambda,
                    double Dxx, double Dyy, double Dzz, int debug);

/* Replaces BICGStab::solve(bounds, phi, rhs, lambda, Dxx, Dyy, Dzz, Dxy, debug,
 * use_offDiagDiffusion) (BICGStab3D/solveLin_BICGStab.C:218-236).  With
 * use_offdiag == 0 the solver treats Dxy as zero; Dxy may then be NULL.  (The
 * reference clears the caller's Dxy array; the C++ adaptor does that.) */
int kry_solve_var(kry_handle h, double *phi, const double *rhs, const double *lambda,
                  const double *Dxx, const double *Dyy, const double *Dzz, const double *Dxy,
                  int use_offdiag, int debug);

/* Device-resident stepping (no counterpart in the reference, which always
 * solves to convergence): kry_load_* stages the fields and runs the prologue of
 * solve_int (solveLin_BICGStab.C:268-320); kry_iterate runs up to n_outer passes
 * of the do{}while body (solveLin_BICGStab.C:327-496) and stops early on
 * convergence or failure; kry_fetch_phi copies the current solution back: after a
 * finished solve (converged or out of passes) exactly what kry_solve_* returns --
 * for constant coefficients that includes the do_BCs the reference's post-loop
 * residual applies to phi (solveLin_BICGStab.C:512-518, Krylov.C:516) -- and
 * before that the untreated iterate. */
int kry_load_const(kry_handle h, const double *phi, const double *rhs, const double *lambda,
                   double Dxx, double Dyy, double Dzz);
int kry_load_var(kry_handle h, const double *phi, const double *rhs, const double *lambda,
                 const double *Dxx, const double *Dyy, const double *Dzz, const double *Dxy,
                 int use_offdiag);
int kry_iterate(kry_handle h, int n_outer);
int kry_fetch_phi(kry_handle h, double *phi);
/* The current residuals[0] of the loop (solveLin_BICGStab.C:294-299 before the first pass, :451-462
 * after a pass), in the host layout: what the reference's disabled per-pass print hashes next to phi
 * (solveLin_BICGStab.C:472-477).  For stepping drivers and cross-checks; the solve does not need it. */
int kry_fetch_residual(kry_handle h, double *r0);

/* Repeated solves with the coefficients resident on the device: what a time-stepping caller of
 * Linsolver3D::solve does next (BICGStab3D/LinSolver3D.hpp:31-46 carries delt / evolve_time for
 * it; BiCGStab3D_CL/BiCGStabCL.hpp:44 names the host code).  After a kry_load_* or kry_solve_*
 * the operator (D fields or constants, lambda) stays in HBM;
 *   kry_update_rhs / _phi / _lambda  replace one field (host or device pointer, host layout);
 *   kry_reset    re-runs the prologue of solve_int (solveLin_BICGStab.C:268-320: norm of rhs,
 *                first residual, r~, zeroed work vectors) on the resident fields, after which
 *                kry_iterate steps as usual;
 *   kry_resolve  = kry_reset + iterate to convergence [+ copy the solution to phi unless NULL].
 * A solve through kry_update_* + kry_resolve gives the bits of kry_solve_* on the same fields.
 * kry_stats.staged_fields counts the fields that crossed into the library for the last solve. */
int kry_update_rhs(kry_handle h, const double *rhs);
int kry_update_phi(kry_handle h, const double *phi);
int kry_update_lambda(kry_handle h, const double *lambda);
int kry_reset(kry_handle h);
int kry_resolve(kry_handle h, double *phi, int debug);

/* Per-kernel timing (HIP events around every launch of the fused path, accumulated in
 * kry_stats.kernel_ms / kernel_calls) on or off from the next kry_iterate on; the start value
 * comes from KRY_PROFILE.  A measurement aid without counterpart in the reference. */
int kry_set_profile(kry_handle h, int on);

int kry_get_stats(kry_handle h, kry_stats *out);

/* Several slabs in ONE process (slab_count > 1, no other process): the analogue of the reference's
 * exchange between MPI ranks (BICGStab3D/BoundaryHandler.C:1633-1677, BICGStab3D/Krylov.C:239-240)
 * inside a process that drives several GPUs.  When every slab has a device of its own and the
 * devices can address each other's memory, the slabs are linked like the processes of
 * kry_mail_connect: the kernel that finishes a reduction adds the slabs' sums through mailboxes
 * in the peers' memory, a kernel of the sender stores the interface planes into landing slots of
 * the neighbour -- per dot product and per operator output no copy, no event between slabs and no
 * extra kernel is queued, and the sums are added in slab order either way (same bits).
 * kry_create sets the link up and tries it once (2 s at most); if that fails the slabs exchange
 * by device-to-device copies ordered by events.  Environment: KRY_SLAB_LINK=0 keeps the copies,
 * =1 also links two slabs that share a device.  kry_stats.slab_link tells which route is in use,
 * this function why ("on", or the reason it is off); the string lives as long as the handle. */
const char *kry_slab_link_note(kry_handle h);
/* `norm` of every outer iteration (solveLin_BICGStab.C:464); returns the count. */
int kry_get_history(kry_handle h, double *norms, int cap);

/* Wall-clock milliseconds of every outer iteration since the last load, the data
 * behind BiCGStabSolver::stepRuntimes / steptimeMin / steptimeMax of the reference's
 * OpenCL solver (BiCGStab3D_CL/BiCGStabCL.hpp:137-143, 196-200).  The fused paths
 * run iterations in batches without a host round trip; a batch's time is spread
 * evenly over its iterations.  Returns the count. */
int kry_get_step_times(kry_handle h, double *ms, int cap);

/* Single operators on host fields, for kernel-level checks.  They run the same
 * device kernels the solver uses.
 *   kry_op_do_bcs   : BoundaryHandler3D::do_BCs(dist, 1)   BICGStab3D/BoundaryHandler.C:370-575
 *   kry_op_spmv_*   : Krylov::multiply_withMat              BICGStab3D/Krylov.C:500-541 / 407-496
 *   kry_op_dot      : Krylov::dot_product                   BICGStab3D/Krylov.C:161-244
 *   kry_op_l2norm   : Krylov::get_l2Norm                    BICGStab3D/Krylov.C:36-157
 * `fused` selects the LDS-tiled Dirichlet kernel (requires all faces Dirichlet)
 * instead of the generic one.
 * They borrow the handle's fields: a problem loaded with kry_load_* is gone
 * afterwards (kry_iterate / kry_fetch_phi then return KRY_ERR_STATE until the
 * next load or solve). */
int kry_op_do_bcs(kry_handle h, double *field);
int kry_op_spmv_const(kry_handle h, double *psi, const double *lambda, double Dxx, double Dyy,
                      double Dzz, double *out, int apply_bcs, int fused);
int kry_op_spmv_var(kry_handle h, const double *psi, const double *lambda, const double *Dxx,
                    const double *Dyy, const double *Dzz, const double *Dxy, int use_offdiag,
                    double *out, int apply_bcs, int fused);
int kry_op_dot(kry_handle h, const double *a, const double *b, double *result);
int kry_op_l2norm(kry_handle h, const double *v, double *result);

/* kry_op_dot / kry_op_l2norm for callers that cannot allocate.  No counterpart in the
 * reference: it is for code under HIP graph capture, tight serving loops and harnesses
 * that forbid hipMalloc.  The same weighted sum (Krylov::dot_product, or with norm_mode
 * != 0 Krylov::get_l2Norm of a == b) by the same kernel in the same summation order, but
 *   - a and b are DEVICE pointers to fields in the host layout described at the top
 *     (ghost-inclusive, x fastest, no padding) with last grid-point indices mx[3];
 *   - the result (one double) is written to the DEVICE pointer `result`;
 *   - everything is queued on `stream` (a hipStream_t, NULL = the default stream) of
 *     HIP device `device`; the call neither synchronises nor allocates: the kernel's
 *     partial sums and ticket live in a file-scope device array of about 33 KB that
 *     every GPU the library runs on holds until the process ends.
 * The array's address on a GPU is asked from the runtime once per GPU, which may load the
 * library's code there (and allocate): kry_device_check(device) and kry_create do that, so
 * call one of them for the device -- or make one warm-up call of this function -- before
 * the first call inside a stream capture; every later call only queues a 256-byte memset
 * and one kernel, both of which a capture records.
 * Fields of more than KRY_STATIC_DOT_MAX_DOUBLES doubles (ghosts included) are refused
 * with KRY_ERR_ARG.  All calls on one GPU share that GPU's array: calls that overlap in
 * time on one GPU (different streams, or threads) race; calls ordered on one stream do not. */
#define KRY_STATIC_DOT_MAX_DOUBLES 2147483648L
int kry_op_dot_static(int device, const double *a, const double *b, const int mx[3], int norm_mode,
                      double *result, void *stream);

const char *kry_status_string(int status);
const char *kry_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
