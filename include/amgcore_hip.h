/*
 * amgcore_hip -- MI355X (gfx950) native replacement for the solve-phase
 * boundary of PyAMG: the `pyamg.amg_core` relaxation kernels, scipy's
 * csr/bsr matvec at the reference's call sites, and a device-resident
 * multigrid hierarchy that runs multilevel_solver.solve() entirely in HBM.
 *
 * C ABI only: plain pointers, ints and doubles; no C++/torch types.  fp64
 * values, int32 indices (the reference instantiates `int` indices only,
 * pyamg/amg_core/amg_core.i:108).  Every function returns 0 on success or a
 * negative AMG_E* code; amg_last_error() gives the message.  Nothing here
 * falls back to the CPU: without a usable HIP device every compute entry
 * point fails with AMG_ENODEV.
 *
 * Section 1 mirrors the reference's SWIG table one-to-one (each (T*, int size)
 * pair of the C++ prototype is kept, so a binding can forward numpy arrays
 * unchanged).  All reference paths are relative to /root/reference.
 */
#ifndef AMGCORE_HIP_H
#define AMGCORE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AMG_OK        0
#define AMG_EINVAL   -1   /* bad argument / inconsistent sizes */
#define AMG_ENODEV   -2   /* no HIP device / HIP runtime error */
#define AMG_ENOMEM   -3
#define AMG_ESTATE   -4   /* hierarchy not finalised, level missing, ... */
#define AMG_ENOTIMPL -5

const char *amg_last_error(void);
int amg_device_count(void);
/* "gfx950:..." of the selected device, or "" */
const char *amg_device_name(int device);

/* ------------------------------------------------------------------------ */
/* 1. amg_core drop-ins.  HOST pointers (numpy buffers), results written in  */
/*    place exactly like the reference; x (and temp / z) are mutated.        */
/*    Arithmetic per row is the reference's (same left-to-right sums, no     */
/*    FMA contraction); sequential sweeps are executed by dependency-level   */
/*    scheduling, which reproduces the sequential iterates bit for bit.      */
/* ------------------------------------------------------------------------ */

/* pyamg/amg_core/relaxation.h:33-62, called from pyamg/relaxation/relaxation.py:349 */
int amgcore_gauss_seidel_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                             const double Ax[], int Ax_size, double x[], int x_size,
                             const double b[], int b_size,
                             int row_start, int row_stop, int row_step);
/* relaxation.h:89-173, relaxation.py:353 */
int amgcore_bsr_gauss_seidel_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                 const double Ax[], int Ax_size, double x[], int x_size,
                                 const double b[], int b_size,
                                 int row_start, int row_stop, int row_step, int blocksize);
/* relaxation.h:201-239, relaxation.py:416 */
int amgcore_jacobi_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                       const double Ax[], int Ax_size, double x[], int x_size,
                       const double b[], int b_size, double temp[], int temp_size,
                       int row_start, int row_stop, int row_step,
                       const double omega[], int omega_size);
/* relaxation.h:267-360, relaxation.py:425 */
int amgcore_bsr_jacobi_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                           const double Ax[], int Ax_size, double x[], int x_size,
                           const double b[], int b_size, double temp[], int temp_size,
                           int row_start, int row_stop, int row_step, int blocksize,
                           const double omega[], int omega_size);
/* relaxation.h:394-426, relaxation.py:739 */
int amgcore_gauss_seidel_indexed_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                     const double Ax[], int Ax_size, double x[], int x_size,
                                     const double b[], int b_size, const int Id[], int Id_size,
                                     int row_start, int row_stop, int row_step);
/* relaxation.h:465-496, relaxation.py:818 */
int amgcore_jacobi_ne_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                          const double Ax[], int Ax_size, double x[], int x_size,
                          const double b[], int b_size, const double Tx[], int Tx_size,
                          double temp[], int temp_size,
                          int row_start, int row_stop, int row_step,
                          const double omega[], int omega_size);
/* relaxation.h:935-1007, relaxation.py:272-277: one sweep of multiplicative overlapping Schwarz over
 * the subdomains row_start, row_start+row_step, ... (Sj/Sp sorted index lists, Tx/Tp their inverted
 * diagonal blocks, row-major) */
int amgcore_overlapping_schwarz_csr_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                        const double Ax[], int Ax_size, double x[], int x_size,
                                        const double b[], int b_size, const double Tx[], int Tx_size,
                                        const int Tp[], int Tp_size, const int Sj[], int Sj_size,
                                        const int Sp[], int Sp_size, int nsdomains, int nrows,
                                        int row_start, int row_stop, int row_step);
/* relaxation.h:529-561, relaxation.py:907 */
int amgcore_gauss_seidel_ne_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                const double Ax[], int Ax_size, double x[], int x_size,
                                const double b[], int b_size,
                                int row_start, int row_stop, int row_step,
                                const double Tx[], int Tx_size, double omega);
/* relaxation.h:594-631, relaxation.py:995 (A in CSC) */
int amgcore_gauss_seidel_nr_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                const double Ax[], int Ax_size, double x[], int x_size,
                                double z[], int z_size,
                                int col_start, int col_stop, int col_step,
                                const double Tx[], int Tx_size, double omega);
/* relaxation.h:661-728, relaxation.py:503 */
int amgcore_block_jacobi_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                             const double Ax[], int Ax_size, double x[], int x_size,
                             const double b[], int b_size, const double Tx[], int Tx_size,
                             double temp[], int temp_size,
                             int row_start, int row_stop, int row_step,
                             const double omega[], int omega_size, int blocksize);
/* relaxation.h:755-810, relaxation.py:588 */
int amgcore_block_gauss_seidel_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                   const double Ax[], int Ax_size, double x[], int x_size,
                                   const double b[], int b_size, const double Tx[], int Tx_size,
                                   int row_start, int row_stop, int row_step, int blocksize);

/* scipy.sparse._sparsetools.csr_matvec / bsr_matvec (third party) as used by
 * `A * x` at pyamg/multilevel.py:496,498,544; pyamg/util/linalg.py:112;
 * pyamg/relaxation/relaxation.py:661,666.  y is ACCUMULATED into, as scipy does. */
int amgcore_csr_matvec_f64(int n_row, int n_col, const int Ap[], const int Aj[],
                           const double Ax[], const double x[], double y[]);
int amgcore_bsr_matvec_f64(int n_brow, int n_bcol, int R, int C, const int Ap[], const int Aj[],
                           const double Ax[], const double x[], double y[]);
/* pyamg/util/linalg.py:17-53 norm(x) (2-norm) */
int amgcore_norm2_f64(const double x[], long n, double *result);

/* ------------------------------------------------------------------------ */
/* 2. Device-resident hierarchy: multilevel_solver.solve()/__solve()         */
/*    (pyamg/multilevel.py:316-548) with A_l, P_l, R_l, smoother constants   */
/*    and all work vectors in HBM.                                           */
/* ------------------------------------------------------------------------ */
typedef struct amg_hier amg_hier;

enum { AMG_FMT_CSR = 0, AMG_FMT_BSR = 1 };
enum { AMG_MAT_A = 0, AMG_MAT_P = 1, AMG_MAT_R = 2 };
enum { AMG_PRE = 0, AMG_POST = 1 };
enum { AMG_CYCLE_V = 0, AMG_CYCLE_W = 1, AMG_CYCLE_F = 2, AMG_CYCLE_AMLI = 3 };
enum { AMG_SWEEP_FORWARD = 0, AMG_SWEEP_BACKWARD = 1, AMG_SWEEP_SYMMETRIC = 2 };
/* smoother kinds = the relaxation.py entry points a smoothing.py closure calls */
enum { AMG_SM_NONE = 0, AMG_SM_JACOBI = 1, AMG_SM_GAUSS_SEIDEL = 2, AMG_SM_SOR = 3,
       AMG_SM_POLYNOMIAL = 4, AMG_SM_BLOCK_JACOBI = 5, AMG_SM_BLOCK_GAUSS_SEIDEL = 6,
       AMG_SM_GAUSS_SEIDEL_INDEXED = 7, AMG_SM_SCHWARZ = 8, AMG_SM_GAUSS_SEIDEL_NE = 9,
       AMG_SM_GAUSS_SEIDEL_NR = 10, AMG_SM_JACOBI_NE = 11 };

typedef struct {
    int kind;            /* AMG_SM_* */
    int iterations;      /* >= 1 */
    int sweep;           /* AMG_SWEEP_* (gauss_seidel, sor, block_gauss_seidel, indexed) */
    double omega;        /* jacobi / block_jacobi: omega already divided by rho; sor: omega */
    int ncoef;           /* polynomial: Horner coefficients (chebyshev: -coeffs[:-1]) */
    const double *coef;  /* host pointer, copied */
    int blocksize;       /* block_jacobi / block_gauss_seidel */
    const double *Dinv;  /* host pointer, (n/bs)*bs*bs row-major inverse diagonal blocks, copied;
                            normal-equation kinds: n entries 1/diag(A A^H) (ne, jacobi_ne) or 1/diag(A^H A) (nr) */
    const int *indices;  /* gauss_seidel_indexed: row order (host pointer, copied) */
    int nindices;
    /* schwarz (relaxation.py:172-278; level A must be CSR or BSR(1,1)): sorted subdomain index lists
     * Sj[Sp[d]..Sp[d+1]) and their inverted diagonal blocks Tx[Tp[d]..Tp[d+1]), row-major
     * (relaxation.schwarz_parameters); host pointers, copied */
    const int *Sj, *Sp, *Tp;
    const double *Tx;
    int nsdomains;
} amg_smoother_desc;

#define AMG_SM_CALLBACK 100
/* A relaxation supplied by the caller (the device-resident Krylov smoothers, pyamg/relaxation/smoothing.py:481-509):
 * called inside the cycle with DEVICE pointers to the level's iterate and right-hand side; it enqueues its work on
 * amg_hier_stream(h) (amg_hier_apply, amg_dev_*) and returns 0, or non-zero to abort the solve. */
typedef int (*amg_relax_callback)(void *user, int level, double *x_dev, const double *b_dev);
/* A coarse solver supplied by the caller (multilevel.py:642-692: Krylov names, callables): HOST vectors of the
 * coarsest level's size; x is zero on entry. */
typedef int (*amg_coarse_callback)(void *user, int n, const double *b_host, double *x_host);

/* flags for amg_hier_solve */
#define AMG_SOLVE_X0_ZERO        1  /* caller guarantees x is all zeros on entry */
#define AMG_SOLVE_NO_EARLY_STOP  2  /* run exactly maxiter cycles; residual norms stay on
                                       the device until the end (no per-iteration sync) */
#define AMG_SOLVE_DEVICE_VECTORS 4  /* b and x are DEVICE pointers */

amg_hier *amg_hier_create(int nlevels, int device);
void amg_hier_destroy(amg_hier *h);

/* Copy one operator of level `lvl` into HBM.  fmt/R/C describe the scipy
 * container (csr_matrix, or bsr_matrix with blocksize (R,C)); nrows/ncols are
 * scalar dimensions; Ax has nnz (CSR) or nblocks*R*C (BSR, blocks row-major)
 * entries.  Pointers are host pointers unless on_device != 0, in which case
 * they are device pointers (CSR / BSR(1,1) only); the arrays are copied either way. */
int amg_hier_set_matrix(amg_hier *h, int lvl, int which, int fmt, int nrows, int ncols,
                        int R, int C, const int *Ap, const int *Aj, const double *Ax,
                        int on_device);
int amg_hier_set_smoother(amg_hier *h, int lvl, int which, const amg_smoother_desc *d);
/* Block smoothers act on A re-blocked to their own blocksize
 * (pyamg/relaxation/relaxation.py:471,563: A = A.tobsr(blocksize=(bs,bs))).  When
 * level A is not already BSR(bs,bs), pass that re-blocked copy here AFTER
 * amg_hier_set_smoother; which = AMG_PRE / AMG_POST, or 2 for the coarse smoother. */
int amg_hier_set_block_matrix(amg_hier *h, int lvl, int which, int nbrows, int bs, const int *Ap,
                              const int *Aj, const double *Ax);
/* Normal-equation smoothers (relaxation.py:744-997; level A must be CSR or BSR(1,1)) work on other
 * layouts of A, passed AFTER amg_hier_set_smoother: slot 0 = A by columns (the CSC arrays: indptr over
 * columns, row indices ascending, values) for gauss_seidel_nr and jacobi_ne; slot 1 = A by rows with
 * sorted column indices, needed by gauss_seidel_nr only when the level's A has unsorted rows. */
int amg_hier_set_aux_matrix(amg_hier *h, int lvl, int which, int slot, int nmajor, int nminor,
                            const int *Ap, const int *Aj, const double *Ax);
/* coarse_grid_solver('pinv'/'pinv2'/'lu'/'cholesky'/'splu'): a dense n x n
 * row-major operator M with x = M b (multilevel.py:608-641) */
int amg_hier_set_coarse_dense(amg_hier *h, const double *M, int n);
/* coarse_grid_solver(<relaxation name>) (multilevel.py:662-680): x = 0, then the smoother */
int amg_hier_set_coarse_smoother(amg_hier *h, const amg_smoother_desc *d);
int amg_hier_set_callback_smoother(amg_hier *h, int lvl, int which, amg_relax_callback cb, void *user);
int amg_hier_set_coarse_callback(amg_hier *h, amg_coarse_callback cb, void *user);
/* y = M x for a stored operator on DEVICE vectors, enqueued on the hierarchy's stream (which = AMG_MAT_A/P/R);
 * amg_hier_apply_aux: the auxiliary operator of a smoother slot (amg_hier_set_aux_matrix: A by columns = A^T by rows) */
int amg_hier_apply(amg_hier *h, int lvl, int which, const double *x_dev, double *y_dev);
int amg_hier_apply_aux(amg_hier *h, int lvl, int which, int slot, const double *x_dev, double *y_dev);
/* device scratch of the hierarchy for the reductions of amg_dev_dot_host / amg_dev_norm_host (>= 1040 doubles) */
double *amg_hier_scratch(amg_hier *h);
/* allocate work vectors, build Gauss-Seidel level schedules */
int amg_hier_finalize(amg_hier *h);
/* After amg_hier_finalize: free the CSR arrays of operators whose stencil / sliced form serves every application the
 * hierarchy's cycles make of them (A_l under polynomial / Jacobi smoothers from a complete stencil form; A_l (l >= 1),
 * P_l, R_l under polynomial smoothers from a complete sliced form).  Lossless forms: same bits.  Returns the bytes
 * freed (0 for partitioned hierarchies).  Afterwards a smoother that needs the arrays cannot be attached (AMG_ESTATE). */
long amg_hier_release_sources(amg_hier *h);

/* multilevel_solver.solve(b, x0, tol, maxiter, cycle) with accel=None
 * (multilevel.py:316-471).  x holds x0 on entry and the solution on exit;
 * residuals must have room for maxiter+1 doubles; *nres = number written. */
int amg_hier_solve(amg_hier *h, const double *b, double *x, double tol, int maxiter, int cycle,
                   double *residuals, int *nres, int flags);
/* multilevel_solver.solve(b, x0, tol, maxiter, cycle, accel='cg'): conjugate gradients
 * (pyamg/krylov/_cg.py:84-179) preconditioned by one cycle from a zero guess
 * (aspreconditioner, multilevel.py:306-314), all vectors resident.  residuals (room for
 * maxiter+1) receives the preconditioner-norm history sqrt(<r,Mr>); *info = 0, or -1 when
 * an indefinite operator / preconditioner stops the iteration as in the reference. */
int amg_hier_pcg(amg_hier *h, const double *b, double *x, double tol, int maxiter, int cycle,
                 double *residuals, int *nres, int *info, int flags);
/* one multilevel_solver.__solve(0, x, b, cycle) (multilevel.py:473-548) on
 * host (flags=0) or device (AMG_SOLVE_DEVICE_VECTORS) vectors */
int amg_hier_cycle(amg_hier *h, const double *b, double *x, int cycle, int flags);
/* levels[lvl].presmoother(A, x, b) / postsmoother on host vectors */
int amg_hier_relax(amg_hier *h, int lvl, int which, const double *b, double *x);
/* y = M x for a stored operator (host vectors) */
int amg_hier_matvec(amg_hier *h, int lvl, int which, const double *x, double *y);

/* bookkeeping for measurement */
/* algorithmic bytes of one cycle per SURVEY.md section 8(d) */
double amg_hier_cycle_bytes(amg_hier *h, int cycle);
/* storage form level lvl's A is applied from: 0 CSR, 1 offset-pattern, 2 stencil, 3 sliced (SELL-64-sigma) */
int amg_hier_operator_form(amg_hier *h, int lvl);
/* Value index for a level whose A is in stencil form and holds at most 255 distinct values (constant-coefficient
 * stencils): one-byte codes into a dictionary instead of the 8-byte values; the products use the same doubles, so
 * results are bit-identical.  r3: built automatically when the operator is set (amg_set_value_index(0) or
 * AMG_VALUE_INDEX=0 turns that off; an operator whose values do not compress keeps them).  This entry toggles one
 * level: on > 0 builds/enables it and returns the number of distinct values (0: not applicable, < 0: error);
 * on == 0 returns to the plain values; on < 0 queries (distinct values when in use, else 0). */
int amg_hier_value_index(amg_hier *h, int lvl, int on);
void amg_set_value_index(int on);       /* process-wide default for operators set afterwards */
int amg_value_index_enabled(void);
/* Setup-side (replaces the host sweeps behind pyamg/aggregation/aggregation.py:313-320 `relaxation_as_linear_operator(
 * ('gauss_seidel', ...), A, 0) * B`, i.e. amg_core gauss_seidel of relaxation.h:34-62 on A x = 0): nsweeps Gauss-Seidel
 * sweeps over level lvl's operator in its own row order, from the CSR arrays in HBM; dirs[k] != 0 = descending rows;
 * b == NULL means a zero right-hand side; x (host) is updated in place; bit-identical to the sequential loop.
 * AMG_ENOTIMPL: a row of more than 8 entries, or the CSR arrays are not on the device -- keep the host sweep. */
int amg_hier_gs_natural(amg_hier *h, int lvl, double *x, const double *b, const unsigned char *dirs, int nsweeps);

/* bytes of one r = b - A x on level lvl: moved = 0 the CSR figure of SURVEY.md 8(d), 1 what the form in use streams */
double amg_hier_operator_bytes(amg_hier *h, int lvl, int moved);
/* bytes one solve() iteration needs as this library runs it: offset-pattern operators without
 * their column indices, and one level-0 application less when the outer residual is kept */
double amg_hier_cycle_bytes_moved(amg_hier *h, int cycle);
/* device time of the timed part of the last amg_hier_solve, ms (hipEvents on the solve stream) */
double amg_hier_last_solve_ms(amg_hier *h);
long amg_hier_device_bytes(amg_hier *h);
/* the hipStream_t the hierarchy launches on, as void* */
void *amg_hier_stream(amg_hier *h);
/* raw device access to level work vectors for a caller that stays on the device */
double *amg_hier_dev_x(amg_hier *h);
double *amg_hier_dev_b(amg_hier *h);

/* Stand-alone device SpMV benchmark hook: y = A x on a stored operator,
 * `reps` back-to-back launches timed with hipEvents on the hierarchy stream;
 * returns average ms per launch in *ms. */
int amg_hier_time_spmv(amg_hier *h, int lvl, int which, int mode, int reps, double *ms);
/* same for one application of a stored smoother (which = AMG_PRE / AMG_POST / 2 = coarse smoother)
 * to the level's resident vectors */
int amg_hier_time_relax(amg_hier *h, int lvl, int which, int reps, double *ms);

/* ------------------------------------------------------------------------ */
/* 3. Device-pointer API: one operator in HBM + the vector kernels on        */
/*    caller-owned DEVICE vectors and stream (hipStream_t as void*).  The    */
/*    row-partitioned multi-GPU cycle (pyamg_amd/distributed.py) is built    */
/*    from these; halos move through torch.distributed (RCCL).               */
/* ------------------------------------------------------------------------ */
typedef struct amg_mat amg_mat;
/* copies a host CSR into HBM (local rows of a partitioned operator; column indices local) */
amg_mat *amg_mat_create(int device, int nrows, int ncols, const int *Ap, const int *Aj, const double *Ax);
void amg_mat_destroy(amg_mat *m);
long amg_mat_nnz(amg_mat *m);
/* storage form the operator is applied from: 0 CSR, 1 offset-pattern, 2 stencil */
int amg_mat_form(amg_mat *m);
/* one csr_stream launch.  mode: 0 out=Mx  1 out+=Mx  2 out=b-Mx  3 out=b-Mx,out2=c0*out
 * 4 out=c0*b+Mx  5 out=v2+(c0*b+Mx)  6 Jacobi (CSR rounding)  7 Jacobi (BSR(1,1) rounding);
 * xg is the gathered vector (owned entries followed by the halo) */
int amg_mat_apply(amg_mat *m, int mode, const double *xg, const double *b, const double *v2, double *out,
                  double *out2, double c0, double gscale, void *stream);   /* products are a_ij*(gscale*xg_j); 0 = 1 */
/* the same launch over rows [row_lo, row_hi) only (interior rows first, boundary rows after the halo arrived) */
int amg_mat_apply_rows(amg_mat *m, int mode, int row_lo, int row_hi, const double *xg, const double *b,
                       const double *v2, double *out, double *out2, double c0, double gscale, void *stream);
/* Gauss-Seidel over the operator's own rows (order NULL: 0..nrows-1, else an index list as for
 * gauss_seidel_indexed); columns >= nrows (halo) stay frozen: GS inside a rank, Jacobi across */
int amg_mat_build_gs(amg_mat *m, const int *order, int norder);
int amg_mat_gs_levels(amg_mat *m);
int amg_mat_gs_sweep(amg_mat *m, double *x, const double *b, int reverse, int bsr1, void *stream);
/* a whole sequence of directional sweeps (seq[k] != 0: backward) as the in-cycle smoothers issue them: the
 * dataflow form runs up to four of them per launch */
int amg_mat_gs_sweeps(amg_mat *m, double *x, const double *b, const unsigned char *seq, int nseq, int bsr1, void *stream);
int amg_dev_scale(double *out, const double *in, double c, long n, void *stream);
int amg_dev_axpy(double *x, const double *h, long n, void *stream);
int amg_dev_axpy_scaled(double *x, const double *r, double c, long n, void *stream);   /* x += c*r */
int amg_dev_norm2(const double *x, long n, double *scratch, double *result_dev, void *stream);
int amg_dev_dot(const double *x, const double *y, long n, double *scratch, double *result_dev, void *stream);
int amg_dev_dense_apply(const double *Mt, const double *b, double *x, int n, void *stream);
int amg_dev_gather(double *out, const double *in, const int *idx, long n, void *stream);
/* device-resident vectors for the Krylov methods around the cycle: allocation, copies (kind 0 H2D, 1 D2H, 2 D2D),
 * BLAS-1 updates, and reductions that return ONE scalar to the host */
double *amg_dev_alloc(long n);
void amg_dev_free(double *p);
int amg_dev_copy(double *dst, const double *src, long n, int kind, void *stream);
int amg_dev_fill(double *x, double v, long n, void *stream);
int amg_dev_axmy(double *w, const double *v, double a, long n, void *stream);          /* w -= a v */
int amg_dev_scale_add(double *p, double beta, const double *z, long n, void *stream);  /* p = beta p + z */
int amg_dev_sub(double *out, const double *a, const double *b, long n, void *stream);  /* out = a - b */
int amg_dev_divide(double *w, double a, long n, void *stream);                          /* w /= a */
int amg_dev_dot_host(const double *x, const double *y, long n, double *scratch, double *result, void *stream);
int amg_dev_norm_host(const double *x, long n, double *scratch, double *result, void *stream);

/* ------------------------------------------------------------------------ */
/* 4. Row-partitioned hierarchies: one process per GPU of a node, every level */
/*    cut into contiguous row blocks, halos pushed GPU-to-GPU over xGMI, one   */
/*    8-byte all-reduce per residual norm.  The reference   */
/*    has no distributed path; the arithmetic contract is the single-GPU cycle  */
/*    (pyamg/multilevel.py:316-548): every row keeps its stored summation order.*/
/* ------------------------------------------------------------------------ */
typedef struct amg_comm amg_comm;
/* transport 0 = "peer": arenas in fine-grained HBM mapped into every rank through HIP IPC; a hand-off is a kernel
 * that stores into the consumer's arena plus a system-scope flag, the consumer polls the flag from a one-wave kernel
 * with a wall-clock budget -- all on the hierarchy's stream, capturable in a hipGraph.
 * transport 1 = "rccl": grouped ncclSend / ncclRecv and ncclAllReduce on the same stream (librccl dlopen-ed). */
amg_comm *amg_comm_create(int rank, int world, int device, int transport);
void amg_comm_destroy(amg_comm *c);
/* declare an exchange plan: counts[dst * world + src] doubles travel from src to dst per exchange (identical matrix
 * on every rank).  Returns the channel id (>= 0). */
int amg_comm_add_channel(amg_comm *c, const int *counts);
/* fix the layout and allocate; peer transport: handle_out[64] = this rank's IPC handle, to be all-gathered */
int amg_comm_commit(amg_comm *c, unsigned char *handle_out);
/* peer transport: map the other ranks' arenas; handles = world x 64 bytes in rank order */
int amg_comm_connect(amg_comm *c, const unsigned char *handles);
/* rccl transport: rank 0 draws a 128-byte unique id (libpath NULL: "librccl.so"), every rank joins with it */
int amg_comm_rccl_unique_id(const char *libpath, unsigned char *id_out);
int amg_comm_rccl_init(amg_comm *c, const char *libpath, const unsigned char *id);
/* stand-alone use on device pointers: dst_halo <- the peers' entries (v[send_idx[k]] on their side, send_idx NULL =
 * identity); *result = sqrt(sum over ranks of *partial), added in rank order */
int amg_comm_exchange(amg_comm *c, int channel, const double *v, const int *send_idx, double *dst_halo, void *stream);
int amg_comm_allreduce_sqrt(amg_comm *c, int channel, const double *partial, double *result, void *stream);
/* peer transport: non-zero (with an error message) if some wait ran out of its budget */
int amg_comm_check(amg_comm *c);
/* Attach a communicator BEFORE the operators are set: level operators are then the rank's rows with columns
 * renumbered [owned | halo]; reduce_channel = a channel with count 1 between every pair of ranks (itself included). */
int amg_hier_set_comm(amg_hier *h, amg_comm *comm, int reduce_channel);
int amg_hier_set_partition(amg_hier *h, int lvl, int n_own, int n_halo, int channel, const int *send_idx, int i0, int i1);
int amg_hier_set_gather(amg_hier *h, int lvl, int channel, int rows);
int amg_hier_set_coarse_gather(amg_hier *h, int channel, int lo);
int amg_hier_comm_check(amg_hier *h);

/* Setup-time helper (hierarchy construction, not the cycle): Arnoldi iteration on
 * M = diag(dinv) * A_lvl (dinv NULL: M = A_lvl) as in pyamg/util/linalg.py:173-279, used for
 * the spectral-radius estimates behind omega and the Chebyshev bounds.  H is
 * (maxiter+1) x maxiter row-major on the host; the basis stays on the device. */
int amg_arnoldi(amg_hier *h, int lvl, const double *dinv, const double *v0, int maxiter,
                double breakdown_tol, double *H, int *steps, int *breakdown);
/* v = V[:, :m] @ coef (restart vector); v is a host buffer of length n */
int amg_arnoldi_combine(amg_hier *h, const double *coef, int m, double *v);
void amg_arnoldi_free(amg_hier *h);

/* replay each iteration (cycle + residual norm) from a hipGraph once its launch sequence has
 * been seen (default on; env AMG_HIP_GRAPHS=0 disables).  Speed only: same kernels, same order. */
void amg_hier_use_graphs(amg_hier *h, int on);
/* on (default): amg_hier_solve keeps the residual vector b - A x it forms for the convergence test
 * (multilevel.py:461) and a polynomial pre-smoother on level 0 starts from it instead of forming
 * the same b - A x again (relaxation.py:655).  Same bits either way; off restores the two passes. */
void amg_hier_keep_residual(amg_hier *h, int on);

/* tuning knobs (speed only): 0 = scalar loads, 1 = 16-byte loads in the CSR stream kernel;
 * XCD chunk: consecutive row blocks given to one XCD (0 = round-robin dispatch order) */
void amg_set_stream_variant(int v);
/* 1 (default): the CSR stream kernel runs as persistent workgroups that prefetch the next row block's row pointers,
 * entries and epilogue operands while they gather for the current one; 0: one row block per workgroup.  Same bits. */
void amg_set_stream_pipe(int on);
void amg_set_xcd_chunk(int c);
/* 1 (default): operators in offset-pattern form map row blocks to XCDs periodically in the slowest
 * grid axis, so one XCD's L2 serves a row's neighbours in the planes above and below; 0: chunked */
void amg_set_xcd_period(int on);
/* 1 (default): operators whose rows are subsets of one stencil of <= 32 offsets are applied from
 * the stencil form (padded values + row masks, no indices); 0: from the pattern / CSR forms */
void amg_set_stencil_form(int on);
/* stencil form: two consecutive rows per lane, every streamed operand a 16-byte access (half the vector-memory
 * instructions for the same bytes): 1 (default) for launches of 30 M rows and more with stencils of up to 7 offsets
 * (where it is measured faster), 2 always, 0 never (one row per lane).  Same bits. */
void amg_set_stencil_pairs(int on);
/* operators without grid structure (Galerkin operators, restriction): 1 (default) whole-operator applications run from
 * the sliced form (rows sorted by length in windows of 256 -- restrictions 64 --, slices of 64 rows stored entry-major: one lane per row,
 * no row pointer, no LDS); 0: from CSR.  Same bits. */
void amg_set_sell_form(int on);
/* the sliced form's column indices as 16-bit window codes (two per word, 16 window origins per slice; slices whose
 * columns need more windows keep their 32-bit indices): 1 (default) built with the form and read by the kernel,
 * 0: 32-bit indices.  10 B per stored entry instead of 12; same columns, same order, same bits. */
void amg_set_sell_index16(int on);
/* runs of narrow Gauss-Seidel dependency levels are swept by one workgroup in one launch: 2 (default) the sweep runs
 * in level-order numbering, new values are handed from level to level through LDS and everything else is requested
 * two levels ahead; 1 operands gathered back from L2 after each barrier (also what index lists and partitioned
 * levels use); 0: one launch per level.  Same bits. */
void amg_set_gs_chain(int on);
/* levels too wide for a chain, one launch each: 1 (default) the launch's workgroups get their entry ranges in the kernel
 * arguments (gs_level_kernel: two memory round trips per launch), 0 the general stream kernel (three) */
void amg_set_gs_level_hint(int on);
/* Gauss-Seidel sweeps (pyamg/amg_core/relaxation.h:34-62, :90-173 with 1x1 blocks) as ONE persistent launch per
 * smoother application ("dataflow" form, csrc/gsflow.hip): rows wait for their own operands instead of for a
 * dependency level.  1 (default): schedules whose levels would otherwise be launches of their own or long-row chains
 * (3-D operators, coarse levels of smoothed-aggregation hierarchies); 2: every schedule the form exists for; 0: never
 * (level launches and chained sweeps).  Read when a schedule is built and when it is swept.  Same bits. */
void amg_set_gs_flow(int mode);
/* look-ahead of the dataflow sweep in dependency levels (resident waves = look-ahead x average chunks per level);
 * 0: the default (4) */
void amg_set_gs_flow_lookahead(int levels);
/* host-synchronous: nonzero when a wave of a dataflow sweep gave up waiting (4 s budget) since the last call -- the
 * iterates of that sweep are then unusable.  amg_hier_solve checks it before it returns. */
int amg_gs_flow_status(void);
/* A of a BSR(bs,bs) level can be applied straight from its blocks (8 B per entry + 4 B per block) instead
 * of from the CSR expansion (12 B per entry); same summation order, same bits.  0: never, 1 (default):
 * for blocks of 3x3 and larger (where it is measured faster), 2: always */
void amg_set_bsr_spmv(int on);
/* 1: operators uploaded from now on also get 16-bit column codes (row blocks whose columns fit 16
 * windows of 4096) and the stream kernel reads those; 0 (default): always the 32-bit indices.
 * Lossless; off by default because the measured gain is within +-8 % per operator */
void amg_set_index16(int on);
/* products per workgroup aimed at when choosing rows per workgroup (default 2048 = one LDS tile) */
void amg_set_tile_target(int t);

/* ---- setup on the device: Galerkin products -------------------------------------------------------------------
 * Ac = (R * A) * P with scipy's csr_matmat arithmetic and output order (what pyamg/aggregation/aggregation.py:425-426
 * computes as R * A * P): per output row the products accumulate in the order (entry of the left row, entry of the
 * right row), columns come out in reverse first-touch order, exact zeros are dropped -- bit-identical to scipy.
 * A is level `level` of a hierarchy handle that already holds it in HBM as CSR (the handle behind the setup-time
 * spectral-radius estimate); R, P are host CSR arrays with 64-bit row pointers.  Cp receives n_coarse + 1 offsets;
 * amg_galerkin_fetch copies the Cp[n_coarse] columns / values to the host and releases the product.
 * AMG_EINVAL (nothing allocated) when the operator is not held as CSR or a row has more distinct result columns than
 * the device tables hold (2048 for rows of more than 1024 products on large levels). */
typedef struct amg_galerkin amg_galerkin;
int amg_hier_galerkin(amg_hier *h, int level, int n_coarse, const int64_t *Rp, const int *Rj, const double *Rx,
                      const int64_t *Pp, const int *Pj, const double *Px, int64_t *Cp, amg_galerkin **out);
int amg_galerkin_fetch(amg_galerkin *g, int *Cj, double *Cx);
/* C = A * B for host CSR operands through the same kernels (n_row x n_inner times n_inner x n_col) */
int amg_csr_matmat_device(int n_row, int n_inner, int n_col, const int64_t *Ap, const int *Aj, const double *Ax,
                          const int64_t *Bp, const int *Bj, const double *Bx, int64_t *Cp, amg_galerkin **out);

#ifdef __cplusplus
}
#endif
#endif
