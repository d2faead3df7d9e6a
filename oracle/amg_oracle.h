/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * CPU restatement (plain C, fp64 values, int32 indices) of the reference's
 * solve-phase arithmetic: the amg_core relaxation kernels, scipy's sequential
 * csr_matvec/bsr_matvec, the relaxation.py shims and the multilevel.py cycle.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (pyamg_amd + libamgcore_hip.so) never links, imports or calls it.
 *
 * Pinning: the restatement is checked (tests/test_oracle_golden.py) against
 *   - the known-answer vectors of pyamg/relaxation/tests/test_relaxation.py,
 *   - kernel outputs of the reference's own _amg_core (built from
 *     amg_core_wrap.cxx into oracle/_ref) captured in tests/golden/kernels.npz,
 *   - residual histories and iterates of the reference's own
 *     multilevel_solver.solve() captured in tests/golden/hier_*.npz
 * (generator: oracle/gen_golden.py).  SpMV arithmetic lives in scipy (not under
 * /root/reference, unpinned version; 1.15.3 here): its bit-level behaviour is
 * pinned by those captured fixtures, not by reference tests.
 */
#ifndef AMG_ORACLE_H
#define AMG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- amg_core kernels (pyamg/amg_core/relaxation.h) ---- */
void oracle_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x,
                         const double *b, int row_start, int row_stop, int row_step);
void oracle_bsr_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x,
                             const double *b, int row_start, int row_stop, int row_step,
                             int blocksize);
void oracle_jacobi(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                   double *temp, int row_start, int row_stop, int row_step, const double *omega);
void oracle_bsr_jacobi(const int *Ap, const int *Aj, const double *Ax, double *x,
                       const double *b, double *temp, int row_start, int row_stop,
                       int row_step, int blocksize, const double *omega);
void oracle_gauss_seidel_indexed(const int *Ap, const int *Aj, const double *Ax, double *x,
                                 const double *b, const int *Id, int row_start, int row_stop,
                                 int row_step);
void oracle_jacobi_ne(const int *Ap, const int *Aj, const double *Ax, double *x,
                      const double *b, const double *Tx, double *temp, int row_start,
                      int row_stop, int row_step, const double *omega);
void oracle_gauss_seidel_ne(const int *Ap, const int *Aj, const double *Ax, double *x,
                            const double *b, int row_start, int row_stop, int row_step,
                            const double *Tx, double omega);
void oracle_gauss_seidel_nr(const int *Ap, const int *Aj, const double *Ax, double *x,
                            double *z, int col_start, int col_stop, int col_step,
                            const double *Tx, double omega);
void oracle_block_jacobi(const int *Ap, const int *Aj, const double *Ax, double *x,
                         const double *b, const double *Dinv, double *temp, int row_start,
                         int row_stop, int row_step, const double *omega, int blocksize);
void oracle_block_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x,
                               const double *b, const double *Dinv, int row_start,
                               int row_stop, int row_step, int blocksize);

/* ---- scipy sparsetools SpMV (third party, restated) ---- */
void oracle_extract_subblocks(const int *Ap, const int *Aj, const double *Ax, double *Tx,
                              const int *Tp, const int *Sj, const int *Sp, int nsdomains, int nrows);
void oracle_overlapping_schwarz_csr(const int *Ap, const int *Aj, const double *Ax, double *x,
                                    const double *b, const double *Tx, const int *Tp, const int *Sj,
                                    const int *Sp, int nsdomains, int nrows, int row_start,
                                    int row_stop, int row_step);

/* threads for the row-parallel loops (results independent of the count); default 1 */
void oracle_set_threads(int n);
int oracle_get_threads(void);

void oracle_csr_matvec(int n_row, const int *Ap, const int *Aj, const double *Ax,
                       const double *x, double *y /* accumulated into */);
void oracle_bsr_matvec(int n_brow, int R, int C, const int *Ap, const int *Aj,
                       const double *Ax, const double *x, double *y /* accumulated into */);

/* ---- util/linalg.py ---- */
double oracle_norm2(const double *x, long n);

/* ---- hierarchy + cycle (multilevel.py, relaxation/relaxation.py) ---- */
enum { ORACLE_FMT_CSR = 0, ORACLE_FMT_BSR = 1 };
enum { ORACLE_SM_NONE = 0, ORACLE_SM_JACOBI = 1, ORACLE_SM_GAUSS_SEIDEL = 2, ORACLE_SM_SOR = 3,
       ORACLE_SM_POLYNOMIAL = 4, ORACLE_SM_BLOCK_JACOBI = 5, ORACLE_SM_BLOCK_GAUSS_SEIDEL = 6,
       ORACLE_SM_GAUSS_SEIDEL_INDEXED = 7, ORACLE_SM_GAUSS_SEIDEL_NE = 8,
       ORACLE_SM_GAUSS_SEIDEL_NR = 9, ORACLE_SM_JACOBI_NE = 10, ORACLE_SM_SCHWARZ = 11 };
enum { ORACLE_SWEEP_FORWARD = 0, ORACLE_SWEEP_BACKWARD = 1, ORACLE_SWEEP_SYMMETRIC = 2 };
enum { ORACLE_CYCLE_V = 0, ORACLE_CYCLE_W = 1, ORACLE_CYCLE_F = 2, ORACLE_CYCLE_AMLI = 3 };

typedef struct {
    int fmt;          /* ORACLE_FMT_* */
    int nrows, ncols; /* scalar dimensions */
    int R, C;         /* block size (1,1 for CSR) */
    const int *Ap, *Aj;
    const double *Ax;
} oracle_mat;

typedef struct {
    int kind;         /* ORACLE_SM_* */
    int iterations;
    int sweep;        /* ORACLE_SWEEP_* */
    double omega;     /* jacobi/sor/block_jacobi (already divided by rho), ne/nr omega */
    int ncoef;        /* polynomial */
    const double *coef;
    int blocksize;    /* block_jacobi / block_gauss_seidel */
    const double *Dinv; /* (n/bs)*bs*bs, or Tx (inverse diag of A A^H / A^H A) for ne/nr */
    const int *indices; /* gauss_seidel_indexed */
    int nindices;
    /* block smoothers / ne / nr act on a re-formatted copy of A, as the
     * reference does with A.tobsr(bs) / lvl.Acsr / lvl.Acsc */
    const oracle_mat *Aalt;
    /* schwarz (relaxation.py:172-278): sorted subdomain index lists and their inverted diagonal
     * blocks (row-major); acts on lvl.Acsr = Aalt when A is not CSR */
    const int *Sj, *Sp, *Tp;
    const double *Tx;
    int nsdomains;
} oracle_smoother;

typedef struct oracle_hier oracle_hier;

oracle_hier *oracle_hier_create(int nlevels);
void oracle_hier_destroy(oracle_hier *h);
/* pointers are borrowed: the caller keeps the arrays alive */
void oracle_hier_set_A(oracle_hier *h, int lvl, const oracle_mat *A);
void oracle_hier_set_PR(oracle_hier *h, int lvl, const oracle_mat *P, const oracle_mat *R);
void oracle_hier_set_smoothers(oracle_hier *h, int lvl, const oracle_smoother *pre,
                               const oracle_smoother *post);
/* dense row-major n x n coarse-solve matrix (the reference's cached pinv) */
void oracle_hier_set_coarse_dense(oracle_hier *h, const double *Pinv, int n);
/* include the fork's discarded second P*coarse_x per level (multilevel.py:548) in the work */
void oracle_hier_set_duplicate_prolongation(oracle_hier *h, int on);

void oracle_relax(const oracle_mat *A, const oracle_smoother *s, double *x, const double *b);
void oracle_cycle(oracle_hier *h, int lvl, double *x, const double *b, int cycle);
/* returns the number of residuals written (<= maxiter+1) */
int oracle_solve(oracle_hier *h, const double *b, double *x, double tol, int maxiter, int cycle,
                 double *residuals);

#ifdef __cplusplus
}
#endif
#endif
