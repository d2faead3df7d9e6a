/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  See amg_oracle.h.
 *
 * Plain-C restatement of the reference's solve-phase arithmetic.  Compile with
 * -ffp-contract=off: the reference (g++, x86-64, no -mfma) never fuses a*b+c,
 * and bit-level agreement with it depends on that.
 *
 * All paths below are relative to /root/reference.
 */
#include "amg_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* amg_core kernels                                                          */
/* ------------------------------------------------------------------------- */

/* Row-parallel variant for the CPU baseline of bench.py: the loops whose iterations are independent
 * (one row of a product, one entry of a vector update) run over `oracle_threads` OpenMP threads.
 * Each row is still summed by one thread left to right, so the results do not depend on the
 * thread count; the Gauss-Seidel family and the norms stay sequential.  Default: 1 thread. */
static int oracle_threads = 1;
void oracle_set_threads(int n)
{
    oracle_threads = n < 1 ? 1 : n;
}
int oracle_get_threads(void) { return oracle_threads; }
#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(oracle_threads) if (oracle_threads > 1)")

/* pyamg/amg_core/relaxation.h:34-62 */
void oracle_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x,
                         const double *b, int row_start, int row_stop, int row_step)
{
    for (int i = row_start; i != row_stop; i += row_step) {
        double rsum = 0, diag = 0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j)
                diag = Ax[jj];
            else
                rsum += Ax[jj] * x[j];
        }
        if (diag != 0.0)
            x[i] = (b[i] - rsum) / diag;
    }
}

/* y = A*x for one dense row-major bs x bs block, y overwritten:
 * pyamg/amg_core/linalg.h:360-449 gemm(...,'F',...,'F',...,'F','T') with Bcols=1 */
static void block_gemv(const double *A, const double *x, double *y, int bs)
{
    for (int i = 0; i < bs; i++) {
        double s = 0.0;
        for (int k = 0; k < bs; k++)
            s += A[i * bs + k] * x[k];
        y[i] = s;
    }
}

/* pyamg/amg_core/relaxation.h:90-173 */
void oracle_bsr_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x,
                             const double *b, int row_start, int row_stop, int row_step,
                             int blocksize)
{
    int bs = blocksize, B2 = bs * bs;
    double *rsum = (double *)malloc(sizeof(double) * bs);
    double *Axloc = (double *)malloc(sizeof(double) * bs);
    int step, step_start, step_end;
    if (row_step < 0) { step = -1; step_start = bs - 1; step_end = -1; }
    else              { step = 1;  step_start = 0;      step_end = bs; }

    for (int i = row_start; i != row_stop; i += row_step) {
        long diag_ptr = -1;
        for (int k = 0; k < bs; k++)
            rsum[k] = b[(long)i * bs + k];
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j) {
                diag_ptr = (long)jj * B2;
            } else {
                block_gemv(Ax + (long)jj * B2, x + (long)j * bs, Axloc, bs);
                for (int m = 0; m < bs; m++)
                    rsum[m] -= Axloc[m];
            }
        }
        if (diag_ptr != -1) {
            for (int k = step_start; k != step_end; k += step) {
                double diag = 1.0;
                for (int kk = step_start; kk != step_end; kk += step) {
                    if (k == kk)
                        diag = Ax[k * bs + kk + diag_ptr];
                    else
                        rsum[k] -= Ax[k * bs + kk + diag_ptr] * x[(long)i * bs + kk];
                }
                if (diag != 0.0)
                    x[(long)i * bs + k] = rsum[k] / diag;
            }
        }
    }
    free(rsum);
    free(Axloc);
}

/* pyamg/amg_core/relaxation.h:202-239 */
void oracle_jacobi(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                   double *temp, int row_start, int row_stop, int row_step, const double *omega)
{
    double one = 1.0, omega2 = omega[0];
    for (int i = row_start; i != row_stop; i += row_step)
        temp[i] = x[i];
    if (oracle_threads > 1 && row_step == 1 && row_start <= row_stop) {
        /* every row reads temp only: independent */
        PAR_FOR
        for (int i = row_start; i < row_stop; i++) {
            double rsum = 0, diag = 0;
            for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                int j = Aj[jj];
                if (i == j)
                    diag = Ax[jj];
                else
                    rsum += Ax[jj] * temp[j];
            }
            if (diag != 0.0)
                x[i] = (one - omega2) * temp[i] + omega2 * ((b[i] - rsum) / diag);
        }
        return;
    }
    for (int i = row_start; i != row_stop; i += row_step) {
        double rsum = 0, diag = 0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j)
                diag = Ax[jj];
            else
                rsum += Ax[jj] * temp[j];
        }
        if (diag != 0.0)
            x[i] = (one - omega2) * temp[i] + omega2 * ((b[i] - rsum) / diag);
    }
}

/* pyamg/amg_core/relaxation.h:268-360 */
void oracle_bsr_jacobi(const int *Ap, const int *Aj, const double *Ax, double *x,
                       const double *b, double *temp, int row_start, int row_stop,
                       int row_step, int blocksize, const double *omega)
{
    int bs = blocksize, B2 = bs * bs;
    double *rsum = (double *)malloc(sizeof(double) * bs);
    double *Axloc = (double *)malloc(sizeof(double) * bs);
    double one = 1.0, omega2 = omega[0];
    int step, step_start, step_end;
    if (row_step < 0) { step = -1; step_start = bs - 1; step_end = -1; }
    else              { step = 1;  step_start = 0;      step_end = bs; }

    /* relaxation.h:303-305 copies the first |stop-start|*bs entries (the shim
     * always sweeps forward over all rows, relaxation.py:407-425) */
    if (step > 0)
        for (long i = 0; i < (long)abs(row_stop - row_start) * bs; i += step)
            temp[i] = x[i];

    for (int i = row_start; i != row_stop; i += row_step) {
        long diag_ptr = -1;
        for (int k = 0; k < bs; k++)
            rsum[k] = b[(long)i * bs + k];
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j) {
                diag_ptr = (long)jj * B2;
            } else {
                block_gemv(Ax + (long)jj * B2, temp + (long)j * bs, Axloc, bs);
                for (int m = 0; m < bs; m++)
                    rsum[m] -= Axloc[m];
            }
        }
        if (diag_ptr != -1) {
            for (int k = step_start; k != step_end; k += step) {
                double diag = 1.0;
                for (int kk = step_start; kk != step_end; kk += step) {
                    if (k == kk)
                        diag = Ax[k * bs + kk + diag_ptr];
                    else
                        rsum[k] -= Ax[k * bs + kk + diag_ptr] * temp[(long)i * bs + kk];
                }
                if (diag != 0.0)
                    x[(long)i * bs + k] =
                        (one - omega2) * temp[(long)i * bs + k] + omega2 * rsum[k] / diag;
            }
        }
    }
    free(rsum);
    free(Axloc);
}

/* pyamg/amg_core/relaxation.h:395-426 */
void oracle_gauss_seidel_indexed(const int *Ap, const int *Aj, const double *Ax, double *x,
                                 const double *b, const int *Id, int row_start, int row_stop,
                                 int row_step)
{
    for (int i = row_start; i != row_stop; i += row_step) {
        int inew = Id[i];
        double rsum = 0, diag = 0;
        for (int jj = Ap[inew]; jj < Ap[inew + 1]; ++jj) {
            int j = Aj[jj];
            if (inew == j)
                diag = Ax[jj];
            else
                rsum += Ax[jj] * x[j];
        }
        if (diag != 0.0)
            x[inew] = (b[inew] - rsum) / diag;
    }
}

/* pyamg/amg_core/relaxation.h:466-496 */
void oracle_jacobi_ne(const int *Ap, const int *Aj, const double *Ax, double *x,
                      const double *b, const double *Tx, double *temp, int row_start,
                      int row_stop, int row_step, const double *omega)
{
    (void)b;
    const double *delta = Tx;
    const double omega2 = omega[0];
    for (int i = row_start; i < row_stop; i += row_step)
        temp[i] = 0.0;
    for (int i = row_start; i < row_stop; i += row_step)
        for (int j = Ap[i]; j < Ap[i + 1]; j++)
            temp[Aj[j]] += omega2 * Ax[j] * delta[i];
    for (int i = row_start; i < row_stop; i += row_step)
        x[i] += temp[i];
}

/* pyamg/amg_core/relaxation.h:530-561 */
void oracle_gauss_seidel_ne(const int *Ap, const int *Aj, const double *Ax, double *x,
                            const double *b, int row_start, int row_stop, int row_step,
                            const double *Tx, double omega)
{
    const double *D_inv = Tx;
    for (int i = row_start; i != row_stop; i += row_step) {
        double delta = 0.0;
        for (int j = Ap[i]; j < Ap[i + 1]; j++)
            delta += Ax[j] * x[Aj[j]];
        delta = (b[i] - delta) * D_inv[i] * omega;
        for (int j = Ap[i]; j < Ap[i + 1]; j++)
            x[Aj[j]] += Ax[j] * delta;
    }
}

/* pyamg/amg_core/relaxation.h:595-631 */
void oracle_gauss_seidel_nr(const int *Ap, const int *Aj, const double *Ax, double *x,
                            double *z, int col_start, int col_stop, int col_step,
                            const double *Tx, double omega)
{
    const double *D_inv = Tx;
    double *r = z;
    for (int i = col_start; i != col_stop; i += col_step) {
        double delta = 0.0;
        for (int j = Ap[i]; j < Ap[i + 1]; j++)
            delta += Ax[j] * r[Aj[j]];
        delta *= (D_inv[i] * omega);
        x[i] += delta;
        for (int j = Ap[i]; j < Ap[i + 1]; j++)
            r[Aj[j]] -= delta * Ax[j];
    }
}

/* pyamg/amg_core/relaxation.h:662-728 */
void oracle_block_jacobi(const int *Ap, const int *Aj, const double *Ax, double *x,
                         const double *b, const double *Dinv, double *temp, int row_start,
                         int row_stop, int row_step, const double *omega, int blocksize)
{
    int bs = blocksize, bsq = bs * bs;
    double one = 1.0, omega2 = omega[0];
    double *rsum = (double *)malloc(sizeof(double) * bs);
    double *v = (double *)malloc(sizeof(double) * bs);

    for (long i = (long)row_start * bs; i != (long)row_stop * bs; i += (long)row_step * bs)
        memcpy(temp + i, x + i, sizeof(double) * bs);

    for (int i = row_start; i != row_stop; i += row_step) {
        for (int k = 0; k < bs; k++)
            rsum[k] = 0.0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j)
                continue;
            block_gemv(Ax + (long)jj * bsq, temp + (long)j * bs, v, bs);
            for (int k = 0; k < bs; k++)
                rsum[k] += v[k];
        }
        long ib = (long)i * bs;
        for (int k = 0; k < bs; k++)
            rsum[k] = b[ib + k] - rsum[k];
        block_gemv(Dinv + (long)i * bsq, rsum, v, bs);
        for (int k = 0; k < bs; k++)
            x[ib + k] = (one - omega2) * temp[ib + k] + omega2 * v[k];
    }
    free(v);
    free(rsum);
}

/* pyamg/amg_core/relaxation.h:756-810 */
void oracle_block_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x,
                               const double *b, const double *Dinv, int row_start,
                               int row_stop, int row_step, int blocksize)
{
    int bs = blocksize, bsq = bs * bs;
    double *rsum = (double *)malloc(sizeof(double) * bs);
    double *v = (double *)malloc(sizeof(double) * bs);
    for (int i = row_start; i != row_stop; i += row_step) {
        for (int k = 0; k < bs; k++)
            rsum[k] = 0.0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j)
                continue;
            block_gemv(Ax + (long)jj * bsq, x + (long)j * bs, v, bs);
            for (int k = 0; k < bs; k++)
                rsum[k] += v[k];
        }
        long ib = (long)i * bs;
        for (int k = 0; k < bs; k++)
            rsum[k] = b[ib + k] - rsum[k];
        block_gemv(Dinv + (long)i * bsq, rsum, x + ib, bs);
    }
    free(v);
    free(rsum);
}

/* pyamg/amg_core/relaxation.h:836-899 (helper of the Schwarz relaxation): dense diagonal block of A
 * for every subdomain, row-major, rows and columns in the subdomain's (sorted) index order.  Both
 * the row of A and the subdomain are sorted, so one merge per row finds the common columns. */
void oracle_extract_subblocks(const int *Ap, const int *Aj, const double *Ax, double *Tx,
                              const int *Tp, const int *Sj, const int *Sp, int nsdomains, int nrows)
{
    (void)nrows;
    for (int k = 0; k < Tp[nsdomains]; k++) Tx[k] = 0.0;
    for (int d = 0; d < nsdomains; d++) {
        int m = Sp[d + 1] - Sp[d];
        const int *S = Sj + Sp[d];
        for (int li = 0; li < m; li++) {
            int row = S[li];
            double *Trow = Tx + Tp[d] + (long)li * m;
            int lc = 0;
            for (int k = Ap[row]; k < Ap[row + 1] && lc < m; k++) {
                int col = Aj[k];
                while (lc < m && S[lc] < col) lc++;
                if (lc < m && S[lc] == col) { Trow[lc] = Ax[k]; lc++; }
            }
        }
    }
}

/* pyamg/amg_core/relaxation.h:935-1007: multiplicative overlapping Schwarz, subdomains visited as
 * for(d = row_start; d != row_stop; d += row_step).  Per subdomain: r_c = 0 - sum_jj Ax*x (left to
 * right), then + b[row]; delta = Tx_d * r with each entry summed from 0.0 left to right (gemm,
 * linalg.h:396-419); x[S] += delta. */
void oracle_overlapping_schwarz_csr(const int *Ap, const int *Aj, const double *Ax, double *x,
                                    const double *b, const double *Tx, const int *Tp, const int *Sj,
                                    const int *Sp, int nsdomains, int nrows, int row_start,
                                    int row_stop, int row_step)
{
    (void)nsdomains;
    double *r = (double *)malloc(sizeof(double) * (size_t)(nrows > 0 ? nrows : 1));
    double *delta = (double *)malloc(sizeof(double) * (size_t)(nrows > 0 ? nrows : 1));
    for (int d = row_start; d != row_stop; d += row_step) {
        int m = Sp[d + 1] - Sp[d];
        const int *S = Sj + Sp[d];
        for (int c = 0; c < m; c++) {
            int row = S[c];
            double acc = 0.0;
            for (int jj = Ap[row]; jj < Ap[row + 1]; jj++) acc -= Ax[jj] * x[Aj[jj]];
            acc += b[row];
            r[c] = acc;
        }
        const double *T = Tx + Tp[d];
        for (int i = 0; i < m; i++) {
            double acc = 0.0;
            for (int k = 0; k < m; k++) acc += T[(long)i * m + k] * r[k];
            delta[i] = acc;
        }
        for (int c = 0; c < m; c++) x[S[c]] += delta[c];
    }
    free(r);
    free(delta);
}

/* ------------------------------------------------------------------------- */
/* scipy.sparse._sparsetools (third party; scipy 1.15.3 in the image):        */
/* csr_matvec: per row, sum starts from y[i] and adds Ax[jj]*x[Aj[jj]] left   */
/* to right.  bsr_matvec: 1x1 blocks -> csr_matvec; otherwise per block row,  */
/* blocks in storage order, each block a row-major gemv accumulating into y.  */
/* Call sites: pyamg/multilevel.py:496,498,544,548; pyamg/util/linalg.py:112; */
/* pyamg/relaxation/relaxation.py:661,666.                                    */
/* ------------------------------------------------------------------------- */
void oracle_csr_matvec(int n_row, const int *Ap, const int *Aj, const double *Ax,
                       const double *x, double *y)
{
    PAR_FOR
    for (int i = 0; i < n_row; i++) {
        double sum = y[i];
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++)
            sum += Ax[jj] * x[Aj[jj]];
        y[i] = sum;
    }
}

void oracle_bsr_matvec(int n_brow, int R, int C, const int *Ap, const int *Aj,
                       const double *Ax, const double *x, double *y)
{
    if (R == 1 && C == 1) {
        oracle_csr_matvec(n_brow, Ap, Aj, Ax, x, y);
        return;
    }
    long RC = (long)R * C;
    PAR_FOR
    for (int i = 0; i < n_brow; i++) {
        double *yb = y + (long)R * i;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            const double *A = Ax + RC * jj;
            const double *xb = x + (long)C * Aj[jj];
            for (int bi = 0; bi < R; bi++) {
                double dot = yb[bi];
                for (int bj = 0; bj < C; bj++)
                    dot += A[(long)C * bi + bj] * xb[bj];
                yb[bi] = dot;
            }
        }
    }
}

/* pyamg/util/linalg.py:17-53: sqrt(inner(conj(x), x)).  np.inner is a BLAS
 * ddot whose summation order is unspecified; blocked pairwise summation is
 * used here (error ~ eps*log n, like a SIMD ddot; a strict sequential sum
 * would drift by ~eps*sqrt(n) at 1.25e8 entries). */
static double sumsq_pairwise(const double *x, long n)
{
    if (n <= 256) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        long i = 0;
        for (; i + 4 <= n; i += 4) {
            s0 += x[i] * x[i];
            s1 += x[i + 1] * x[i + 1];
            s2 += x[i + 2] * x[i + 2];
            s3 += x[i + 3] * x[i + 3];
        }
        for (; i < n; i++)
            s0 += x[i] * x[i];
        return (s0 + s1) + (s2 + s3);
    }
    long h = (n / 2) & ~3L;
    return sumsq_pairwise(x, h) + sumsq_pairwise(x + h, n - h);
}

double oracle_norm2(const double *x, long n)
{
    return sqrt(sumsq_pairwise(x, n));
}

/* ------------------------------------------------------------------------- */
/* hierarchy                                                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    oracle_mat A, P, R;
    int hasA, hasPR;
    oracle_smoother pre, post;
} oracle_level;

struct oracle_hier {
    int nlevels;
    oracle_level *lv;
    const double *coarse_dense;
    int coarse_n;
    int dup_prolong;
};

oracle_hier *oracle_hier_create(int nlevels)
{
    oracle_hier *h = (oracle_hier *)calloc(1, sizeof(*h));
    h->nlevels = nlevels;
    h->lv = (oracle_level *)calloc((size_t)nlevels, sizeof(oracle_level));
    return h;
}

void oracle_hier_destroy(oracle_hier *h)
{
    if (!h) return;
    free(h->lv);
    free(h);
}

void oracle_hier_set_A(oracle_hier *h, int lvl, const oracle_mat *A)
{
    h->lv[lvl].A = *A;
    h->lv[lvl].hasA = 1;
}

void oracle_hier_set_PR(oracle_hier *h, int lvl, const oracle_mat *P, const oracle_mat *R)
{
    h->lv[lvl].P = *P;
    h->lv[lvl].R = *R;
    h->lv[lvl].hasPR = 1;
}

void oracle_hier_set_smoothers(oracle_hier *h, int lvl, const oracle_smoother *pre,
                               const oracle_smoother *post)
{
    h->lv[lvl].pre = *pre;
    h->lv[lvl].post = *post;
}

void oracle_hier_set_coarse_dense(oracle_hier *h, const double *Pinv, int n)
{
    h->coarse_dense = Pinv;
    h->coarse_n = n;
}

void oracle_hier_set_duplicate_prolongation(oracle_hier *h, int on)
{
    h->dup_prolong = on;
}

/* y = M*x (fresh y), the scipy `A * x` operator */
static void mat_apply(const oracle_mat *M, const double *x, double *y)
{
    if (oracle_threads > 1) {                /* first touch of a fresh buffer: spread the page faults */
        int n = M->nrows;
        PAR_FOR
        for (int i = 0; i < n; i++) y[i] = 0.0;
    } else {
        memset(y, 0, sizeof(double) * (size_t)M->nrows);
    }
    if (M->fmt == ORACLE_FMT_CSR)
        oracle_csr_matvec(M->nrows, M->Ap, M->Aj, M->Ax, x, y);
    else
        oracle_bsr_matvec(M->nrows / M->R, M->R, M->C, M->Ap, M->Aj, M->Ax, x, y);
}

/* pyamg/relaxation/relaxation.py:280-354 */
static void relax_gauss_seidel(const oracle_mat *A, double *x, const double *b, int iterations,
                               int sweep)
{
    int bs = (A->fmt == ORACLE_FMT_CSR) ? 1 : A->R;
    int nb = A->nrows / bs;
    if (sweep == ORACLE_SWEEP_SYMMETRIC) {
        for (int it = 0; it < iterations; it++) {
            relax_gauss_seidel(A, x, b, 1, ORACLE_SWEEP_FORWARD);
            relax_gauss_seidel(A, x, b, 1, ORACLE_SWEEP_BACKWARD);
        }
        return;
    }
    int rs, re, rt;
    if (sweep == ORACLE_SWEEP_FORWARD) { rs = 0; re = nb; rt = 1; }
    else                               { rs = nb - 1; re = -1; rt = -1; }
    for (int it = 0; it < iterations; it++) {
        if (A->fmt == ORACLE_FMT_CSR)
            oracle_gauss_seidel(A->Ap, A->Aj, A->Ax, x, b, rs, re, rt);
        else
            oracle_bsr_gauss_seidel(A->Ap, A->Aj, A->Ax, x, b, rs, re, rt, bs);
    }
}

/* pyamg/relaxation/relaxation.py:357-427 */
static void relax_jacobi(const oracle_mat *A, double *x, const double *b, int iterations,
                         double omega)
{
    int n = A->nrows;
    if (n <= 0) return;
    double *temp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int it = 0; it < iterations; it++) {
        if (A->fmt == ORACLE_FMT_CSR)
            oracle_jacobi(A->Ap, A->Aj, A->Ax, x, b, temp, 0, n, 1, &omega);
        else
            oracle_bsr_jacobi(A->Ap, A->Aj, A->Ax, x, b, temp, 0, n / A->R, 1, A->R, &omega);
    }
    free(temp);
}

/* pyamg/relaxation/relaxation.py:108-169 */
static void relax_sor(const oracle_mat *A, double *x, const double *b, double omega,
                      int iterations, int sweep)
{
    int n = A->nrows;
    double *x_old = (double *)malloc(sizeof(double) * (size_t)n);
    for (int it = 0; it < iterations; it++) {
        memcpy(x_old, x, sizeof(double) * (size_t)n);
        relax_gauss_seidel(A, x, b, 1, sweep);
        for (int i = 0; i < n; i++) x[i] *= omega;
        for (int i = 0; i < n; i++) x_old[i] *= (1 - omega);
        for (int i = 0; i < n; i++) x[i] += x_old[i];
    }
    free(x_old);
}

/* pyamg/relaxation/relaxation.py:593-668 */
static void relax_polynomial(const oracle_mat *A, double *x, const double *b, const double *coef,
                             int ncoef, int iterations)
{
    int n = A->nrows;
    double *res = (double *)malloc(sizeof(double) * (size_t)n);
    double *h = (double *)malloc(sizeof(double) * (size_t)n);
    double *Ah = (double *)malloc(sizeof(double) * (size_t)n);
    for (int it = 0; it < iterations; it++) {
        int xzero = 1;
        for (int i = 0; i < n; i++)
            if (x[i] != 0.0) { xzero = 0; break; }   /* norm(x) == 0 */
        if (xzero) {
            memcpy(res, b, sizeof(double) * (size_t)n);
        } else {
            mat_apply(A, x, Ah);
            PAR_FOR
            for (int i = 0; i < n; i++) res[i] = b[i] - Ah[i];
        }
        PAR_FOR
        for (int i = 0; i < n; i++) h[i] = coef[0] * res[i];
        for (int c = 1; c < ncoef; c++) {
            mat_apply(A, h, Ah);
            PAR_FOR
            for (int i = 0; i < n; i++) h[i] = coef[c] * res[i] + Ah[i];
        }
        PAR_FOR
        for (int i = 0; i < n; i++) x[i] += h[i];
    }
    free(res);
    free(h);
    free(Ah);
}

/* pyamg/relaxation/relaxation.py:430-506 (A already in BSR(bs,bs): s->Aalt) */
static void relax_block_jacobi(const oracle_mat *A, const oracle_smoother *s, double *x,
                               const double *b)
{
    int bs = s->blocksize, nb = A->nrows / bs;
    if (nb <= 0) return;
    double omega = s->omega;
    double *temp = (double *)malloc(sizeof(double) * (size_t)A->nrows);
    for (int it = 0; it < s->iterations; it++)
        oracle_block_jacobi(A->Ap, A->Aj, A->Ax, x, b, s->Dinv, temp, 0, nb, 1, &omega, bs);
    free(temp);
}

/* pyamg/relaxation/relaxation.py:509-590 */
static void relax_block_gauss_seidel(const oracle_mat *A, const oracle_smoother *s, double *x,
                                     const double *b, int iterations, int sweep)
{
    int bs = s->blocksize, nb = A->nrows / bs;
    if (sweep == ORACLE_SWEEP_SYMMETRIC) {
        for (int it = 0; it < iterations; it++) {
            relax_block_gauss_seidel(A, s, x, b, 1, ORACLE_SWEEP_FORWARD);
            relax_block_gauss_seidel(A, s, x, b, 1, ORACLE_SWEEP_BACKWARD);
        }
        return;
    }
    int rs, re, rt;
    if (sweep == ORACLE_SWEEP_FORWARD) { rs = 0; re = nb; rt = 1; }
    else                               { rs = nb - 1; re = -1; rt = -1; }
    for (int it = 0; it < iterations; it++)
        oracle_block_gauss_seidel(A->Ap, A->Aj, A->Ax, x, b, s->Dinv, rs, re, rt, bs);
}

/* pyamg/relaxation/relaxation.py:671-741 */
static void relax_gs_indexed(const oracle_mat *A, const oracle_smoother *s, double *x,
                             const double *b, int iterations, int sweep)
{
    if (sweep == ORACLE_SWEEP_SYMMETRIC) {
        for (int it = 0; it < iterations; it++) {
            relax_gs_indexed(A, s, x, b, 1, ORACLE_SWEEP_FORWARD);
            relax_gs_indexed(A, s, x, b, 1, ORACLE_SWEEP_BACKWARD);
        }
        return;
    }
    int rs, re, rt;
    if (sweep == ORACLE_SWEEP_FORWARD) { rs = 0; re = s->nindices; rt = 1; }
    else                               { rs = s->nindices - 1; re = -1; rt = -1; }
    for (int it = 0; it < iterations; it++)
        oracle_gauss_seidel_indexed(A->Ap, A->Aj, A->Ax, x, b, s->indices, rs, re, rt);
}

/* pyamg/relaxation/relaxation.py:821-908 (A in CSR; Tx = 1/diag(A A^H)) */
static void relax_gs_ne(const oracle_mat *A, const oracle_smoother *s, double *x, const double *b,
                        int iterations, int sweep)
{
    if (sweep == ORACLE_SWEEP_SYMMETRIC) {
        for (int it = 0; it < iterations; it++) {
            relax_gs_ne(A, s, x, b, 1, ORACLE_SWEEP_FORWARD);
            relax_gs_ne(A, s, x, b, 1, ORACLE_SWEEP_BACKWARD);
        }
        return;
    }
    int n = A->nrows, rs, re, rt;
    if (sweep == ORACLE_SWEEP_FORWARD) { rs = 0; re = n; rt = 1; }
    else                               { rs = n - 1; re = -1; rt = -1; }
    for (int it = 0; it < iterations; it++)
        oracle_gauss_seidel_ne(A->Ap, A->Aj, A->Ax, x, b, rs, re, rt, s->Dinv, s->omega);
}

/* pyamg/relaxation/relaxation.py:911-997 (A in CSC; Tx = 1/diag(A^H A)) */
static void relax_gs_nr(const oracle_mat *Acsr, const oracle_mat *Acsc, const oracle_smoother *s,
                        double *x, const double *b, int iterations, int sweep)
{
    if (sweep == ORACLE_SWEEP_SYMMETRIC) {
        for (int it = 0; it < iterations; it++) {
            relax_gs_nr(Acsr, Acsc, s, x, b, 1, ORACLE_SWEEP_FORWARD);
            relax_gs_nr(Acsr, Acsc, s, x, b, 1, ORACLE_SWEEP_BACKWARD);
        }
        return;
    }
    int n = Acsc->ncols, m = Acsc->nrows, cs, ce, ct;
    if (sweep == ORACLE_SWEEP_FORWARD) { cs = 0; ce = n; ct = 1; }
    else                               { cs = n - 1; ce = -1; ct = -1; }
    double *r = (double *)malloc(sizeof(double) * (size_t)m);
    double *Ax_ = (double *)malloc(sizeof(double) * (size_t)m);
    /* r = b - A*x once, before the iteration loop (relaxation.py:992) with A in
     * CSC: scipy csc_matvec scatters column by column */
    memset(Ax_, 0, sizeof(double) * (size_t)m);
    for (int j = 0; j < n; j++)
        for (int k = Acsc->Ap[j]; k < Acsc->Ap[j + 1]; k++)
            Ax_[Acsc->Aj[k]] += Acsc->Ax[k] * x[j];
    for (int i = 0; i < m; i++) r[i] = b[i] - Ax_[i];
    for (int it = 0; it < iterations; it++)
        oracle_gauss_seidel_nr(Acsc->Ap, Acsc->Aj, Acsc->Ax, x, r, cs, ce, ct, s->Dinv, s->omega);
    (void)Acsr;
    free(r);
    free(Ax_);
}

/* pyamg/relaxation/relaxation.py:744-818 */
static void relax_jacobi_ne(const oracle_mat *A, const oracle_smoother *s, double *x,
                            const double *b)
{
    int n = A->nrows;
    double *rn = (double *)malloc(sizeof(double) * (size_t)n);
    double *Axv = (double *)malloc(sizeof(double) * (size_t)n);
    double *temp = (double *)malloc(sizeof(double) * (size_t)n);
    double omega = s->omega;
    for (int it = 0; it < s->iterations; it++) {
        /* delta = ravel(b - A*x) * Dinv  (relaxation.py:816) */
        mat_apply(A, x, Axv);
        for (int i = 0; i < n; i++) rn[i] = (b[i] - Axv[i]) * s->Dinv[i];
        oracle_jacobi_ne(A->Ap, A->Aj, A->Ax, x, b, rn, temp, 0, n, 1, &omega);
    }
    free(rn);
    free(Axv);
    free(temp);
}

/* dispatch of a smoother closure built by pyamg/relaxation/smoothing.py:320-515 */
/* pyamg/relaxation/relaxation.py:254-277 */
static void relax_schwarz(const oracle_mat *A, const oracle_smoother *s, double *x, const double *b)
{
    int nsd = s->nsdomains, n = A->nrows;
    for (int it = 0; it < s->iterations; it++) {
        if (s->sweep == ORACLE_SWEEP_FORWARD || s->sweep == ORACLE_SWEEP_SYMMETRIC)
            oracle_overlapping_schwarz_csr(A->Ap, A->Aj, A->Ax, x, b, s->Tx, s->Tp, s->Sj, s->Sp, nsd, n,
                                           0, nsd, 1);
        if (s->sweep == ORACLE_SWEEP_BACKWARD || s->sweep == ORACLE_SWEEP_SYMMETRIC)
            oracle_overlapping_schwarz_csr(A->Ap, A->Aj, A->Ax, x, b, s->Tx, s->Tp, s->Sj, s->Sp, nsd, n,
                                           nsd - 1, -1, -1);
    }
}

void oracle_relax(const oracle_mat *A, const oracle_smoother *s, double *x, const double *b)
{
    switch (s->kind) {
    case ORACLE_SM_NONE: break;
    case ORACLE_SM_JACOBI: relax_jacobi(A, x, b, s->iterations, s->omega); break;
    case ORACLE_SM_GAUSS_SEIDEL: relax_gauss_seidel(A, x, b, s->iterations, s->sweep); break;
    case ORACLE_SM_SOR: relax_sor(A, x, b, s->omega, s->iterations, s->sweep); break;
    case ORACLE_SM_POLYNOMIAL: relax_polynomial(A, x, b, s->coef, s->ncoef, s->iterations); break;
    case ORACLE_SM_BLOCK_JACOBI: relax_block_jacobi(s->Aalt ? s->Aalt : A, s, x, b); break;
    case ORACLE_SM_BLOCK_GAUSS_SEIDEL:
        relax_block_gauss_seidel(s->Aalt ? s->Aalt : A, s, x, b, s->iterations, s->sweep);
        break;
    case ORACLE_SM_GAUSS_SEIDEL_INDEXED:
        relax_gs_indexed(s->Aalt ? s->Aalt : A, s, x, b, s->iterations, s->sweep);
        break;
    case ORACLE_SM_GAUSS_SEIDEL_NE:
        relax_gs_ne(s->Aalt ? s->Aalt : A, s, x, b, s->iterations, s->sweep);
        break;
    case ORACLE_SM_GAUSS_SEIDEL_NR: relax_gs_nr(A, s->Aalt, s, x, b, s->iterations, s->sweep); break;
    case ORACLE_SM_JACOBI_NE: relax_jacobi_ne(s->Aalt ? s->Aalt : A, s, x, b); break;
    case ORACLE_SM_SCHWARZ: relax_schwarz(s->Aalt ? s->Aalt : A, s, x, b); break;
    default: abort();
    }
}

/* coarse_grid_solver('pinv2') after its first call: np.dot(self.P, b)
 * pyamg/multilevel.py:608-612, generic_solver :694-712 */
static void coarse_solve(oracle_hier *h, const oracle_mat *A, const double *b, double *x)
{
    int n = A->nrows;
    long nnz = (long)A->Ap[(A->fmt == ORACLE_FMT_CSR) ? n : n / A->R];
    if (nnz == 0 || !h->coarse_dense) {
        memset(x, 0, sizeof(double) * (size_t)n);
        return;
    }
    const double *M = h->coarse_dense;
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++)
            s += M[(long)i * n + k] * b[k];
        x[i] = s;
    }
}

static double dot_seq(const double *a, const double *b, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* pyamg/multilevel.py:473-548 */
void oracle_cycle(oracle_hier *h, int lvl, double *x, const double *b, int cycle)
{
    oracle_level *L = &h->lv[lvl];
    const oracle_mat *A = &L->A;
    int n = A->nrows, nc = L->R.nrows;

    oracle_relax(A, &L->pre, x, b);                        /* :494 */

    double *residual = (double *)malloc(sizeof(double) * (size_t)n);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    mat_apply(A, x, tmp);                                  /* :496 */
    PAR_FOR
    for (int i = 0; i < n; i++) residual[i] = b[i] - tmp[i];

    double *coarse_b = (double *)malloc(sizeof(double) * (size_t)nc);
    double *coarse_x = (double *)calloc((size_t)nc, sizeof(double));
    mat_apply(&L->R, residual, coarse_b);                  /* :498-499 */

    if (lvl == h->nlevels - 2) {
        coarse_solve(h, &h->lv[h->nlevels - 1].A, coarse_b, coarse_x);   /* :501-502 */
    } else if (cycle == ORACLE_CYCLE_V) {
        oracle_cycle(h, lvl + 1, coarse_x, coarse_b, ORACLE_CYCLE_V);
    } else if (cycle == ORACLE_CYCLE_W) {
        oracle_cycle(h, lvl + 1, coarse_x, coarse_b, cycle);
        oracle_cycle(h, lvl + 1, coarse_x, coarse_b, cycle);
    } else if (cycle == ORACLE_CYCLE_F) {
        oracle_cycle(h, lvl + 1, coarse_x, coarse_b, cycle);
        oracle_cycle(h, lvl + 1, coarse_x, coarse_b, ORACLE_CYCLE_V);
    } else { /* AMLI :512-540 */
        enum { nAMLI = 2 };
        const oracle_mat *Ac = &h->lv[lvl + 1].A;
        double *p[nAMLI];
        double beta[nAMLI][nAMLI];
        double *Ap_ = (double *)malloc(sizeof(double) * (size_t)nc);
        double *Apj = (double *)malloc(sizeof(double) * (size_t)nc);
        for (int k = 0; k < nAMLI; k++) {
            p[k] = (double *)malloc(sizeof(double) * (size_t)nc);
            for (int i = 0; i < nc; i++) p[k][i] = 1.0;
            oracle_cycle(h, lvl + 1, p[k], coarse_b, cycle);
            for (int j = 0; j < k; j++) {
                mat_apply(Ac, p[k], Ap_);
                mat_apply(Ac, p[j], Apj);
                beta[k][j] = dot_seq(p[j], Ap_, nc) / dot_seq(p[j], Apj, nc);
                for (int i = 0; i < nc; i++) p[k][i] -= beta[k][j] * p[j][i];
            }
            mat_apply(Ac, p[k], Ap_);
            double alpha = dot_seq(p[k], coarse_b, nc) / dot_seq(p[k], Ap_, nc);
            for (int i = 0; i < nc; i++) coarse_x[i] += alpha * p[k][i];
            for (int i = 0; i < nc; i++) coarse_b[i] -= alpha * Ap_[i];
        }
        for (int k = 0; k < nAMLI; k++) free(p[k]);
        free(Ap_);
        free(Apj);
    }

    mat_apply(&L->P, coarse_x, tmp);                       /* :544 */
    PAR_FOR
    for (int i = 0; i < n; i++) x[i] += tmp[i];
    oracle_relax(A, &L->post, x, b);                       /* :545 */
    if (h->dup_prolong)
        mat_apply(&L->P, coarse_x, tmp);                   /* :548 (result discarded) */

    free(residual);
    free(tmp);
    free(coarse_b);
    free(coarse_x);
}

static double residual_norm(const oracle_mat *A, const double *x, const double *b, double *w1,
                            double *w2)
{
    /* pyamg/util/linalg.py:109-112 */
    int n = A->nrows;
    mat_apply(A, x, w1);
    PAR_FOR
    for (int i = 0; i < n; i++) w2[i] = b[i] - w1[i];
    return oracle_norm2(w2, n);
}

/* pyamg/multilevel.py:316-471 (accel=None path) */
int oracle_solve(oracle_hier *h, const double *b, double *x, double tol, int maxiter, int cycle,
                 double *residuals)
{
    const oracle_mat *A = &h->lv[0].A;
    int n = A->nrows, nres = 0;
    double *w1 = (double *)malloc(sizeof(double) * (size_t)n);
    double *w2 = (double *)malloc(sizeof(double) * (size_t)n);

    double normb = oracle_norm2(b, n);                     /* :427-429 */
    if (normb != 0.0) tol = tol * normb;

    residuals[nres++] = residual_norm(A, x, b, w1, w2);    /* :450 */
    while (nres <= maxiter && residuals[nres - 1] > tol) { /* :454 */
        if (h->nlevels == 1)
            coarse_solve(h, A, b, x);                      /* :455-457 */
        else
            oracle_cycle(h, 0, x, b, cycle);               /* :459 */
        residuals[nres++] = residual_norm(A, x, b, w1, w2); /* :461 */
    }
    free(w1);
    free(w2);
    return nres;
}
