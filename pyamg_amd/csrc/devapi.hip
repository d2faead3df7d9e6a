// Device-pointer API (include/amgcore_hip.h section 3): stand-alone operators in HBM and the
// vector kernels, on caller-supplied device pointers and stream.  This is what the
// row-partitioned multi-GPU driver (pyamg_amd/distributed.py) is built from: its vectors are
// torch tensors so that torch.distributed (RCCL) can move the halos.
#include "hier.hpp"

#include <cstdlib>
#include <cstring>

namespace amg {
int upload_csr(DevCsr &M, int nrows, int ncols, const int *Ap, const int *Aj, const double *Ax, long *acct);
int gs_sweep_csr(const Schedule &S, bool bsr1, double *x, const double *b, bool reverse, hipStream_t st, bool allow_flow = true);
int gs_sweep_csr(const Schedule &S, bool bsr1, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st, bool allow_flow = true);
int try_patterns(DevCsr &M, const int *Ap, const int *Aj, long *acct);
int apply_operator(const DevCsr &M, StreamMode mode, const StreamArgs &a, hipStream_t st);
void free_csr(DevCsr &M);
int build_index16(DevCsr &M, const int *Ap_host, long *acct);
}
using namespace amg;

#define CHK(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

struct amg_mat {
    int device = 0;
    DevCsr M;
    int rpw = 256;
    Schedule *sched = nullptr;     // Gauss-Seidel over the local rows (amg_mat_build_gs)
};

__global__ void gather_kernel(double *out, const double *in, const int *idx, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = in[idx[i]];
}

extern "C" {

amg_mat *amg_mat_create(int device, int nrows, int ncols, const int *Ap, const int *Aj, const double *Ax)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (amgcore_hip has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) { set_error("bad device"); return nullptr; }
    if (nrows < 0 || ncols < 0 || !Ap) { set_error("bad matrix"); return nullptr; }
    amg_mat *m = new amg_mat();
    m->device = device;
    if (upload_csr(m->M, nrows, ncols, Ap, Aj, Ax, nullptr) != 0) { delete m; return nullptr; }
    // structured-grid operators (also rank-local ones with a halo) get the pattern / stencil forms
    if (Aj && try_patterns(m->M, Ap, Aj, nullptr) != 0) { free_csr(m->M); delete m; return nullptr; }
    if (Aj && build_index16(m->M, Ap, nullptr) != 0) { free_csr(m->M); delete m; return nullptr; }
    m->rpw = rows_per_wg_for(m->M.nnz, m->M.nrows);
    return m;
}

int amg_mat_form(amg_mat *m)
{
    if (!m) return -1;
    if (m->M.st_vals && stencil_enabled()) return 2;
    return m->M.pat ? 1 : 0;
}

void amg_mat_destroy(amg_mat *m)
{
    if (!m) return;
    hipSetDevice(m->device);
    free_csr(m->M);
    if (m->sched) { m->sched->release(); delete m->sched; }
    delete m;
}

// Dependency-level schedule for Gauss-Seidel over this operator's rows in the given order
// (order == NULL: rows 0..nrows-1).  Columns >= nrows (the halo of a partitioned operator) are
// operands that no local row writes: they stay frozen during the sweep -- "GS inside the rank,
// Jacobi across ranks" (configuration C4).
int amg_mat_build_gs(amg_mat *m, const int *order, int norder)
{
    if (!m) { set_error("null matrix"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(m->device));
    const int n = m->M.nrows;
    std::vector<int> ap((size_t)n + 1), aj((size_t)m->M.nnz);
    std::vector<double> ax((size_t)m->M.nnz);
    AMG_HIP(hipMemcpy(ap.data(), m->M.Ap, sizeof(int) * ap.size(), hipMemcpyDeviceToHost));
    if (m->M.nnz) {
        AMG_HIP(hipMemcpy(aj.data(), m->M.Aj, sizeof(int) * aj.size(), hipMemcpyDeviceToHost));
        AMG_HIP(hipMemcpy(ax.data(), m->M.Ax, sizeof(double) * ax.size(), hipMemcpyDeviceToHost));
    }
    if (m->sched) { m->sched->release(); delete m->sched; m->sched = nullptr; }
    m->sched = new Schedule();
    // halo columns (ncols > nrows) mean a rank of a partitioned solve: dataflow sweeps only where ranks do not share a device
    const char *df = std::getenv("AMG_DIST_FLOW");
    const bool allow_flow = m->M.ncols == n || (df && std::atoi(df) != 0);
    int rc = build_csr_schedule(ap.data(), aj.data(), ax.data(), n, order, order ? norder : n, *m->sched, nullptr, allow_flow, m->M.ncols);
    if (rc != 0) { m->sched->release(); delete m->sched; m->sched = nullptr; }
    return rc;
}

int amg_mat_gs_levels(amg_mat *m) { return (m && m->sched) ? m->sched->nlevels() : 0; }

// one directional sweep (reverse != 0: the reversed order); bsr1 selects bsr_gauss_seidel's rounding
int amg_mat_gs_sweep(amg_mat *m, double *x, const double *b, int reverse, int bsr1, void *stream)
{
    if (!m || !m->sched) { set_error("amg_mat_build_gs was not called"); return AMG_ESTATE; }
    AMG_HIP(hipSetDevice(m->device));
    return gs_sweep_csr(*m->sched, bsr1 != 0, x, b, reverse != 0, (hipStream_t)stream);
}

int amg_mat_gs_sweeps(amg_mat *m, double *x, const double *b, const unsigned char *seq, int nseq, int bsr1, void *stream)
{
    if (!m || !m->sched) { set_error("amg_mat_build_gs was not called"); return AMG_ESTATE; }
    if (nseq < 0 || (nseq > 0 && !seq)) { set_error("bad sweep sequence"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(m->device));
    return gs_sweep_csr(*m->sched, bsr1 != 0, x, b, seq, nseq, (hipStream_t)stream);
}

long amg_mat_nnz(amg_mat *m) { return m ? m->M.nnz : 0; }

// mode: 0 MATVEC, 1 MATVEC_ACC, 2 RESIDUAL, 3 POLY_FIRST, 4 POLY_STEP, 5 POLY_LAST, 6 JACOBI, 7 JACOBI_BSR1
int amg_mat_apply(amg_mat *m, int mode, const double *xg, const double *b, const double *v2, double *out,
                  double *out2, double c0, double gscale, void *stream)
{
    if (!m) { set_error("null matrix"); return AMG_EINVAL; }
    if (mode < 0 || mode > SM_JACOBI_BSR1) { set_error("bad mode"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(m->device));
    StreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Ap = m->M.Ap; a.Aj = m->M.Aj; a.Ax = m->M.Ax;
    a.row_lo = 0; a.row_hi = m->M.nrows; a.nnz_total = m->M.nnz; a.rows_per_wg = m->rpw;
    if (m->M.Aj16 && m->M.i16_rpb == m->rpw) { a.Aj16 = m->M.Aj16; a.wg_base = m->M.wg_base; a.wg_flag = m->M.wg_flag; }
    a.xg = xg; a.b = b; a.v2 = v2; a.out = out; a.out2 = out2; a.c0 = c0; a.gscale = gscale;
    return apply_operator(m->M, (StreamMode)mode, a, (hipStream_t)stream);
}

// the same launch restricted to rows [row_lo, row_hi): lets a partitioned driver run the rows that
// read no halo entry while the halo exchange is still in flight
int amg_mat_apply_rows(amg_mat *m, int mode, int row_lo, int row_hi, const double *xg, const double *b,
                       const double *v2, double *out, double *out2, double c0, double gscale, void *stream)
{
    if (!m) { set_error("null matrix"); return AMG_EINVAL; }
    if (mode < 0 || mode > SM_JACOBI_BSR1) { set_error("bad mode"); return AMG_EINVAL; }
    if (row_lo < 0 || row_hi > m->M.nrows || row_lo > row_hi) { set_error("bad row range"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(m->device));
    StreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Ap = m->M.Ap; a.Aj = m->M.Aj; a.Ax = m->M.Ax;
    a.row_lo = row_lo; a.row_hi = row_hi; a.nnz_total = m->M.nnz; a.rows_per_wg = m->rpw;
    if (m->M.Aj16 && m->M.i16_rpb == m->rpw) { a.Aj16 = m->M.Aj16; a.wg_base = m->M.wg_base; a.wg_flag = m->M.wg_flag; }
    a.xg = xg; a.b = b; a.v2 = v2; a.out = out; a.out2 = out2; a.c0 = c0; a.gscale = gscale;
    return apply_operator(m->M, (StreamMode)mode, a, (hipStream_t)stream);
}

int amg_dev_scale(double *out, const double *in, double c, long n, void *stream)
{ return launch_scale(out, in, c, n, (hipStream_t)stream); }
int amg_dev_axpy(double *x, const double *h, long n, void *stream)
{ return launch_axpy_inplace(x, h, n, (hipStream_t)stream); }
int amg_dev_axpy_scaled(double *x, const double *r, double c, long n, void *stream)
{ return launch_axpy_scaled(x, r, c, n, (hipStream_t)stream); }
// result_dev[0] = ||x||_2 of the LOCAL part; scratch >= 1040 doubles
int amg_dev_norm2(const double *x, long n, double *scratch, double *result_dev, void *stream)
{ return launch_norm2(x, n, scratch, result_dev, (hipStream_t)stream); }
int amg_dev_dot(const double *x, const double *y, long n, double *scratch, double *result_dev, void *stream)
{ return launch_dot(x, y, n, scratch, result_dev, (hipStream_t)stream); }
int amg_dev_dense_apply(const double *Mt, const double *b, double *x, int n, void *stream)
{ return launch_dense_apply(Mt, b, x, n, (hipStream_t)stream); }
// ---- device-resident vectors for the Krylov methods that wrap the cycle (pyamg_amd/krylov.py): the vectors live in
// HBM, only scalars (inner products, norms, the few leading entries the Householder GMRES variants look at) cross PCIe
double *amg_dev_alloc(long n)
{
    double *p = nullptr;
    if (n < 0 || hipMalloc((void **)&p, sizeof(double) * (size_t)(n + PAD)) != hipSuccess) { set_error("device allocation failed"); return nullptr; }
    hipMemset(p, 0, sizeof(double) * (size_t)(n + PAD));
    hipDeviceSynchronize();
    return p;
}
void amg_dev_free(double *p) { if (p) hipFree(p); }
/* kind: 0 host->device, 1 device->host, 2 device->device; ordered on `stream`, host copies synchronous on return */
int amg_dev_copy(double *dst, const double *src, long n, int kind, void *stream)
{
    if (n <= 0) return 0;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    AMG_HIP(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, k, (hipStream_t)stream));
    if (kind != 2) AMG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int amg_dev_fill(double *x, double v, long n, void *stream) { return launch_fill(x, v, n, (hipStream_t)stream); }
int amg_dev_axmy(double *w, const double *v, double a, long n, void *stream)          /* w -= a v */
{ return launch_axmy(w, v, a, n, (hipStream_t)stream); }
int amg_dev_scale_add(double *p, double beta, const double *z, long n, void *stream)  /* p = beta p + z */
{ return launch_scale_add(p, beta, z, n, (hipStream_t)stream); }
int amg_dev_sub(double *out, const double *a, const double *b, long n, void *stream)  /* out = a - b */
{ return launch_sub(out, a, b, n, (hipStream_t)stream); }
int amg_dev_divide(double *w, double a, long n, void *stream)                          /* w /= a */
{ return launch_divide(w, a, n, (hipStream_t)stream); }
/* host scalar = <x, y> / ||x||_2 (deterministic two-stage reductions; one 8-byte read-back, stream synchronised) */
int amg_dev_dot_host(const double *x, const double *y, long n, double *scratch, double *result, void *stream)
{
    CHK(launch_dot(x, y, n, scratch, scratch + 1030, (hipStream_t)stream));
    AMG_HIP(hipMemcpyAsync(result, scratch + 1030, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
    AMG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int amg_dev_norm_host(const double *x, long n, double *scratch, double *result, void *stream)
{
    CHK(launch_norm2(x, n, scratch, scratch + 1030, (hipStream_t)stream));
    AMG_HIP(hipMemcpyAsync(result, scratch + 1030, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
    AMG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int amg_dev_gather(double *out, const double *in, const int *idx, long n, void *stream)
{
    if (n <= 0) return 0;
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(gather_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, out, in, idx, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gather", __FILE__, __LINE__);
    return 0;
}

}  // extern "C"
