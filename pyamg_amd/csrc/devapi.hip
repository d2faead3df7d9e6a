// Device-pointer API (include/amgcore_hip.h section 3): stand-alone operators in HBM and the
// vector kernels, on caller-supplied device pointers and stream.  This is what the
// row-partitioned multi-GPU driver (pyamg_amd/distributed.py) is built from: its vectors are
// torch tensors so that torch.distributed (RCCL) can move the halos.
#include "hier.hpp"

#include <cstring>

namespace amg {
int upload_csr(DevCsr &M, int nrows, int ncols, const int *Ap, const int *Aj, const double *Ax, long *acct);
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
    m->rpw = rows_per_wg_for(m->M.nnz, m->M.nrows);
    return m;
}

void amg_mat_destroy(amg_mat *m)
{
    if (!m) return;
    hipSetDevice(m->device);
    if (m->M.Ap) hipFree(m->M.Ap);
    if (m->M.Aj) hipFree(m->M.Aj);
    if (m->M.Ax) hipFree(m->M.Ax);
    delete m;
}

long amg_mat_nnz(amg_mat *m) { return m ? m->M.nnz : 0; }

// mode: 0 MATVEC, 1 MATVEC_ACC, 2 RESIDUAL, 3 POLY_FIRST, 4 POLY_STEP, 5 POLY_LAST, 6 JACOBI, 7 JACOBI_BSR1
int amg_mat_apply(amg_mat *m, int mode, const double *xg, const double *b, const double *v2, double *out,
                  double *out2, double c0, void *stream)
{
    if (!m) { set_error("null matrix"); return AMG_EINVAL; }
    if (mode < 0 || mode > SM_JACOBI_BSR1) { set_error("bad mode"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(m->device));
    StreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Ap = m->M.Ap; a.Aj = m->M.Aj; a.Ax = m->M.Ax;
    a.row_lo = 0; a.row_hi = m->M.nrows; a.nnz_total = m->M.nnz; a.rows_per_wg = m->rpw;
    a.xg = xg; a.b = b; a.v2 = v2; a.out = out; a.out2 = out2; a.c0 = c0;
    return launch_stream((StreamMode)mode, a, (hipStream_t)stream);
}

int amg_dev_scale(double *out, const double *in, double c, long n, void *stream)
{ return launch_scale(out, in, c, n, (hipStream_t)stream); }
int amg_dev_axpy(double *x, const double *h, long n, void *stream)
{ return launch_axpy_inplace(x, h, n, (hipStream_t)stream); }
// result_dev[0] = ||x||_2 of the LOCAL part; scratch >= 1040 doubles
int amg_dev_norm2(const double *x, long n, double *scratch, double *result_dev, void *stream)
{ return launch_norm2(x, n, scratch, result_dev, (hipStream_t)stream); }
int amg_dev_dot(const double *x, const double *y, long n, double *scratch, double *result_dev, void *stream)
{ return launch_dot(x, y, n, scratch, result_dev, (hipStream_t)stream); }
int amg_dev_dense_apply(const double *Mt, const double *b, double *x, int n, void *stream)
{ return launch_dense_apply(Mt, b, x, n, (hipStream_t)stream); }
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
