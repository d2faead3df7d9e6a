// Flat amg_core drop-ins on HOST pointers (include/amgcore_hip.h, section 1).
// Each call stages its operands in HBM, runs the same HIP kernels the resident
// hierarchy uses, and copies the mutated vectors back -- the calling convention
// of /root/reference/pyamg/amg_core (numpy arrays in, results in place).
#include <algorithm>
#include "hier.hpp"

#include <cstdlib>
#include <cstring>

namespace amg {
int upload_csr(DevCsr &M, int nrows, int ncols, const int *Ap, const int *Aj, const double *Ax, long *acct);
int upload_bsr(DevBsr &M, int nbrows, int bs, const int *Ap, const int *Aj, const double *Ax, long *acct);
int spmv(const DevCsr &M, StreamMode mode, const double *xg, const double *b, const double *v2,
         double *out, double *out2, double c0, hipStream_t st);
void expand_bsr(int nbrows, int R, int C, const int *Ap, const int *Aj, const double *Ax,
                std::vector<int> &cp, std::vector<int> &cj, std::vector<double> &cx);
}
namespace amg {
int gs_sweep_csr(const Schedule &S, bool bsr1, double *x, const double *b, bool reverse, hipStream_t st, bool allow_flow = true);
}
using namespace amg;

#define CHK(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

namespace {

// RAII device buffer
struct DBuf {
    void *p = nullptr;
    ~DBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes + 128);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        return 0;
    }
    int from_host(const void *src, size_t bytes)
    {
        CHK(alloc(bytes));
        if (bytes) AMG_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    int to_host(void *dst, size_t bytes) const
    {
        if (bytes) AMG_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    double *d() const { return (double *)p; }
    int *i() const { return (int *)p; }
};

struct CsrHolder {
    DevCsr M;
    ~CsrHolder()
    {
        if (M.Ap) hipFree(M.Ap);
        if (M.Aj) hipFree(M.Aj);
        if (M.Ax) hipFree(M.Ax);
    }
};
struct BsrHolder {
    DevBsr M;
    ~BsrHolder()
    {
        if (M.Ap) hipFree(M.Ap);
        if (M.Aj) hipFree(M.Aj);
        if (M.Ax) hipFree(M.Ax);
    }
};
struct SchedHolder {
    Schedule S;
    ~SchedHolder() { S.release(); }
};

int require_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available (amgcore_hip has no CPU fallback)");
        return AMG_ENODEV;
    }
    return 0;
}

// the rows visited by for(i = start; i != stop; i += step)
int sweep_rows(int start, int stop, int step, int limit, std::vector<int> &rows)
{
    rows.clear();
    if (step == 0) { set_error("row_step == 0"); return AMG_EINVAL; }
    long span = (long)stop - start;
    if (span == 0) return 0;
    if (span % step != 0 || span / step < 0) {
        set_error("row_start/row_stop/row_step never terminate");
        return AMG_EINVAL;
    }
    long cnt = span / step;
    rows.resize((size_t)cnt);
    for (long t = 0; t < cnt; ++t) {
        long i = start + t * step;
        if (i < 0 || i >= limit) { set_error("sweep leaves the matrix"); return AMG_EINVAL; }
        rows[(size_t)t] = (int)i;
    }
    return 0;
}

int check_csr(const int *Ap, int Ap_size, int Aj_size, int Ax_size, int per_entry)
{
    if (!Ap || Ap_size < 1) { set_error("bad Ap"); return AMG_EINVAL; }
    long nnz = Ap[Ap_size - 1];
    if (nnz < 0 || nnz > Aj_size || nnz * per_entry > Ax_size) {
        set_error("Aj/Ax shorter than Ap[-1]");
        return AMG_EINVAL;
    }
    return 0;
}

int run_csr_levels(const Schedule &S, bool bsr1, double *x, const double *b)
{
    // the schedule lists the tasks in the caller's sweep order: always walked forward
    return amg::gs_sweep_csr(S, bsr1, x, b, false, nullptr);
}

int gs_csr_common(const int *Ap, int Ap_size, const int *Aj, const double *Ax, double *x, int x_size,
                  const double *b, int b_size, const std::vector<int> &tasks, bool bsr1)
{
    const int n = Ap_size - 1;
    if (tasks.empty()) return 0;
    SchedHolder sh;
    CHK(build_csr_schedule(Ap, Aj, Ax, n, tasks.data(), (int)tasks.size(), sh.S, nullptr));
    DBuf dx, db;
    CHK(dx.from_host(x, sizeof(double) * (size_t)x_size));
    CHK(db.from_host(b, sizeof(double) * (size_t)b_size));
    CHK(run_csr_levels(sh.S, bsr1, dx.d(), db.d()));
    AMG_HIP(hipDeviceSynchronize());
    if (gs_flow_status() != 0) { set_error("a dataflow Gauss-Seidel sweep ran out of its time budget"); return AMG_ESTATE; }
    return dx.to_host(x, sizeof(double) * (size_t)x_size);
}

}  // namespace

extern "C" {

int amgcore_gauss_seidel_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                             const double Ax[], int Ax_size, double x[], int x_size,
                             const double b[], int b_size, int row_start, int row_stop, int row_step)
{
    CHK(require_device());
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, 1));
    std::vector<int> tasks;
    CHK(sweep_rows(row_start, row_stop, row_step, std::min(Ap_size - 1, std::min(x_size, b_size)), tasks));
    return gs_csr_common(Ap, Ap_size, Aj, Ax, x, x_size, b, b_size, tasks, false);
}

int amgcore_gauss_seidel_indexed_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                     const double Ax[], int Ax_size, double x[], int x_size,
                                     const double b[], int b_size, const int Id[], int Id_size,
                                     int row_start, int row_stop, int row_step)
{
    CHK(require_device());
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, 1));
    std::vector<int> pos, tasks;
    CHK(sweep_rows(row_start, row_stop, row_step, Id_size, pos));
    tasks.resize(pos.size());
    const int n = std::min(Ap_size - 1, std::min(x_size, b_size));
    for (size_t t = 0; t < pos.size(); ++t) {
        tasks[t] = Id[pos[t]];
        if (tasks[t] < 0 || tasks[t] >= n) { set_error("Id entry out of range"); return AMG_EINVAL; }
    }
    return gs_csr_common(Ap, Ap_size, Aj, Ax, x, x_size, b, b_size, tasks, false);
}

int amgcore_bsr_gauss_seidel_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                 const double Ax[], int Ax_size, double x[], int x_size,
                                 const double b[], int b_size, int row_start, int row_stop,
                                 int row_step, int blocksize)
{
    CHK(require_device());
    if (blocksize < 1) { set_error("blocksize < 1"); return AMG_EINVAL; }
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, blocksize * blocksize));
    const int nb = Ap_size - 1;
    std::vector<int> tasks;
    CHK(sweep_rows(row_start, row_stop, row_step, std::min(nb, std::min(x_size, b_size) / blocksize), tasks));
    if (tasks.empty()) return 0;
    if (blocksize == 1) return gs_csr_common(Ap, Ap_size, Aj, Ax, x, x_size, b, b_size, tasks, true);
    SchedHolder sh;
    CHK(build_block_schedule(Ap, Aj, nb, tasks.data(), (int)tasks.size(), sh.S, nullptr, Ax, blocksize));
    DBuf dx, db;
    CHK(dx.from_host(x, sizeof(double) * (size_t)x_size));
    CHK(db.from_host(b, sizeof(double) * (size_t)b_size));
    // the task list is already in sweep order: levels ascending; a backward sweep reverses the order inside a block
    {
        Schedule &S = sh.S;
        BsrStreamArgs a;
        std::memset(&a, 0, sizeof(a));
        a.Ap = S.Gb.Ap; a.Aj = S.Gb.Aj; a.Ax = S.Gb.Ax; a.bs = blocksize; a.rowmap = S.rows;
        a.intra_reverse = row_step < 0 ? 1 : 0;
        a.xin = dx.d(); a.xout = dx.d(); a.b = db.d(); a.omega = 1.0;
        for (int l = 0; l < S.nlevels(); ++l) {
            a.brow_lo = S.level_ptr[l]; a.brow_hi = S.level_ptr[l + 1];
            CHK(launch_bsr_stream(BM_BSR_GS, a, 0, nullptr));
        }
    }
    AMG_HIP(hipDeviceSynchronize());
    return dx.to_host(x, sizeof(double) * (size_t)x_size);
}

int amgcore_jacobi_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size, const double Ax[],
                       int Ax_size, double x[], int x_size, const double b[], int b_size,
                       double temp[], int temp_size, int row_start, int row_stop, int row_step,
                       const double omega[], int omega_size)
{
    CHK(require_device());
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, 1));
    if (omega_size < 1 || !omega) { set_error("omega must be a length-1 array"); return AMG_EINVAL; }
    const int n = Ap_size - 1;
    std::vector<int> rows;
    CHK(sweep_rows(row_start, row_stop, row_step, std::min(n, std::min(std::min(x_size, b_size), temp_size)), rows));
    if (rows.empty()) return 0;
    CsrHolder ch;
    CHK(upload_csr(ch.M, n, n, Ap, Aj, Ax, nullptr));
    DBuf dx, db, dt;
    CHK(dx.from_host(x, sizeof(double) * (size_t)x_size));
    CHK(db.from_host(b, sizeof(double) * (size_t)b_size));
    CHK(dt.from_host(temp, sizeof(double) * (size_t)temp_size));
    const int count = (int)rows.size();
    CHK(launch_copy_strided(dt.d(), dx.d(), row_start, count, row_step, nullptr));   // relaxation.h:216-218
    if (row_step == 1 || row_step == -1) {
        StreamArgs a;
        std::memset(&a, 0, sizeof(a));
        a.Ap = ch.M.Ap; a.Aj = ch.M.Aj; a.Ax = ch.M.Ax; a.nnz_total = ch.M.nnz;
        a.rows_per_wg = rows_per_wg_for(ch.M.nnz, ch.M.nrows);
        a.row_lo = (row_step == 1) ? row_start : row_stop + 1;
        a.row_hi = (row_step == 1) ? row_stop : row_start + 1;
        a.xg = dt.d(); a.v2 = dt.d(); a.b = db.d(); a.out = dx.d(); a.c0 = omega[0];
        CHK(launch_stream(SM_JACOBI, a, nullptr));
    } else {
        CHK(launch_jacobi_rows(ch.M, dt.d(), db.d(), dx.d(), row_start, count, row_step, omega[0], nullptr));
    }
    AMG_HIP(hipDeviceSynchronize());
    CHK(dx.to_host(x, sizeof(double) * (size_t)x_size));
    return dt.to_host(temp, sizeof(double) * (size_t)temp_size);
}

int amgcore_bsr_jacobi_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                           const double Ax[], int Ax_size, double x[], int x_size, const double b[],
                           int b_size, double temp[], int temp_size, int row_start, int row_stop,
                           int row_step, int blocksize, const double omega[], int omega_size)
{
    CHK(require_device());
    if (blocksize < 1) { set_error("blocksize < 1"); return AMG_EINVAL; }
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, blocksize * blocksize));
    if (omega_size < 1 || !omega) { set_error("omega must be a length-1 array"); return AMG_EINVAL; }
    if (row_step < 0) {
        // relaxation.h:303-305 never terminates for a negative step
        set_error("bsr_jacobi: backward sweeps are not defined by the reference");
        return AMG_EINVAL;
    }
    const int nb = Ap_size - 1;
    std::vector<int> rows;
    CHK(sweep_rows(row_start, row_stop, row_step, std::min(nb, std::min(std::min(x_size, b_size), temp_size) / blocksize), rows));
    if (rows.empty()) return 0;
    SchedHolder sh;          // the listed block rows, copied in list order: one streamed slice
    CHK(build_block_schedule(Ap, Aj, nb, rows.data(), (int)rows.size(), sh.S, nullptr, Ax, blocksize, true));
    DBuf dx, db, dt;
    CHK(dx.from_host(x, sizeof(double) * (size_t)x_size));
    CHK(db.from_host(b, sizeof(double) * (size_t)b_size));
    CHK(dt.from_host(temp, sizeof(double) * (size_t)temp_size));
    long ncopy = (long)std::abs(row_stop - row_start) * blocksize;     // relaxation.h:303-305
    if (ncopy > std::min(x_size, temp_size)) { set_error("temp/x too short"); return AMG_EINVAL; }
    AMG_HIP(hipMemcpy(dt.p, dx.p, sizeof(double) * (size_t)ncopy, hipMemcpyDeviceToDevice));
    CHK(sweep_block_schedule(sh.S, BM_BSR_JACOBI, nullptr, dt.d(), dx.d(), db.d(), omega[0], false, nullptr));
    AMG_HIP(hipDeviceSynchronize());
    CHK(dx.to_host(x, sizeof(double) * (size_t)x_size));
    return dt.to_host(temp, sizeof(double) * (size_t)temp_size);
}

int amgcore_block_jacobi_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                             const double Ax[], int Ax_size, double x[], int x_size, const double b[],
                             int b_size, const double Tx[], int Tx_size, double temp[], int temp_size,
                             int row_start, int row_stop, int row_step, const double omega[],
                             int omega_size, int blocksize)
{
    CHK(require_device());
    if (blocksize < 1) { set_error("blocksize < 1"); return AMG_EINVAL; }
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, blocksize * blocksize));
    if (omega_size < 1 || !omega) { set_error("omega must be a length-1 array"); return AMG_EINVAL; }
    const int nb = Ap_size - 1;
    std::vector<int> rows;
    CHK(sweep_rows(row_start, row_stop, row_step, std::min(nb, std::min(std::min(x_size, b_size), temp_size) / blocksize), rows));
    if (rows.empty()) return 0;
    if ((long)nb * blocksize * blocksize > Tx_size) { set_error("Dinv too short"); return AMG_EINVAL; }
    SchedHolder sh;
    CHK(build_block_schedule(Ap, Aj, nb, rows.data(), (int)rows.size(), sh.S, nullptr, Ax, blocksize, true));
    DBuf dx, db, dt, dd, dr;
    CHK(dx.from_host(x, sizeof(double) * (size_t)x_size));
    CHK(db.from_host(b, sizeof(double) * (size_t)b_size));
    CHK(dt.from_host(temp, sizeof(double) * (size_t)temp_size));
    CHK(dd.from_host(Tx, sizeof(double) * (size_t)Tx_size));
    // relaxation.h:686-688: temp[i*bs..] = x[i*bs..] for the swept block rows
    if (row_step == 1 || row_step == -1) {
        // a contiguous range of block rows: one copy (not one per block row)
        const int lo = *std::min_element(rows.begin(), rows.end());
        AMG_HIP(hipMemcpyAsync(dt.d() + (long)lo * blocksize, dx.d() + (long)lo * blocksize,
                               sizeof(double) * (size_t)rows.size() * (size_t)blocksize, hipMemcpyDeviceToDevice, nullptr));
    } else {
        for (int r : rows)
            AMG_HIP(hipMemcpyAsync(dt.d() + (long)r * blocksize, dx.d() + (long)r * blocksize,
                                   sizeof(double) * (size_t)blocksize, hipMemcpyDeviceToDevice, nullptr));
    }
    CHK(sweep_block_schedule(sh.S, BM_BLOCK_JACOBI, dd.d(), dt.d(), dx.d(), db.d(), omega[0], false, nullptr));
    AMG_HIP(hipDeviceSynchronize());
    CHK(dx.to_host(x, sizeof(double) * (size_t)x_size));
    return dt.to_host(temp, sizeof(double) * (size_t)temp_size);
}

int amgcore_block_gauss_seidel_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                   const double Ax[], int Ax_size, double x[], int x_size,
                                   const double b[], int b_size, const double Tx[], int Tx_size,
                                   int row_start, int row_stop, int row_step, int blocksize)
{
    CHK(require_device());
    if (blocksize < 1) { set_error("blocksize < 1"); return AMG_EINVAL; }
    CHK(check_csr(Ap, Ap_size, Aj_size, Ax_size, blocksize * blocksize));
    const int nb = Ap_size - 1;
    std::vector<int> tasks;
    CHK(sweep_rows(row_start, row_stop, row_step, std::min(nb, std::min(x_size, b_size) / blocksize), tasks));
    if (tasks.empty()) return 0;
    if ((long)nb * blocksize * blocksize > Tx_size) { set_error("Dinv too short"); return AMG_EINVAL; }
    SchedHolder sh;
    CHK(build_block_schedule(Ap, Aj, nb, tasks.data(), (int)tasks.size(), sh.S, nullptr, Ax, blocksize, false, true));
    DBuf dx, db, dd;
    CHK(dx.from_host(x, sizeof(double) * (size_t)x_size));
    CHK(db.from_host(b, sizeof(double) * (size_t)b_size));
    CHK(dd.from_host(Tx, sizeof(double) * (size_t)Tx_size));
    const unsigned char fwd = 0;
    CHK(block_gs_sweeps(sh.S, dd.d(), dx.d(), db.d(), &fwd, 1, nullptr));
    AMG_HIP(hipDeviceSynchronize());
    if (gs_flow_status() != 0) { set_error("a dataflow block Gauss-Seidel sweep ran out of its time budget"); return AMG_ESTATE; }
    return dx.to_host(x, sizeof(double) * (size_t)x_size);
}

int amgcore_csr_matvec_f64(int n_row, int n_col, const int Ap[], const int Aj[], const double Ax[],
                           const double x[], double y[])
{
    CHK(require_device());
    if (n_row < 0 || n_col < 0 || !Ap) { set_error("bad csr_matvec arguments"); return AMG_EINVAL; }
    CsrHolder ch;
    CHK(upload_csr(ch.M, n_row, n_col, Ap, Aj, Ax, nullptr));
    DBuf dx, dy;
    CHK(dx.from_host(x, sizeof(double) * (size_t)n_col));
    CHK(dy.from_host(y, sizeof(double) * (size_t)n_row));
    CHK(spmv(ch.M, SM_MATVEC_ACC, dx.d(), nullptr, nullptr, dy.d(), nullptr, 0.0, nullptr));
    AMG_HIP(hipDeviceSynchronize());
    return dy.to_host(y, sizeof(double) * (size_t)n_row);
}

int amgcore_bsr_matvec_f64(int n_brow, int n_bcol, int R, int C, const int Ap[], const int Aj[],
                           const double Ax[], const double x[], double y[])
{
    CHK(require_device());
    if (n_brow < 0 || n_bcol < 0 || R < 1 || C < 1 || !Ap) { set_error("bad bsr_matvec arguments"); return AMG_EINVAL; }
    std::vector<int> cp, cj;
    std::vector<double> cx;
    expand_bsr(n_brow, R, C, Ap, Aj, Ax, cp, cj, cx);
    return amgcore_csr_matvec_f64(n_brow * R, n_bcol * C, cp.data(), cj.data(), cx.data(), x, y);
}

int amgcore_norm2_f64(const double x[], long n, double *result)
{
    CHK(require_device());
    if (n < 0 || !result) { set_error("bad norm arguments"); return AMG_EINVAL; }
    DBuf dx, ds;
    CHK(dx.from_host(x, sizeof(double) * (size_t)n));
    CHK(ds.alloc(sizeof(double) * 1040));
    CHK(launch_norm2(dx.d(), n, ds.d(), ds.d() + 1032, nullptr));
    AMG_HIP(hipDeviceSynchronize());
    AMG_HIP(hipMemcpy(result, ds.d() + 1032, sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
