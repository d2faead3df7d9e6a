// Normal-equation relaxation family (Kaczmarz / Cimmino type sweeps):
// amg_core.gauss_seidel_ne, gauss_seidel_nr, jacobi_ne
// (/root/reference/pyamg/amg_core/relaxation.h:465-631).
//
// gauss_seidel_ne/nr are sequential sweeps in which task i reads AND writes the
// vector entries listed in row (column) i.  Tasks that share no entry commute,
// so the sweep is executed by dependency levels (tasks touching a common entry
// keep their sequential order) -- bit-identical to the sequential loop.
// jacobi_ne scatters into temp; here each temp entry gathers its contributions
// through the transposed pattern in the same (row, position) order.
#include "hier.hpp"

#include <algorithm>
#include <cstring>

using namespace amg;

#define CHK(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

namespace {

struct DB {
    void *p = nullptr;
    ~DB() { if (p) hipFree(p); }
    int put(const void *src, size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes + 64);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        if (bytes) AMG_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    int get(void *dst, size_t bytes) { if (bytes) AMG_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost)); return 0; }
    double *d() { return (double *)p; }
    int *i() { return (int *)p; }
};

// level(t) = 1 + max level of earlier tasks touching a common entry
int touch_levels(int nvec, const int *Ap, const int *Aj, const std::vector<int> &tasks,
                 std::vector<int> &level_ptr, std::vector<int> &order)
{
    std::vector<int> last((size_t)nvec, 0), lvl(tasks.size());
    int maxl = 0;
    for (size_t t = 0; t < tasks.size(); ++t) {
        int i = tasks[t], l = 0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            int j = Aj[jj];
            if (j < 0 || j >= nvec) { set_error("index out of range"); return AMG_EINVAL; }
            l = std::max(l, last[j]);
        }
        l += 1;
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) last[Aj[jj]] = l;
        lvl[t] = l;
        maxl = std::max(maxl, l);
    }
    level_ptr.assign((size_t)maxl + 1, 0);
    for (size_t t = 0; t < tasks.size(); ++t) level_ptr[lvl[t]]++;
    int run = 0;
    for (int l = 1; l <= maxl; ++l) { int c = level_ptr[l]; level_ptr[l - 1] = run; run += c; }
    level_ptr[maxl] = run;
    order.resize(tasks.size());
    std::vector<int> cur(level_ptr.begin(), level_ptr.end() - 1);
    for (size_t t = 0; t < tasks.size(); ++t) order[cur[lvl[t] - 1]++] = tasks[t];
    return 0;
}

int sweep(int start, int stop, int step, int limit, std::vector<int> &rows)
{
    rows.clear();
    if (step == 0) { set_error("step == 0"); return AMG_EINVAL; }
    long span = (long)stop - start;
    if (span == 0) return 0;
    if (span % step != 0 || span / step < 0) { set_error("sweep never terminates"); return AMG_EINVAL; }
    for (long i = start; i != stop; i += step) {
        if (i < 0 || i >= limit) { set_error("sweep leaves the matrix"); return AMG_EINVAL; }
        rows.push_back((int)i);
    }
    return 0;
}

// relaxation.h:540-560
__global__ void gs_ne_level(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                            const double *Dinv, double omega, const int *rows, int count)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    int i = rows[t];
    int s = Ap[i], e = Ap[i + 1];
    double delta = 0.0;
    for (int j = s; j < e; ++j) delta = delta + Ax[j] * x[Aj[j]];
    delta = ((b[i] - delta) * Dinv[i]) * omega;
    for (int j = s; j < e; ++j) x[Aj[j]] = x[Aj[j]] + Ax[j] * delta;
}

// relaxation.h:605-630
__global__ void gs_nr_level(const int *Ap, const int *Aj, const double *Ax, double *x, double *r,
                            const double *Dinv, double omega, const int *cols, int count)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    int i = cols[t];
    int s = Ap[i], e = Ap[i + 1];
    double delta = 0.0;
    for (int j = s; j < e; ++j) delta = delta + Ax[j] * r[Aj[j]];
    delta = delta * (Dinv[i] * omega);
    x[i] = x[i] + delta;
    for (int j = s; j < e; ++j) r[Aj[j]] = r[Aj[j]] - delta * Ax[j];
}

// relaxation.h:481-495 through the transposed pattern
__global__ void jacobi_ne_gather(const int *Tp, const int *Trow, const double *Tval, const double *delta,
                                 double omega, double *temp, const unsigned char *in_range, int n)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double acc = (!in_range || in_range[c]) ? 0.0 : temp[c];      // null mask: every row is swept
    for (int k = Tp[c]; k < Tp[c + 1]; ++k) acc = acc + (omega * Tval[k]) * delta[Trow[k]];
    temp[c] = acc;
}
__global__ void jacobi_ne_update(double *x, const double *temp, const int *rows, int count)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    int i = rows[t];
    x[i] = x[i] + temp[i];
}

}  // namespace

// entry points for the in-cycle smoothers of the resident hierarchy (hier.hip)
namespace amg {
int ne_touch_levels(int nvec, const int *Ap, const int *Aj, int ntasks, std::vector<int> &level_ptr,
                    std::vector<int> &order)
{
    std::vector<int> tasks((size_t)ntasks);
    for (int i = 0; i < ntasks; ++i) tasks[i] = i;
    return touch_levels(nvec, Ap, Aj, tasks, level_ptr, order);
}
int launch_gs_ne_level(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, const double *Dinv,
                       double omega, const int *rows, int count, hipStream_t st)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(gs_ne_level, dim3((count + 127) / 128), dim3(128), 0, st, Ap, Aj, Ax, x, b, Dinv, omega, rows, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gs_ne level launch", __FILE__, __LINE__);
    return 0;
}
int launch_gs_nr_level(const int *Ap, const int *Aj, const double *Ax, double *x, double *r, const double *Dinv,
                       double omega, const int *cols, int count, hipStream_t st)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(gs_nr_level, dim3((count + 127) / 128), dim3(128), 0, st, Ap, Aj, Ax, x, r, Dinv, omega, cols, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gs_nr level launch", __FILE__, __LINE__);
    return 0;
}
// temp[c] = sum over column c of A, rows ascending, of (omega * a) * delta[row]  (relaxation.h:485-495)
int launch_jacobi_ne_gather(const int *Tp, const int *Trow, const double *Tval, const double *delta, double omega,
                            double *temp, int n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(jacobi_ne_gather, dim3((n + 127) / 128), dim3(128), 0, st, Tp, Trow, Tval, delta, omega, temp,
                       (const unsigned char *)nullptr, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "jacobi_ne gather launch", __FILE__, __LINE__);
    return 0;
}
}  // namespace amg

extern "C" {

int amgcore_gauss_seidel_ne_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                const double Ax[], int Ax_size, double x[], int x_size,
                                const double b[], int b_size, int row_start, int row_stop,
                                int row_step, const double Tx[], int Tx_size, double omega)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (amgcore_hip has no CPU fallback)"); return AMG_ENODEV; }
    if (!Ap || Ap_size < 1 || Ap[Ap_size - 1] > Aj_size || Ap[Ap_size - 1] > Ax_size) { set_error("bad CSR"); return AMG_EINVAL; }
    const int n = Ap_size - 1;
    std::vector<int> tasks, lp, order;
    CHK(sweep(row_start, row_stop, row_step, std::min(n, std::min(b_size, Tx_size)), tasks));
    if (tasks.empty()) return 0;
    CHK(touch_levels(x_size, Ap, Aj, tasks, lp, order));
    DB dAp, dAj, dAx, dx, db, dT, dord;
    CHK(dAp.put(Ap, sizeof(int) * (size_t)Ap_size));
    CHK(dAj.put(Aj, sizeof(int) * (size_t)Ap[n]));
    CHK(dAx.put(Ax, sizeof(double) * (size_t)Ap[n]));
    CHK(dx.put(x, sizeof(double) * (size_t)x_size));
    CHK(db.put(b, sizeof(double) * (size_t)b_size));
    CHK(dT.put(Tx, sizeof(double) * (size_t)Tx_size));
    CHK(dord.put(order.data(), sizeof(int) * order.size()));
    for (size_t l = 0; l + 1 < lp.size(); ++l) {
        int cnt = lp[l + 1] - lp[l];
        if (cnt <= 0) continue;
        hipLaunchKernelGGL(gs_ne_level, dim3((cnt + 127) / 128), dim3(128), 0, nullptr, dAp.i(), dAj.i(),
                           dAx.d(), dx.d(), db.d(), dT.d(), omega, dord.i() + lp[l], cnt);
    }
    AMG_HIP(hipGetLastError());
    AMG_HIP(hipDeviceSynchronize());
    return dx.get(x, sizeof(double) * (size_t)x_size);
}

int amgcore_gauss_seidel_nr_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                const double Ax[], int Ax_size, double x[], int x_size, double z[],
                                int z_size, int col_start, int col_stop, int col_step,
                                const double Tx[], int Tx_size, double omega)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (amgcore_hip has no CPU fallback)"); return AMG_ENODEV; }
    if (!Ap || Ap_size < 1 || Ap[Ap_size - 1] > Aj_size || Ap[Ap_size - 1] > Ax_size) { set_error("bad CSC"); return AMG_EINVAL; }
    const int n = Ap_size - 1;
    std::vector<int> tasks, lp, order;
    CHK(sweep(col_start, col_stop, col_step, std::min(n, std::min(x_size, Tx_size)), tasks));
    if (tasks.empty()) return 0;
    CHK(touch_levels(z_size, Ap, Aj, tasks, lp, order));
    DB dAp, dAj, dAx, dx, dz, dT, dord;
    CHK(dAp.put(Ap, sizeof(int) * (size_t)Ap_size));
    CHK(dAj.put(Aj, sizeof(int) * (size_t)Ap[n]));
    CHK(dAx.put(Ax, sizeof(double) * (size_t)Ap[n]));
    CHK(dx.put(x, sizeof(double) * (size_t)x_size));
    CHK(dz.put(z, sizeof(double) * (size_t)z_size));
    CHK(dT.put(Tx, sizeof(double) * (size_t)Tx_size));
    CHK(dord.put(order.data(), sizeof(int) * order.size()));
    for (size_t l = 0; l + 1 < lp.size(); ++l) {
        int cnt = lp[l + 1] - lp[l];
        if (cnt <= 0) continue;
        hipLaunchKernelGGL(gs_nr_level, dim3((cnt + 127) / 128), dim3(128), 0, nullptr, dAp.i(), dAj.i(),
                           dAx.d(), dx.d(), dz.d(), dT.d(), omega, dord.i() + lp[l], cnt);
    }
    AMG_HIP(hipGetLastError());
    AMG_HIP(hipDeviceSynchronize());
    CHK(dx.get(x, sizeof(double) * (size_t)x_size));
    return dz.get(z, sizeof(double) * (size_t)z_size);
}

int amgcore_jacobi_ne_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size, const double Ax[],
                          int Ax_size, double x[], int x_size, const double b[], int b_size,
                          const double Tx[], int Tx_size, double temp[], int temp_size,
                          int row_start, int row_stop, int row_step, const double omega[],
                          int omega_size)
{
    (void)b; (void)b_size;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (amgcore_hip has no CPU fallback)"); return AMG_ENODEV; }
    if (!Ap || Ap_size < 1 || Ap[Ap_size - 1] > Aj_size || Ap[Ap_size - 1] > Ax_size) { set_error("bad CSR"); return AMG_EINVAL; }
    if (omega_size < 1 || !omega) { set_error("omega must be a length-1 array"); return AMG_EINVAL; }
    if (row_step <= 0) { set_error("jacobi_ne: row_step must be positive (relaxation.h:481 uses '<')"); return AMG_EINVAL; }
    const int n = Ap_size - 1;
    std::vector<int> rows;
    for (long i = row_start; i < row_stop; i += row_step) {
        if (i < 0 || i >= n || i >= x_size || i >= temp_size || i >= Tx_size) { set_error("sweep leaves the matrix"); return AMG_EINVAL; }
        rows.push_back((int)i);
    }
    if (rows.empty()) return 0;
    // transposed pattern of the swept rows, contributions in (row, position) order
    std::vector<unsigned char> in_range((size_t)temp_size, 0);
    for (int i : rows) in_range[i] = 1;
    std::vector<int> Tp((size_t)temp_size + 1, 0);
    for (int i : rows)
        for (int j = Ap[i]; j < Ap[i + 1]; ++j) {
            if (Aj[j] < 0 || Aj[j] >= temp_size) { set_error("column out of range"); return AMG_EINVAL; }
            Tp[Aj[j] + 1]++;
        }
    for (int c = 0; c < temp_size; ++c) Tp[c + 1] += Tp[c];
    std::vector<int> Trow((size_t)Tp[temp_size]);
    std::vector<double> Tval((size_t)Tp[temp_size]);
    std::vector<int> cur(Tp.begin(), Tp.end() - 1);
    for (int i : rows)
        for (int j = Ap[i]; j < Ap[i + 1]; ++j) {
            int k = cur[Aj[j]]++;
            Trow[k] = i;
            Tval[k] = Ax[j];
        }
    DB dTp, dTr, dTv, dx, dtemp, ddelta, dmask, drows;
    CHK(dTp.put(Tp.data(), sizeof(int) * Tp.size()));
    CHK(dTr.put(Trow.data(), sizeof(int) * Trow.size()));
    CHK(dTv.put(Tval.data(), sizeof(double) * Tval.size()));
    CHK(dx.put(x, sizeof(double) * (size_t)x_size));
    CHK(dtemp.put(temp, sizeof(double) * (size_t)temp_size));
    CHK(ddelta.put(Tx, sizeof(double) * (size_t)Tx_size));
    CHK(dmask.put(in_range.data(), in_range.size()));
    CHK(drows.put(rows.data(), sizeof(int) * rows.size()));
    hipLaunchKernelGGL(jacobi_ne_gather, dim3((temp_size + 127) / 128), dim3(128), 0, nullptr, dTp.i(), dTr.i(),
                       dTv.d(), ddelta.d(), omega[0], dtemp.d(), (const unsigned char *)dmask.p, temp_size);
    int cnt = (int)rows.size();
    hipLaunchKernelGGL(jacobi_ne_update, dim3((cnt + 127) / 128), dim3(128), 0, nullptr, dx.d(), dtemp.d(),
                       drows.i(), cnt);
    AMG_HIP(hipGetLastError());
    AMG_HIP(hipDeviceSynchronize());
    CHK(dx.get(x, sizeof(double) * (size_t)x_size));
    return dtemp.get(temp, sizeof(double) * (size_t)temp_size);
}

}  // extern "C"
