// Overlapping multiplicative Schwarz relaxation: amg_core.overlapping_schwarz_csr
// (/root/reference/pyamg/amg_core/relaxation.h:935-1007; shim relaxation.py:172-278).
//
// The reference visits the subdomains one after the other: restricted residual, product with the
// stored (pseudo-)inverse of the subdomain's diagonal block, update of the subdomain's unknowns.
// Subdomain d READS x on every column of its rows and WRITES x on its own indices.  Two subdomains
// commute unless one writes what the other reads or writes, so the sweep is executed by dependency
// levels: level(d) = 1 + max level of the earlier subdomains it conflicts with.  Subdomains of one
// level run concurrently, one thread each, with the reference's own summation order inside --
// bit-identical to the sequential loop.  The conflict relation is symmetric, so the levels in
// descending order are the backward sweep.
#include "hier.hpp"

#include <algorithm>
#include <cstring>

using namespace amg;

#define CHK(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

namespace amg {

// levels of the subdomain tasks `tasks` (visit order): read-after-write, write-after-read and
// write-after-write all keep their order
int schwarz_levels(int nrows, const int *Ap, const int *Aj, const int *Sj, const int *Sp,
                   const std::vector<int> &tasks, std::vector<int> &level_ptr, std::vector<int> &order)
{
    std::vector<int> lastw((size_t)nrows, 0), lastr((size_t)nrows, 0), lvl(tasks.size());
    int maxl = 0;
    for (size_t t = 0; t < tasks.size(); ++t) {
        const int d = tasks[t];
        int l = 0;
        for (int j = Sp[d]; j < Sp[d + 1]; ++j) {
            const int row = Sj[j];
            if (row < 0 || row >= nrows) { set_error("subdomain index out of range"); return AMG_EINVAL; }
            l = std::max(l, std::max(lastw[row], lastr[row]));                       // we write x[row]
            for (int jj = Ap[row]; jj < Ap[row + 1]; ++jj) {
                const int c = Aj[jj];
                if (c < 0 || c >= nrows) { set_error("column index out of range"); return AMG_EINVAL; }
                l = std::max(l, lastw[c]);                                           // we read x[c]
            }
        }
        l += 1;
        for (int j = Sp[d]; j < Sp[d + 1]; ++j) {
            const int row = Sj[j];
            lastw[row] = l;
            for (int jj = Ap[row]; jj < Ap[row + 1]; ++jj) lastr[Aj[jj]] = std::max(lastr[Aj[jj]], l);
        }
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

// one dependency level; thread = subdomain.  r lives in scratch[Sp[d] .. Sp[d+1])
__global__ void schwarz_level_kernel(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                                     const double *Tx, const int *Tp, const int *Sj, const int *Sp,
                                     double *scratch, const int *doms, int count)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int d = doms[t];
    const int s0 = Sp[d], m = Sp[d + 1] - s0;
    double *r = scratch + s0;
    for (int c = 0; c < m; ++c) {                       // relaxation.h:969-981
        const int row = Sj[s0 + c];
        double acc = 0.0;
        for (int jj = Ap[row]; jj < Ap[row + 1]; ++jj) acc = acc - Ax[jj] * x[Aj[jj]];
        r[c] = acc + b[row];
    }
    const double *T = Tx + Tp[d];
    for (int i = 0; i < m; ++i) {                       // relaxation.h:984-993 (gemm from 0.0, left to right)
        double acc = 0.0;
        for (int k = 0; k < m; ++k) acc = acc + T[(long)i * m + k] * r[k];
        const int row = Sj[s0 + i];
        x[row] = x[row] + acc;
    }
}

int launch_schwarz_level(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                         const double *Tx, const int *Tp, const int *Sj, const int *Sp, double *scratch,
                         const int *doms, int count, hipStream_t st)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(schwarz_level_kernel, dim3((count + 63) / 64), dim3(64), 0, st, Ap, Aj, Ax, x, b, Tx, Tp, Sj,
                       Sp, scratch, doms, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "schwarz level launch", __FILE__, __LINE__);
    return 0;
}

}  // namespace amg

namespace {
struct DB {
    void *p = nullptr;
    ~DB() { if (p) hipFree(p); }
    int put(const void *src, size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes + 64);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        if (bytes && src) AMG_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    int get(void *dst, size_t bytes) { if (bytes) AMG_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost)); return 0; }
    double *d() { return (double *)p; }
    int *i() { return (int *)p; }
};
}  // namespace

extern "C" {

int amgcore_overlapping_schwarz_csr_f64(const int Ap[], int Ap_size, const int Aj[], int Aj_size,
                                        const double Ax[], int Ax_size, double x[], int x_size,
                                        const double b[], int b_size, const double Tx[], int Tx_size,
                                        const int Tp[], int Tp_size, const int Sj[], int Sj_size,
                                        const int Sp[], int Sp_size, int nsdomains, int nrows,
                                        int row_start, int row_stop, int row_step)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (amgcore_hip has no CPU fallback)"); return AMG_ENODEV; }
    if (!Ap || Ap_size < 1 || Ap[Ap_size - 1] > Aj_size || Ap[Ap_size - 1] > Ax_size) { set_error("bad CSR"); return AMG_EINVAL; }
    const int n = Ap_size - 1;
    if (nrows != n || x_size < n || b_size < n) { set_error("vector / matrix sizes disagree"); return AMG_EINVAL; }
    if (nsdomains < 0 || Sp_size < nsdomains + 1 || Tp_size < nsdomains + 1) { set_error("bad subdomain pointers"); return AMG_EINVAL; }
    if (nsdomains && (Sp[nsdomains] > Sj_size || Tp[nsdomains] > Tx_size)) { set_error("bad subdomain arrays"); return AMG_EINVAL; }
    if (row_step == 0) { set_error("row_step == 0"); return AMG_EINVAL; }
    std::vector<int> tasks;
    {
        long span = (long)row_stop - row_start;
        if (span != 0) {
            if (span % row_step != 0 || span / row_step < 0) { set_error("sweep never terminates"); return AMG_EINVAL; }
            for (long d = row_start; d != row_stop; d += row_step) {
                if (d < 0 || d >= nsdomains) { set_error("sweep leaves the subdomain list"); return AMG_EINVAL; }
                tasks.push_back((int)d);
            }
        }
    }
    if (tasks.empty()) return 0;
    for (int d : tasks)
        if ((long)(Sp[d + 1] - Sp[d]) * (Sp[d + 1] - Sp[d]) != (long)Tp[d + 1] - Tp[d]) { set_error("inverse block size does not match its subdomain"); return AMG_EINVAL; }
    std::vector<int> lp, order;
    CHK(schwarz_levels(n, Ap, Aj, Sj, Sp, tasks, lp, order));
    DB dAp, dAj, dAx, dx, db, dT, dTp, dSj, dSp, dord, dscr;
    CHK(dAp.put(Ap, sizeof(int) * (size_t)Ap_size));
    CHK(dAj.put(Aj, sizeof(int) * (size_t)Ap[n]));
    CHK(dAx.put(Ax, sizeof(double) * (size_t)Ap[n]));
    CHK(dx.put(x, sizeof(double) * (size_t)x_size));
    CHK(db.put(b, sizeof(double) * (size_t)b_size));
    CHK(dT.put(Tx, sizeof(double) * (size_t)Tp[nsdomains]));
    CHK(dTp.put(Tp, sizeof(int) * (size_t)(nsdomains + 1)));
    CHK(dSj.put(Sj, sizeof(int) * (size_t)Sp[nsdomains]));
    CHK(dSp.put(Sp, sizeof(int) * (size_t)(nsdomains + 1)));
    CHK(dord.put(order.data(), sizeof(int) * order.size()));
    CHK(dscr.put(nullptr, sizeof(double) * (size_t)Sp[nsdomains]));
    for (size_t l = 0; l + 1 < lp.size(); ++l)
        CHK(launch_schwarz_level(dAp.i(), dAj.i(), dAx.d(), dx.d(), db.d(), dT.d(), dTp.i(), dSj.i(), dSp.i(),
                                 dscr.d(), dord.i() + lp[l], lp[l + 1] - lp[l], nullptr));
    AMG_HIP(hipDeviceSynchronize());
    return dx.get(x, sizeof(double) * (size_t)x_size);
}

}  // extern "C"
