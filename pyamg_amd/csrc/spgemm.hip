// Galerkin products of the setup on the device: C = A * B for CSR operands with scipy's csr_matmat arithmetic and
// output order (scipy/sparse/sparsetools/csr.h csr_matmat, called by pyamg/aggregation/aggregation.py:425-426 as
// R * A * P) -- per output row the products are accumulated in the order (entry of A's row, entry of B's row), the
// output columns come out in REVERSE first-touch order, exact-zero results are dropped.
//
// Short rows: one thread per output row, a private open-addressing table in HBM per resident thread (keys, running
// sums, the insertion order); long rows of large levels: one wave per row, table in LDS (spgemm_wave_kernel).  The accesses of a row are sequential by construction -- that IS the summation order -- so there
// is nothing to share between lanes; what the GPU adds is ~10^5 rows in flight against the latency of the table
// accesses.  Two passes (count the non-zero results of every row, then form them again and write them at their final
// places): both do the full arithmetic, neither allocates per row.  Bit-identical to the host restatement
// (setup_host.cpp amgsetup_csr_matmat_*) and to scipy: same products, same order, separate multiply and add.
#include "hier.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

using namespace amg;

#define CHK(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

namespace {

struct Lap {                 // AMG_SETUP_VERBOSE=1: stage times on stderr
    bool on = std::getenv("AMG_SETUP_VERBOSE") && std::getenv("AMG_SETUP_VERBOSE")[0] != '0';
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what)
    {
        if (!on) return;
        hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[setup]     device Galerkin: %-28s %6.2fs\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

struct DCsr {              // a CSR operand in HBM (row pointer as 64-bit offsets)
    int n_row = 0, n_col = 0;
    long nnz = 0;
    long *Ap = nullptr;
    int *Aj = nullptr;
    double *Ax = nullptr;
    bool owned = true;
};

void dcsr_free(DCsr &M)
{
    if (M.owned) {
        if (M.Ap) hipFree(M.Ap);
        if (M.Aj) hipFree(M.Aj);
        if (M.Ax) hipFree(M.Ax);
    }
    M.Ap = nullptr; M.Aj = nullptr; M.Ax = nullptr;
}

// largest number of products of any output row (sizes the tables)
__global__ void spgemm_upper_kernel(int n_row, const long *Ap, const int *Aj, const long *Bp, int *row_upper)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_row) return;
    long u = 0;
    for (long jj = Ap[i]; jj < Ap[i + 1]; ++jj) { const int j = Aj[jj]; u += Bp[j + 1] - Bp[j]; }
    row_upper[i] = (int)min(u, 2147483647L);
}

// FILL = false: count[i] = number of non-zero results of row i.  FILL = true: write them at Cp[i] .. in reverse
// first-touch order.  Thread t owns table t (cap entries): keys stay -1 between rows (cleared through the order list).
template <bool FILL>
__global__ __launch_bounds__(256) void spgemm_rows_kernel(int n_row, const long *Ap, const int *Aj, const double *Ax,
                                                         const long *Bp, const int *Bj, const double *Bx, int cap,
                                                         int *keys, double *sums, int *order, int *count, const long *Cp,
                                                         int *Cj, double *Cx)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    int *key = keys + tid * cap;
    double *sum = sums + tid * cap;
    int *ord = order + tid * cap;
    const unsigned mask = (unsigned)cap - 1u;
    for (long i = tid; i < n_row; i += nthreads) {
        int n_ins = 0;
        for (long jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            const int j = Aj[jj];
            const double v = Ax[jj];
            for (long kk = Bp[j]; kk < Bp[j + 1]; ++kk) {
                const int c = Bj[kk];
                unsigned h = ((unsigned)c * 2654435761u) & mask;
                for (;;) {
                    const int k = key[h];
                    if (k == c) break;
                    if (k == -1) { key[h] = c; sum[h] = 0.0; ord[n_ins++] = (int)h; break; }
                    h = (h + 1u) & mask;
                }
                const double p = v * Bx[kk];
                sum[h] = sum[h] + p;
            }
        }
        if (FILL) {
            long at = Cp[i];
            for (int q = n_ins - 1; q >= 0; --q) {
                const int h = ord[q];
                const double s = sum[h];
                if (s != 0.0) { Cj[at] = key[h]; Cx[at] = s; ++at; }
                key[h] = -1;
            }
        } else {
            int nz = 0;
            for (int q = 0; q < n_ins; ++q) {
                const int h = ord[q];
                nz += (sum[h] != 0.0) ? 1 : 0;
                key[h] = -1;
            }
            count[i] = nz;
        }
    }
}

// Long rows (thousands of products, e.g. the Galerkin product of the second level of a 3-D hierarchy): one WAVE per
// output row, the table in LDS.  The entries of the left-hand row are taken strictly one after the other; the lanes
// take the entries of the right-hand row that entry selects -- distinct columns, so no two lanes ever add to the
// same sum in one step and every sum still receives its products in the sequential order.  New columns of a step are
// numbered in lane (= entry) order by a ballot, which is the sequential first-touch order.  A row whose distinct
// columns do not fit the table raises `overflow` (the caller then takes its host path).
constexpr int WCAP = 4096;                 // table entries per wave: 64 KB of LDS, two waves per compute unit
template <bool FILL>
__global__ __launch_bounds__(64) void spgemm_wave_kernel(int n_row, const long *Ap, const int *Aj, const double *Ax,
                                                        const long *Bp, const int *Bj, const double *Bx, int *count,
                                                        const long *Cp, int *Cj, double *Cx, int *overflow)
{
    __shared__ int key[WCAP];
    __shared__ double sum[WCAP];
    __shared__ int ord[WCAP];
    const int lane = threadIdx.x;
    const unsigned mask = WCAP - 1;
    for (int q = lane; q < WCAP; q += 64) key[q] = -1;
    __syncthreads();
    for (long i = blockIdx.x; i < n_row; i += gridDim.x) {
        int n_ins = 0;
        bool bad = false;
        for (long jj = Ap[i]; jj < Ap[i + 1] && !bad; ++jj) {
            const int j = Aj[jj];
            const double v = Ax[jj];
            const long b0 = Bp[j], b1 = Bp[j + 1];
            for (long base = b0; base < b1; base += 64) {
                if (n_ins + 64 > WCAP / 2) { bad = true; break; }          // keep the table at most half full
                const long kk = base + lane;
                bool inserted = false;
                int h = 0;
                if (kk < b1) {
                    const int c = Bj[kk];
                    h = (int)(((unsigned)c * 2654435761u) & mask);
                    for (;;) {
                        const int k = atomicCAS(&key[h], -1, c);
                        if (k == -1) { inserted = true; sum[h] = 0.0; break; }
                        if (k == c) break;
                        h = (h + 1) & (int)mask;
                    }
                    const double p = v * Bx[kk];
                    sum[h] = sum[h] + p;
                }
                const unsigned long long m = __ballot(inserted);
                if (inserted) ord[n_ins + __popcll(m & ((1ULL << lane) - 1ULL))] = h;
                n_ins += __popcll(m);
                __syncthreads();                                            // (one wave: orders the LDS accesses of the step)
            }
        }
        if (bad) { if (lane == 0) *overflow = 1; }
        if (FILL) {
            long at = Cp[i];
            for (int q0 = n_ins - 1; q0 >= 0; q0 -= 64) {                   // reverse first-touch order, zeros dropped
                const int q = q0 - lane;
                double sv = 0.0; int kv = 0;
                if (q >= 0) { const int h = ord[q]; sv = sum[h]; kv = key[h]; }
                const bool keep = (q >= 0) && sv != 0.0 && !bad;
                const unsigned long long m = __ballot(keep);
                if (keep) { const long w = at + __popcll(m & ((1ULL << lane) - 1ULL)); Cj[w] = kv; Cx[w] = sv; }
                at += __popcll(m);
            }
        } else {
            int nz = 0;
            for (int q0 = 0; q0 < n_ins; q0 += 64) {
                const int q = q0 + lane;
                const bool keep = (q < n_ins) && sum[ord[q]] != 0.0;
                nz += __popcll(__ballot(keep));
            }
            if (lane == 0) count[i] = nz;
        }
        __syncthreads();
        for (int q = lane; q < n_ins; q += 64) key[ord[q]] = -1;
        __syncthreads();
    }
}

int upload_dcsr(DCsr &M, int n_row, int n_col, const long *Ap, const int *Aj, const double *Ax)
{
    M.n_row = n_row; M.n_col = n_col; M.nnz = Ap[n_row];
    AMG_HIP(hipMalloc((void **)&M.Ap, sizeof(long) * ((size_t)n_row + 1)));
    AMG_HIP(hipMalloc((void **)&M.Aj, sizeof(int) * (size_t)std::max(M.nnz, 1L)));
    AMG_HIP(hipMalloc((void **)&M.Ax, sizeof(double) * (size_t)std::max(M.nnz, 1L)));
    AMG_HIP(hipMemcpy(M.Ap, Ap, sizeof(long) * ((size_t)n_row + 1), hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(M.Aj, Aj, sizeof(int) * (size_t)M.nnz, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(M.Ax, Ax, sizeof(double) * (size_t)M.nnz, hipMemcpyHostToDevice));
    return 0;
}

// C = A * B, both in HBM; C allocated here
int matmat(const DCsr &A, const DCsr &B, DCsr &C)
{
    if (A.n_col != B.n_row) { set_error("spgemm: inner dimensions differ"); return AMG_EINVAL; }
    C.n_row = A.n_row; C.n_col = B.n_col;
    const int n = A.n_row;
    int *upper = nullptr;
    AMG_HIP(hipMalloc((void **)&upper, sizeof(int) * (size_t)std::max(n, 1)));
    hipLaunchKernelGGL(spgemm_upper_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, A.Ap, A.Aj, B.Ap, upper);
    std::vector<int> hu((size_t)n);
    AMG_HIP(hipMemcpy(hu.data(), upper, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    hipFree(upper);
    const int max_upper = n ? *std::max_element(hu.begin(), hu.end()) : 0;
    int cap = 64;
    while (cap < 2 * max_upper && cap < (1 << 30)) cap <<= 1;
    // one thread per row (private table in HBM) pays while the tables stay small or the rows are few; long rows of a
    // large level go one wave per row with the table in LDS
    const bool by_thread = max_upper <= 1024 || (double)n * (double)cap * 16.0 <= 2.0e9;
    if (!by_thread) {
        int *count = nullptr, *overflow = nullptr;
        AMG_HIP(hipMalloc((void **)&count, sizeof(int) * (size_t)std::max(n, 1)));
        AMG_HIP(hipMalloc((void **)&overflow, sizeof(int)));
        AMG_HIP(hipMemset(overflow, 0, sizeof(int)));
        const int blocks = std::min(n, 256 * 2 * 4);
        hipLaunchKernelGGL((spgemm_wave_kernel<false>), dim3(blocks), dim3(64), 0, nullptr, n, A.Ap, A.Aj, A.Ax, B.Ap, B.Aj, B.Ax,
                           count, (const long *)nullptr, (int *)nullptr, (double *)nullptr, overflow);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "spgemm wave count launch", __FILE__, __LINE__);
        int ovf = 0;
        AMG_HIP(hipMemcpy(&ovf, overflow, sizeof(int), hipMemcpyDeviceToHost));
        if (ovf) {
            hipFree(count); hipFree(overflow);
            set_error("spgemm: a row with more distinct columns than the LDS table holds (host path)");
            return AMG_EINVAL;
        }
        std::vector<int> hc((size_t)n);
        AMG_HIP(hipMemcpy(hc.data(), count, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
        std::vector<long> cp((size_t)n + 1);
        cp[0] = 0;
        for (int i = 0; i < n; ++i) cp[(size_t)i + 1] = cp[(size_t)i] + hc[(size_t)i];
        C.nnz = cp[(size_t)n];
        AMG_HIP(hipMalloc((void **)&C.Ap, sizeof(long) * ((size_t)n + 1)));
        AMG_HIP(hipMalloc((void **)&C.Aj, sizeof(int) * (size_t)std::max(C.nnz, 1L)));
        AMG_HIP(hipMalloc((void **)&C.Ax, sizeof(double) * (size_t)std::max(C.nnz, 1L)));
        AMG_HIP(hipMemcpy(C.Ap, cp.data(), sizeof(long) * ((size_t)n + 1), hipMemcpyHostToDevice));
        hipLaunchKernelGGL((spgemm_wave_kernel<true>), dim3(blocks), dim3(64), 0, nullptr, n, A.Ap, A.Aj, A.Ax, B.Ap, B.Aj, B.Ax,
                           count, C.Ap, C.Aj, C.Ax, overflow);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "spgemm wave fill launch", __FILE__, __LINE__);
        AMG_HIP(hipDeviceSynchronize());
        hipFree(count); hipFree(overflow);
        return 0;
    }
    // resident threads: as many as 3 GB of tables allow, at most 512 per compute unit
    long threads = std::min<long>(256L * 512L, (3L << 30) / ((long)cap * 16L));
    threads = std::max<long>(256, std::min<long>(threads, ((long)n + 255) / 256 * 256));
    const int blocks = (int)((threads + 255) / 256);
    threads = (long)blocks * 256;
    int *keys = nullptr, *order = nullptr, *count = nullptr;
    double *sums = nullptr;
    AMG_HIP(hipMalloc((void **)&keys, sizeof(int) * (size_t)(threads * cap)));
    AMG_HIP(hipMalloc((void **)&order, sizeof(int) * (size_t)(threads * cap)));
    AMG_HIP(hipMalloc((void **)&sums, sizeof(double) * (size_t)(threads * cap)));
    AMG_HIP(hipMalloc((void **)&count, sizeof(int) * (size_t)std::max(n, 1)));
    AMG_HIP(hipMemset(keys, 0xFF, sizeof(int) * (size_t)(threads * cap)));
    hipLaunchKernelGGL((spgemm_rows_kernel<false>), dim3(blocks), dim3(256), 0, nullptr, n, A.Ap, A.Aj, A.Ax, B.Ap, B.Aj, B.Ax,
                       cap, keys, sums, order, count, (const long *)nullptr, (int *)nullptr, (double *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "spgemm count launch", __FILE__, __LINE__);
    std::vector<int> hc((size_t)n);
    AMG_HIP(hipMemcpy(hc.data(), count, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<long> cp((size_t)n + 1);
    cp[0] = 0;
    for (int i = 0; i < n; ++i) cp[(size_t)i + 1] = cp[(size_t)i] + hc[(size_t)i];
    C.nnz = cp[(size_t)n];
    AMG_HIP(hipMalloc((void **)&C.Ap, sizeof(long) * ((size_t)n + 1)));
    AMG_HIP(hipMalloc((void **)&C.Aj, sizeof(int) * (size_t)std::max(C.nnz, 1L)));
    AMG_HIP(hipMalloc((void **)&C.Ax, sizeof(double) * (size_t)std::max(C.nnz, 1L)));
    AMG_HIP(hipMemcpy(C.Ap, cp.data(), sizeof(long) * ((size_t)n + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((spgemm_rows_kernel<true>), dim3(blocks), dim3(256), 0, nullptr, n, A.Ap, A.Aj, A.Ax, B.Ap, B.Aj, B.Ax,
                       cap, keys, sums, order, count, C.Ap, C.Aj, C.Ax);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "spgemm fill launch", __FILE__, __LINE__);
    AMG_HIP(hipDeviceSynchronize());
    hipFree(keys); hipFree(order); hipFree(sums); hipFree(count);
    return 0;
}

__global__ void widen_rowptr_kernel(int n, const int *Ap32, long *Ap64)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) Ap64[i] = Ap32[i];
}

}   // namespace

struct amg_galerkin {
    DCsr C;
    std::vector<long> cp;
};

extern "C" {

// Ac = (R * A) * P with A taken from level `level` of a hierarchy handle that already holds it in HBM (the handle of
// the setup-time spectral-radius estimate), R and P handed in as host CSR arrays (64-bit row pointers).  On return
// Cp (n_coarse + 1 entries) is filled and *out holds the product; amg_galerkin_fetch copies its columns and values
// (Cp[n_coarse] entries each) to the host and releases it.  AMG_EINVAL when the operator is not held as plain CSR or a
// row is too long for the device tables: the caller then uses its host path.
int amg_hier_galerkin(amg_hier *h, int level, int n_coarse, const int64_t *Rp, const int *Rj, const double *Rx,
                      const int64_t *Pp, const int *Pj, const double *Px, int64_t *Cp, amg_galerkin **out)
{
    if (!h || !out || level < 0 || level >= (int)h->lv.size()) { set_error("galerkin: bad arguments"); return AMG_EINVAL; }
    const DevCsr &M = h->lv[(size_t)level].A;
    if (!M.Ap || !M.Aj || !M.Ax) { set_error("galerkin: the operator is not held as CSR"); return AMG_EINVAL; }
    AMG_HIP(hipSetDevice(h->device));
    const int n = M.nrows;
    Lap lap;
    DCsr A, R, P, RA;
    A.n_row = n; A.n_col = M.ncols; A.nnz = M.nnz; A.Aj = M.Aj; A.Ax = M.Ax; A.owned = false;
    AMG_HIP(hipMalloc((void **)&A.Ap, sizeof(long) * ((size_t)n + 1)));
    hipLaunchKernelGGL(widen_rowptr_kernel, dim3((n + 256) / 256), dim3(256), 0, nullptr, n, M.Ap, A.Ap);
    int rc = upload_dcsr(R, n_coarse, n, (const long *)Rp, Rj, Rx);
    lap("R to HBM");
    if (rc == 0) rc = matmat(R, A, RA);
    lap("R*A");
    dcsr_free(R);
    hipFree(A.Ap);
    if (rc != 0) { dcsr_free(RA); return rc; }
    rc = upload_dcsr(P, n, n_coarse, (const long *)Pp, Pj, Px);
    lap("P to HBM");
    amg_galerkin *g = new amg_galerkin;
    if (rc == 0) rc = matmat(RA, P, g->C);
    lap("(R*A)*P");
    dcsr_free(P);
    dcsr_free(RA);
    if (rc != 0) { dcsr_free(g->C); delete g; return rc; }
    if (hipMemcpy(Cp, g->C.Ap, sizeof(long) * ((size_t)n_coarse + 1), hipMemcpyDeviceToHost) != hipSuccess) {
        dcsr_free(g->C); delete g;
        set_error("galerkin: row pointer download failed");
        return AMG_ENODEV;
    }
    *out = g;
    return 0;
}

int amg_galerkin_fetch(amg_galerkin *g, int *Cj, double *Cx)
{
    if (!g) return AMG_EINVAL;
    int rc = 0;
    if (g->C.nnz > 0) {
        if (hipMemcpy(Cj, g->C.Aj, sizeof(int) * (size_t)g->C.nnz, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(Cx, g->C.Ax, sizeof(double) * (size_t)g->C.nnz, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("galerkin: download failed");
            rc = AMG_ENODEV;
        }
    }
    dcsr_free(g->C);
    delete g;
    return rc;
}

// C = A * B for host CSR operands through the device kernels (tests; same arithmetic as amg_hier_galerkin's products)
int amg_csr_matmat_device(int n_row, int n_inner, int n_col, const int64_t *Ap, const int *Aj, const double *Ax,
                          const int64_t *Bp, const int *Bj, const double *Bx, int64_t *Cp, amg_galerkin **out)
{
    if (!out) return AMG_EINVAL;
    DCsr A, B;
    int rc = upload_dcsr(A, n_row, n_inner, (const long *)Ap, Aj, Ax);
    if (rc == 0) rc = upload_dcsr(B, n_inner, n_col, (const long *)Bp, Bj, Bx);
    amg_galerkin *g = new amg_galerkin;
    if (rc == 0) rc = matmat(A, B, g->C);
    dcsr_free(A); dcsr_free(B);
    if (rc != 0) { dcsr_free(g->C); delete g; return rc; }
    if (hipMemcpy(Cp, g->C.Ap, sizeof(long) * ((size_t)n_row + 1), hipMemcpyDeviceToHost) != hipSuccess) {
        dcsr_free(g->C); delete g;
        set_error("matmat: row pointer download failed");
        return AMG_ENODEV;
    }
    *out = g;
    return 0;
}

}   // extern "C"
