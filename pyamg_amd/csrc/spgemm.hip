// Galerkin products of the setup on the device: C = A * B for CSR operands with scipy's csr_matmat arithmetic and
// output order (scipy/sparse/sparsetools/csr.h csr_matmat, called by pyamg/aggregation/aggregation.py:425-426 as
// R * A * P) -- per output row the products are accumulated in the order (entry of A's row, entry of B's row), the
// output columns come out in REVERSE first-touch order, exact-zero results are dropped.
//
// Main kernel (spgemm_group_kernel): G lanes per output row, every row with a table of its own in LDS; left-hand
// entries strictly in order, the lanes over the right-hand row each entry selects.  Fall-backs for rows whose distinct
// columns outgrow the LDS table: the whole wave per row, then one thread per row with private tables in HBM
// (spgemm_rows_kernel, two tiers), then the caller's host path.  Every path runs a count pass (non-zero results per
// row) and a fill pass (results written at their final places); both do the full arithmetic, neither allocates per
// row.  Bit-identical to the host restatement (setup_host.cpp amgsetup_csr_matmat_*) and to scipy: same products, same
// order, separate multiply and add.
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

// FILL = false: count[i] = number of non-zero results of row i (-1: the row has more than `limit` distinct columns and
// is left to a launch with larger tables).  FILL = true: write them at Cp[i] .. in reverse first-touch order (rows
// marked -1 are skipped).  Rows: all (`rows` null) or the listed ones.  Thread t owns table t (cap entries): keys stay
// -1 between rows (cleared through the order list).
// Tables sized by the PRODUCTS of the longest row would be 16 KB per thread here and live in HBM (2 GB for 131 k
// threads): every probe a random 64-byte HBM access, 0.77 s per product of the 500^3 level.  The rows' DISTINCT
// columns are far fewer, so the first launch runs with 256-entry tables (4 KB per thread, 256 MB in all: Infinity
// Cache / L2) and only rows that outgrow them are redone with the large ones.
template <bool FILL>
__global__ __launch_bounds__(256) void spgemm_rows_kernel(int n_work, const int *rows, const long *Ap, const int *Aj, const double *Ax,
                                                         const long *Bp, const int *Bj, const double *Bx, int cap, int limit,
                                                         int *keys, double *sums, int *order, int *count, const long *Cp,
                                                         int *Cj, double *Cx)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    int *key = keys + tid * cap;
    double *sum = sums + tid * cap;
    int *ord = order + tid * cap;
    const unsigned mask = (unsigned)cap - 1u;
    for (long w = tid; w < n_work; w += nthreads) {
        const long i = rows ? rows[w] : w;
        if (FILL && !rows && count[i] < 0) continue;                // done by the large-table launch
        int n_ins = 0;
        bool over = false;
        for (long jj = Ap[i]; jj < Ap[i + 1] && !over; ++jj) {
            const int j = Aj[jj];
            const double v = Ax[jj];
            for (long kk = Bp[j]; kk < Bp[j + 1]; ++kk) {
                const int c = Bj[kk];
                unsigned h = ((unsigned)c * 2654435761u) & mask;
                for (;;) {
                    const int k = key[h];
                    if (k == c) break;
                    if (k == -1) {
                        if (n_ins >= limit) { over = true; break; }
                        key[h] = c; sum[h] = 0.0; ord[n_ins++] = (int)h;
                        break;
                    }
                    h = (h + 1u) & mask;
                }
                if (over) break;
                const double p = v * Bx[kk];
                sum[h] = sum[h] + p;
            }
        }
        if (over) {
            for (int q = 0; q < n_ins; ++q) key[ord[q]] = -1;
            if (!FILL) count[i] = -1;
            continue;
        }
        if (FILL) {
            long at = Cp[i];
            for (int q = n_ins - 1; q >= 0; --q) {
                const int h = ord[q];
                const double sv = sum[h];
                if (sv != 0.0) { Cj[at] = key[h]; Cx[at] = sv; ++at; }
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

// The main kernel: G lanes per output row (64 / G rows per wave), every row with a table of its own in LDS.  The
// entries of the left-hand row are taken strictly one after the other; the G lanes take the entries of the right-hand
// row that entry selects (contiguous in memory: coalesced) -- distinct columns, so no two lanes ever add to the same
// sum in one step and every sum still receives its products in the sequential order.  New columns of a step are
// numbered in lane (= entry) order by a ballot, which is the sequential first-touch order.  G is chosen from the
// right-hand operand's average row length (7-point operator: 8 lanes, eight rows per wave with 512-entry tables;
// long rows: the whole wave on one row with 4096 entries).  A row whose distinct columns do not fit its table raises
// `overflow`: the caller retries with the whole wave per row, then with the one-thread-per-row kernel (tables in HBM),
// then hands the product back to its host path.
constexpr int WCAP = 4096;                 // table entries per wave (64 KB of LDS, two waves per compute unit)
// HBM = true (G = 64 only): the table of the wave's row lives in HBM instead (gcap entries per workgroup, keys preset
// to -1 by the host) -- rows of tens of thousands of products on small coarse levels, where a few hundred waves with
// 2 MB tables each are plenty.
template <bool FILL, int G, bool HBM>
__global__ __launch_bounds__(64) void spgemm_group_kernel(int n_row, const long *Ap, const int *Aj, const double *Ax,
                                                         const long *Bp, const int *Bj, const double *Bx, int *count,
                                                         const long *Cp, int *Cj, double *Cx, int *overflow,
                                                         int *gkey, double *gsum, int *gord, int gcap)
{
    static_assert(!HBM || G == 64, "tables in HBM: one row per wave");
    constexpr int NG = 64 / G;             // rows per wave
    constexpr int LB = 64;                 // left-hand entries staged per batch and row
    __shared__ int key_s[HBM ? 1 : WCAP];
    __shared__ double sum_s[HBM ? 1 : WCAP];
    __shared__ int ord_s[HBM ? 1 : WCAP];
    __shared__ double stv_s[NG * LB];
    __shared__ long stb_s[NG * LB];
    __shared__ int stl_s[NG * LB];
    const int lane = threadIdx.x;
    const int g = lane / G, gl = lane % G;                          // group (row slot) and lane within the group
    const int CAP = HBM ? gcap : WCAP / NG;                         // table entries per row
    int *key = HBM ? gkey + (long)blockIdx.x * gcap : key_s + g * CAP;
    double *sum = HBM ? gsum + (long)blockIdx.x * gcap : sum_s + g * CAP;
    int *ord = HBM ? gord + (long)blockIdx.x * gcap : ord_s + g * CAP;
    double *st_v = stv_s + g * LB;
    long *st_b0 = stb_s + g * LB;
    int *st_len = stl_s + g * LB;
    const unsigned mask = (unsigned)CAP - 1u;
    const unsigned long long gmask = (G == 64) ? ~0ULL : (((1ULL << G) - 1ULL) << (g * G));
    const unsigned long long below = (G == 64 ? ((1ULL << lane) - 1ULL) : (((1ULL << gl) - 1ULL) << (g * G)));
    if (!HBM) for (int q = lane; q < WCAP; q += 64) key_s[q] = -1;
    __syncthreads();
    const long stride = (long)gridDim.x * NG;
    const long first = (long)blockIdx.x * NG + g;
    const long rounds = (n_row + stride - 1) / stride;              // the same for every lane: ballots below are wave-wide
    for (long r = 0; r < rounds; ++r) {
        const long i = first + r * stride;
        const bool have = i < n_row;
        int n_ins = 0;
        bool bad = false;
        const long jbeg = have ? Ap[i] : 0, jend = have ? Ap[i + 1] : 0;
        // The left-hand row is staged LB entries at a time: the group's lanes fetch (value, start and length of the
        // right-hand row it selects) for a whole batch at once -- two dependent round trips per batch instead of per
        // entry -- and the first chunks of the next entries' right-hand rows are requested four entries ahead of
        // use.  What remains per entry is LDS work.
        for (long j0 = jbeg;; j0 += LB) {
            if (__ballot(!bad && j0 < jend) == 0ULL) break;
            __syncthreads();
            for (int e = gl; e < LB; e += G) {
                const long jj = j0 + e;
                int len = -1;                                           // -1: past the end of the row
                if (!bad && jj < jend) {
                    const int j = Aj[jj];
                    const long b0 = Bp[j];
                    len = (int)(Bp[j + 1] - b0);
                    st_v[e] = Ax[jj]; st_b0[e] = b0;
                }
                st_len[e] = len;
            }
            __syncthreads();
            // the first chunks of the next PD entries' right-hand rows are requested ahead of use (a ring of PD
            // register slots, the loop unrolled by PD so that the slot index is a constant)
            constexpr int PD = 4;
            int pc[PD]; double px[PD];
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                pc[u] = 0; px[u] = 0.0;
                const int l = st_len[u];
                if (gl < l) { const long b = st_b0[u]; pc[u] = Bj[b + gl]; px[u] = Bx[b + gl]; }
            }
            bool batch_done = false;
            for (int e0 = 0; e0 < LB && !batch_done; e0 += PD) {
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int e = e0 + u;
                    const int len = st_len[e];
                    if (__ballot(len >= 0) == 0ULL) { batch_done = true; break; }   // every row of the wave is through its batch
                    const long b0 = len >= 0 ? st_b0[e] : 0;
                    const int c0 = pc[u]; const double x0 = px[u];
                    const double v = len >= 0 ? st_v[e] : 0.0;
                    // refill this slot with entry e + PD before touching the table
                    pc[u] = 0; px[u] = 0.0;
                    if (e + PD < LB) {
                        const int l = st_len[e + PD];
                        if (gl < l) { const long b = st_b0[e + PD]; pc[u] = Bj[b + gl]; px[u] = Bx[b + gl]; }
                    }
                    // this entry: its right-hand row in chunks of G (the first one is already here)
                    for (int kb = 0;; kb += G) {
                        const bool active = !bad && kb < len;
                        if (__ballot(active) == 0ULL) break;
                        bool inserted = false;
                        int h = 0;
                        if (active) {
                            if (n_ins + G > CAP / 2) bad = true;        // keep the table at most half full
                            else if (kb + gl < len) {
                                int c; double x;
                                if (kb == 0) { c = c0; x = x0; } else { c = Bj[b0 + kb + gl]; x = Bx[b0 + kb + gl]; }
                                h = (int)(((unsigned)c * 2654435761u) & mask);
                                for (;;) {
                                    const int k = atomicCAS(&key[h], -1, c);
                                    if (k == -1) { inserted = true; sum[h] = 0.0; break; }
                                    if (k == c) break;
                                    h = (h + 1) & (int)mask;
                                }
                                const double p = v * x;
                                sum[h] = sum[h] + p;
                            }
                        }
                        const unsigned long long m = __ballot(inserted);
                        if (inserted) ord[n_ins + __popcll(m & below)] = h;
                        n_ins += __popcll(m & gmask);
                        __syncthreads();                                // (one wave: orders the LDS accesses of the step)
                    }
                }
            }
        }
        if (bad && gl == 0) *overflow = 1;
        if (FILL) {
            long at = have ? Cp[i] : 0;
            int q0 = n_ins - 1;
            for (;;) {                                              // reverse first-touch order, zeros dropped
                if (__ballot(q0 >= 0) == 0ULL) break;
                const int q = q0 - gl;
                double sv = 0.0; int kv = 0;
                if (q >= 0) { const int hh = ord[q]; sv = sum[hh]; kv = key[hh]; }
                const bool keep = (q >= 0) && sv != 0.0 && !bad;
                const unsigned long long m = __ballot(keep);
                if (keep) { const long w = at + __popcll(m & below); Cj[w] = kv; Cx[w] = sv; }
                at += __popcll(m & gmask);
                q0 -= G;
            }
        } else {
            int nz = 0, q0 = 0;
            for (;;) {
                if (__ballot(q0 < n_ins) == 0ULL) break;
                const int q = q0 + gl;
                const bool keep = (q < n_ins) && sum[ord[q]] != 0.0;
                nz += __popcll(__ballot(keep) & gmask);
                q0 += G;
            }
            if (have && gl == 0) count[i] = nz;
        }
        __syncthreads();
        for (int q = gl; q < n_ins; q += G) key[ord[q]] = -1;
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
    // G lanes per row with the tables in LDS; rows whose distinct columns outgrow their table send the whole product to
    // the one-thread-per-row kernel below (tables in HBM), which in turn refuses levels too large for it
    const double avgB = B.n_row > 0 ? (double)B.nnz / (double)B.n_row : 1.0;
    const int G0 = avgB <= 10.0 ? 8 : (avgB <= 20.0 ? 16 : (avgB <= 40.0 ? 32 : 64));
    for (int G = G0; G <= 64; G = (G == 64 ? 128 : 64)) {          // the preferred width, then the whole wave per row
        int *count = nullptr, *overflow = nullptr;
        AMG_HIP(hipMalloc((void **)&count, sizeof(int) * (size_t)std::max(n, 1)));
        AMG_HIP(hipMalloc((void **)&overflow, sizeof(int)));
        AMG_HIP(hipMemset(overflow, 0, sizeof(int)));
        const int rows_per_wave = 64 / G;
        const int blocks = (int)std::min<long>(((long)n + rows_per_wave - 1) / rows_per_wave, 256L * 2 * 8);
#define GROUP_LAUNCH(FILL, GG) hipLaunchKernelGGL((spgemm_group_kernel<FILL, GG, false>), dim3(blocks), dim3(64), 0, nullptr, n, A.Ap, A.Aj, A.Ax, \
                                                  B.Ap, B.Aj, B.Ax, count, (const long *)C.Ap, C.Aj, C.Ax, overflow, (int *)nullptr, \
                                                  (double *)nullptr, (int *)nullptr, 0)
#define GROUP_DISPATCH(FILL) do { if (G == 8) GROUP_LAUNCH(FILL, 8); else if (G == 16) GROUP_LAUNCH(FILL, 16); \
                                  else if (G == 32) GROUP_LAUNCH(FILL, 32); else GROUP_LAUNCH(FILL, 64); } while (0)
        C.Ap = nullptr; C.Aj = nullptr; C.Ax = nullptr;
        GROUP_DISPATCH(false);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "spgemm group count launch", __FILE__, __LINE__);
        int ovf = 0;
        AMG_HIP(hipMemcpy(&ovf, overflow, sizeof(int), hipMemcpyDeviceToHost));
        if (std::getenv("AMG_SETUP_VERBOSE") && std::getenv("AMG_SETUP_VERBOSE")[0] != '0')
            std::fprintf(stderr, "[setup]     device product %d x %d: %d lanes per row (right-hand rows avg %.1f, longest row %d products)%s\n",
                         A.n_row, B.n_col, G, avgB, max_upper, ovf ? " -- a row outgrew its LDS table" : "");
        if (!ovf) {
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
            GROUP_DISPATCH(true);
            e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "spgemm group fill launch", __FILE__, __LINE__);
            AMG_HIP(hipDeviceSynchronize());
            hipFree(count); hipFree(overflow);
            return 0;
        }
#undef GROUP_DISPATCH
#undef GROUP_LAUNCH
        hipFree(count); hipFree(overflow);
    }
    // whole wave per row with the row's table in HBM: long rows on a small level (a few thousand rows of tens of
    // thousands of products -- the coarse Galerkin products)
    if (max_upper > 1024 && max_upper <= (1 << 22)) {
        long distinct_bound = std::min<long>(max_upper, (long)B.n_col);
        int gcap = 1024;
        while (gcap < 2 * distinct_bound + 256) gcap <<= 1;
        const int blocks = (int)std::max<long>(1, std::min<long>(std::min<long>(n, 1024), (4L << 30) / ((long)gcap * 16L)));
        int *count = nullptr, *overflow = nullptr, *gkey = nullptr, *gord = nullptr;
        double *gsum = nullptr;
        AMG_HIP(hipMalloc((void **)&count, sizeof(int) * (size_t)std::max(n, 1)));
        AMG_HIP(hipMalloc((void **)&overflow, sizeof(int)));
        AMG_HIP(hipMemset(overflow, 0, sizeof(int)));
        AMG_HIP(hipMalloc((void **)&gkey, sizeof(int) * (size_t)blocks * gcap));
        AMG_HIP(hipMalloc((void **)&gord, sizeof(int) * (size_t)blocks * gcap));
        AMG_HIP(hipMalloc((void **)&gsum, sizeof(double) * (size_t)blocks * gcap));
        AMG_HIP(hipMemset(gkey, 0xFF, sizeof(int) * (size_t)blocks * gcap));
        C.Ap = nullptr; C.Aj = nullptr; C.Ax = nullptr;
        hipLaunchKernelGGL((spgemm_group_kernel<false, 64, true>), dim3(blocks), dim3(64), 0, nullptr, n, A.Ap, A.Aj, A.Ax, B.Ap, B.Aj, B.Ax,
                           count, (const long *)nullptr, (int *)nullptr, (double *)nullptr, overflow, gkey, gsum, gord, gcap);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "spgemm HBM-table count launch", __FILE__, __LINE__);
        int ovf = 0;
        AMG_HIP(hipMemcpy(&ovf, overflow, sizeof(int), hipMemcpyDeviceToHost));
        if (std::getenv("AMG_SETUP_VERBOSE") && std::getenv("AMG_SETUP_VERBOSE")[0] != '0')
            std::fprintf(stderr, "[setup]     device product %d x %d: one wave per row, %d-entry tables in HBM%s\n", A.n_row, B.n_col, gcap,
                         ovf ? " -- overflow" : "");
        int rc = 0;
        if (!ovf) {
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
            hipLaunchKernelGGL((spgemm_group_kernel<true, 64, true>), dim3(blocks), dim3(64), 0, nullptr, n, A.Ap, A.Aj, A.Ax, B.Ap, B.Aj, B.Ax,
                               count, (const long *)C.Ap, C.Aj, C.Ax, overflow, gkey, gsum, gord, gcap);
            e = hipGetLastError();
            if (e != hipSuccess) rc = hip_fail(e, "spgemm HBM-table fill launch", __FILE__, __LINE__);
            if (hipDeviceSynchronize() != hipSuccess && rc == 0) rc = AMG_ENODEV;
        }
        hipFree(count); hipFree(overflow); hipFree(gkey); hipFree(gord); hipFree(gsum);
        if (!ovf) return rc;
    }
    if (!(max_upper <= 1024 || (double)n * (double)cap * 16.0 <= 2.0e9)) {
        set_error("spgemm: rows with more distinct columns than the LDS tables hold on a level too large for per-thread tables (host path)");
        return AMG_EINVAL;
    }
    // first tier: 256-entry tables (up to 128 distinct columns per row), 65536 threads
    struct Tables {
        int *keys = nullptr, *order = nullptr; double *sums = nullptr; int cap = 0; long threads = 0; int blocks = 0;
        int make(int cap_, long want_threads)
        {
            cap = cap_;
            blocks = (int)((std::max<long>(256, want_threads) + 255) / 256);
            threads = (long)blocks * 256;
            AMG_HIP(hipMalloc((void **)&keys, sizeof(int) * (size_t)(threads * cap)));
            AMG_HIP(hipMalloc((void **)&order, sizeof(int) * (size_t)(threads * cap)));
            AMG_HIP(hipMalloc((void **)&sums, sizeof(double) * (size_t)(threads * cap)));
            AMG_HIP(hipMemset(keys, 0xFF, sizeof(int) * (size_t)(threads * cap)));
            return 0;
        }
        void drop() { if (keys) hipFree(keys); if (order) hipFree(order); if (sums) hipFree(sums); keys = order = nullptr; sums = nullptr; }
    };
    const int small_cap = std::min(cap, 256);
    Tables T1, T2;
    CHK(T1.make(small_cap, std::min<long>(65536, n)));
    int *count = nullptr, *big_rows = nullptr;
    AMG_HIP(hipMalloc((void **)&count, sizeof(int) * (size_t)std::max(n, 1)));
    hipLaunchKernelGGL((spgemm_rows_kernel<false>), dim3(T1.blocks), dim3(256), 0, nullptr, n, (const int *)nullptr, A.Ap, A.Aj, A.Ax,
                       B.Ap, B.Aj, B.Ax, T1.cap, T1.cap / 2, T1.keys, T1.sums, T1.order, count, (const long *)nullptr,
                       (int *)nullptr, (double *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "spgemm count launch", __FILE__, __LINE__);
    std::vector<int> hc((size_t)n);
    AMG_HIP(hipMemcpy(hc.data(), count, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<int> big;
    for (int i = 0; i < n; ++i) if (hc[(size_t)i] < 0) big.push_back(i);
    if (!big.empty()) {
        // second tier: the rows that outgrew the small tables, with tables sized by the longest row's products
        long threads = std::min<long>(256L * 512L, (3L << 30) / ((long)cap * 16L));
        CHK(T2.make(cap, std::min<long>(threads, (long)big.size())));
        AMG_HIP(hipMalloc((void **)&big_rows, sizeof(int) * big.size()));
        AMG_HIP(hipMemcpy(big_rows, big.data(), sizeof(int) * big.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL((spgemm_rows_kernel<false>), dim3(T2.blocks), dim3(256), 0, nullptr, (int)big.size(), big_rows, A.Ap, A.Aj, A.Ax,
                           B.Ap, B.Aj, B.Ax, T2.cap, T2.cap / 2, T2.keys, T2.sums, T2.order, count, (const long *)nullptr,
                           (int *)nullptr, (double *)nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "spgemm count launch (large tables)", __FILE__, __LINE__);
        AMG_HIP(hipMemcpy(hc.data(), count, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
        for (int i : big) if (hc[(size_t)i] < 0) { set_error("spgemm: row overflowed the large table"); return AMG_ESTATE; }
    }
    std::vector<long> cp((size_t)n + 1);
    cp[0] = 0;
    for (int i = 0; i < n; ++i) cp[(size_t)i + 1] = cp[(size_t)i] + hc[(size_t)i];
    C.nnz = cp[(size_t)n];
    AMG_HIP(hipMalloc((void **)&C.Ap, sizeof(long) * ((size_t)n + 1)));
    AMG_HIP(hipMalloc((void **)&C.Aj, sizeof(int) * (size_t)std::max(C.nnz, 1L)));
    AMG_HIP(hipMalloc((void **)&C.Ax, sizeof(double) * (size_t)std::max(C.nnz, 1L)));
    AMG_HIP(hipMemcpy(C.Ap, cp.data(), sizeof(long) * ((size_t)n + 1), hipMemcpyHostToDevice));
    if (!big.empty()) {
        // mark the large rows in `count` again (the first-tier fill skips rows with count < 0)
        std::vector<int> mark(hc);
        for (int i : big) mark[(size_t)i] = -1;
        AMG_HIP(hipMemcpy(count, mark.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((spgemm_rows_kernel<true>), dim3(T1.blocks), dim3(256), 0, nullptr, n, (const int *)nullptr, A.Ap, A.Aj, A.Ax,
                       B.Ap, B.Aj, B.Ax, T1.cap, T1.cap / 2, T1.keys, T1.sums, T1.order, count, C.Ap, C.Aj, C.Ax);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "spgemm fill launch", __FILE__, __LINE__);
    if (!big.empty()) {
        hipLaunchKernelGGL((spgemm_rows_kernel<true>), dim3(T2.blocks), dim3(256), 0, nullptr, (int)big.size(), big_rows, A.Ap, A.Aj, A.Ax,
                           B.Ap, B.Aj, B.Ax, T2.cap, T2.cap / 2, T2.keys, T2.sums, T2.order, count, C.Ap, C.Aj, C.Ax);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "spgemm fill launch (large tables)", __FILE__, __LINE__);
    }
    AMG_HIP(hipDeviceSynchronize());
    T1.drop(); T2.drop();
    hipFree(count);
    if (big_rows) hipFree(big_rows);
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
