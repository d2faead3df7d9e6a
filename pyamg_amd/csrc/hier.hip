// Device-resident multigrid hierarchy: multilevel_solver.solve()/__solve()
// (/root/reference/pyamg/multilevel.py:316-548) with every operator, smoother
// constant and work vector in HBM.  The host only sequences kernel launches on
// one HIP stream; one 8-byte copy per iteration brings the residual norm back
// when a tolerance has to be checked.
#include <array>
#include <thread>
#include "hier.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace amg {

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const std::string &last_error() { return g_err; }
int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" +
            std::to_string(line) + ")";
    return (e == hipErrorOutOfMemory) ? AMG_ENOMEM : AMG_ENODEV;
}

#define CHK(call)                 \
    do {                          \
        int rc__ = (call);        \
        if (rc__ != 0) return rc__; \
    } while (0)

// ------------------------------------------------------------------ memory
template <class T> static int dev_alloc(T **p, long count, long *acct)
{
    size_t bytes = sizeof(T) * (size_t)(count + PAD);
    AMG_HIP(hipMalloc((void **)p, bytes));
    AMG_HIP(hipMemset(*p, 0, bytes));
    // the fill runs on the NULL stream and is asynchronous to the host; the hierarchy's stream is
    // non-blocking, so without this wait a later copy/kernel on it could be overtaken by the fill
    AMG_HIP(hipDeviceSynchronize());
    if (acct) *acct += (long)bytes;
    return 0;
}

void free_csr(DevCsr &M)
{
    if (M.Ap) hipFree(M.Ap);
    if (M.Aj) hipFree(M.Aj);
    if (M.Ax) hipFree(M.Ax);
    if (M.pat) hipFree(M.pat);
    if (M.dict_ptr) hipFree(M.dict_ptr);
    if (M.dict_off) hipFree(M.dict_off);
    free_sell(M);
    if (M.st_vals) hipFree(M.st_vals);
    if (M.st_mask) hipFree(M.st_mask);
    if (M.st_codes) hipFree(M.st_codes);
    if (M.st_dict) hipFree(M.st_dict);
    if (M.Aj16) hipFree(M.Aj16);
    if (M.wg_base) hipFree(M.wg_base);
    if (M.wg_flag) hipFree(M.wg_flag);
    M = DevCsr();
}
static void free_bsr(DevBsr &M)
{
    free_bsell(M);
    if (M.Ap) hipFree(M.Ap);
    if (M.Aj) hipFree(M.Aj);
    if (M.Ax) hipFree(M.Ax);
    M = DevBsr();
}

int upload_csr(DevCsr &M, int nrows, int ncols, const int *Ap, const int *Aj, const double *Ax,
               long *acct)
{
    long nnz = Ap[nrows];
    if (nnz < 0 || nrows < 0) { set_error("row pointer overflows int32 (more than 2^31-1 stored entries)"); return AMG_EINVAL; }
    M.nrows = nrows; M.ncols = ncols; M.nnz = nnz;
    CHK(dev_alloc(&M.Ap, nrows + 1, acct));
    CHK(dev_alloc(&M.Aj, nnz, acct));
    CHK(dev_alloc(&M.Ax, nnz, acct));
    AMG_HIP(hipMemcpy(M.Ap, Ap, sizeof(int) * (size_t)(nrows + 1), hipMemcpyHostToDevice));
    if (nnz) {
        AMG_HIP(hipMemcpy(M.Aj, Aj, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        AMG_HIP(hipMemcpy(M.Ax, Ax, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    return 0;
}

// 16-bit column codes for the stream kernel (amg_dev.hpp, DevCsr::Aj16).  Built on the device for the
// row blocking the kernel will use; kept when at least 30 % of the entries could be coded.
int build_index16(DevCsr &M, const int *Ap_host, long *acct)
{
    const char *env = getenv("AMG_INDEX16");
    if (!(index16_enabled() || (env && atoi(env) != 0))) return 0;      // opt-in (amg_set_index16 before upload)
    if (M.nnz < 65536 || M.ncols < 8192 || M.st_vals) return 0;
    const int rpb = rows_per_wg_for(M.nnz, M.nrows);
    const int nwg = (M.nrows + rpb - 1) / rpb;
    long before = acct ? *acct : 0;
    CHK(dev_alloc(&M.Aj16, M.nnz, acct));
    CHK(dev_alloc(&M.wg_base, (long)nwg * 16, acct));
    CHK(dev_alloc(&M.wg_flag, nwg, acct));
    M.i16_rpb = rpb;
    int rc = launch_index16_build(M, rpb, 0);
    AMG_HIP(hipDeviceSynchronize());
    std::vector<unsigned char> flag((size_t)nwg);
    AMG_HIP(hipMemcpy(flag.data(), M.wg_flag, (size_t)nwg, hipMemcpyDeviceToHost));
    double coded = 0.0;
    for (int w = 0; w < nwg; ++w)
        if (flag[w]) {
            const int r0 = w * rpb, r1 = std::min(M.nrows, r0 + rpb);
            coded += (double)(Ap_host[r1] - Ap_host[r0]);
        }
    M.i16_frac = M.nnz ? coded / (double)M.nnz : 0.0;
    if (rc != 0 || M.i16_frac < 0.3) {
        hipFree(M.Aj16); hipFree(M.wg_base); hipFree(M.wg_flag);
        M.Aj16 = nullptr; M.wg_base = nullptr; M.wg_flag = nullptr; M.i16_rpb = 0; M.i16_frac = 0.0;
        if (acct) *acct = before;
    }
    return rc;
}

// Stencil form on top of the pattern dictionary.  U = the offsets that at least 1/32 of the rows
// use (all of them when that keeps the padding small); rows with a rarer offset -- the halo columns
// of a rank-local operator, an irregular corner of a mostly structured grid -- stay with the
// offset-pattern kernel as up to STENCIL_RANGES contiguous row ranges.  Slot order is a topological
// order of "precedes in some row's stored sequence", so that the slots with their mask bit set, in
// slot order, are exactly the row's stored order (the increasing order when rows are sorted).
// The value copy is built on the device from the CSR arrays already there.
static int try_stencil(DevCsr &M, const std::vector<int> &dptr, const std::vector<int> &doff,
                       const std::vector<long> &pcount, const std::vector<int> &pat, long *acct)
{
    const char *env = getenv("AMG_STENCIL");
    if (env && atoi(env) == 0) return 0;
    const int npat = (int)dptr.size() - 1;
    const long n = M.nrows;
    std::vector<int> V(doff);                       // distinct offsets
    std::sort(V.begin(), V.end());
    V.erase(std::unique(V.begin(), V.end()), V.end());
    auto id_in = [](const std::vector<int> &W, int o) {
        auto it = std::lower_bound(W.begin(), W.end(), o);
        return (it != W.end() && *it == o) ? (int)(it - W.begin()) : -1;
    };
    // drop the least-used offset until U fits and the padding is small; the rows that lose an
    // offset leave the stencil form
    std::vector<long> use(V.size(), 0);
    for (int p = 0; p < npat; ++p)
        for (int q = dptr[p]; q < dptr[p + 1]; ++q) use[(size_t)id_in(V, doff[q])] += pcount[p];
    std::vector<char> covered((size_t)npat, 1);
    long rows_cov = 0, nnz_cov = 0;
    for (;;) {
        rows_cov = nnz_cov = 0;
        for (int p = 0; p < npat; ++p) {
            covered[p] = 1;
            for (int q = dptr[p]; q < dptr[p + 1]; ++q) if (id_in(V, doff[q]) < 0) covered[p] = 0;
            if (covered[p]) { rows_cov += pcount[p]; nnz_cov += pcount[p] * (dptr[p + 1] - dptr[p]); }
        }
        if (rows_cov * 4 < n * 3 || V.empty()) return 0;                     // less than 3/4 of the rows fit
        if ((int)V.size() <= STENCIL_MAX && (double)nnz_cov >= 0.8 * (double)rows_cov * (double)V.size()) break;
        size_t least = 0;
        for (size_t c = 1; c < V.size(); ++c) if (use[c] < use[least]) least = c;
        V.erase(V.begin() + (long)least);
        use.erase(use.begin() + (long)least);
    }
    const int nu = (int)V.size();
    // uncovered rows as contiguous ranges; runs separated by fewer than 4096 covered rows are
    // merged (the rows in between then go with their range: cheaper than another launch)
    int nr = 0, ranges[STENCIL_RANGES][2];
    if (rows_cov < n) {
        std::vector<std::pair<long, long>> runs;
        for (long i = 0; i < n;) {
            if (covered[pat[i]]) { ++i; continue; }
            long j = i;
            while (j < n && !covered[pat[j]]) ++j;
            if (!runs.empty() && i - runs.back().second < 4096) runs.back().second = j;
            else runs.emplace_back(i, j);
            i = j;
        }
        if ((int)runs.size() > STENCIL_RANGES) return 0;
        for (auto &r : runs) { ranges[nr][0] = (int)r.first; ranges[nr][1] = (int)r.second; ++nr; }
    }
    // topological slot order over the covered patterns
    std::vector<unsigned> succ((size_t)nu, 0u);
    std::vector<int> indeg((size_t)nu, 0);
    for (int p = 0; p < npat; ++p) {
        if (!covered[p]) continue;
        for (int q = dptr[p] + 1; q < dptr[p + 1]; ++q) {
            const int a = id_in(V, doff[q - 1]), b = id_in(V, doff[q]);
            if (a == b) return 0;                                           // duplicate column in a row
            if (!((succ[a] >> b) & 1u)) { succ[a] |= 1u << b; ++indeg[b]; }
        }
    }
    std::vector<int> U;                             // offsets in slot order
    std::vector<int> slot_of((size_t)nu, -1);
    std::vector<char> done((size_t)nu, 0);
    for (int step = 0; step < nu; ++step) {
        int pick = -1;
        for (int c = 0; c < nu; ++c) if (!done[c] && indeg[c] == 0) { pick = c; break; }   // smallest offset first
        if (pick < 0) return 0;                                              // patterns disagree on the order
        done[pick] = 1;
        slot_of[pick] = (int)U.size();
        U.push_back(V[pick]);
        for (int c = 0; c < nu; ++c) if ((succ[pick] >> c) & 1u) --indeg[c];
    }
    std::vector<int> slot(doff.size(), -1);
    std::vector<unsigned> pmask((size_t)npat, 0u);
    for (int p = 0; p < npat; ++p) {
        if (!covered[p]) { pmask[p] = 0x80000000u; continue; }
        for (int q = dptr[p]; q < dptr[p + 1]; ++q) {
            slot[q] = slot_of[id_in(V, doff[q])];
            if (q > dptr[p] && slot[q] <= slot[q - 1]) return 0;
            pmask[p] |= 1u << slot[q];
        }
    }
    M.st_nu = nu;
    M.st_u0 = -1;
    for (int u = 0; u < nu; ++u) { M.st_off[u] = U[u]; if (U[u] == 0) M.st_u0 = u; }
    M.st_nranges = nr;
    for (int r = 0; r < nr; ++r) { M.st_range[r][0] = ranges[r][0]; M.st_range[r][1] = ranges[r][1]; }
    const size_t nblk = ((size_t)M.nrows + 255) / 256;
    int *dslot = nullptr;
    unsigned *dmask = nullptr;
    CHK(dev_alloc(&M.st_vals, nblk * nu * 256, acct));
    {
        unsigned char *mb = nullptr;             // 1 byte per row for |U| <= 7, 4 otherwise
        CHK(dev_alloc(&mb, (size_t)M.nrows * (nu <= 7 ? 1 : 4) + 16, acct));
        M.st_mask = mb;
    }
    CHK(dev_alloc(&dslot, slot.size(), nullptr));
    CHK(dev_alloc(&dmask, npat, nullptr));
    AMG_HIP(hipMemcpy(dslot, slot.data(), sizeof(int) * slot.size(), hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(dmask, pmask.data(), sizeof(unsigned) * pmask.size(), hipMemcpyHostToDevice));
    int rc = launch_stencil_build(M, dslot, dmask, 0);
    AMG_HIP(hipDeviceSynchronize());
    hipFree(dslot);
    hipFree(dmask);
    return rc;
}

// Offset-pattern analysis of a square operator (host, O(nnz)): succeeds when the rows use at most
// PAT_MAX distinct tuples of (column - row) with PAT_DICT_MAX offsets in total -- i.e. stencil
// operators.  On success the pattern ids and the dictionary go to HBM next to the CSR arrays.
int try_patterns(DevCsr &M, const int *Ap, const int *Aj, long *acct)
{
    const char *env = getenv("AMG_PATTERN");
    if (env && atoi(env) == 0) return 0;
    const int n = M.nrows;
    if (n > M.ncols || n < 1024) return 0;      // square, or a rank-local [owned | halo] operator
    std::vector<int> pat((size_t)n), dptr(1, 0), doff;
    std::vector<long> pcount;
    // Row ranges are analysed by host threads, each with a dictionary of its own (open-addressing table: hash of the
    // offset tuple -> local pattern id, in order of first occurrence); the dictionaries are then merged in range order,
    // which gives the ids the one-thread loop would give, and the rows' ids are renumbered in parallel.
    struct Local {
        std::vector<int> dptr{0}, doff;
        std::vector<long> count;
        bool overflow = false;
    };
    int nthreads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < (1 << 20)) nthreads = 1;
    std::vector<Local> loc((size_t)nthreads);
    auto analyse = [&](int t) {
        Local &Lc = loc[(size_t)t];
        const int lo = (int)((long)n * t / nthreads), hi = (int)((long)n * (t + 1) / nthreads);
        const int TBL = 1024;
        std::vector<int> tbl((size_t)TBL, -1);
        std::vector<uint64_t> thash((size_t)TBL, 0);
        for (int i = lo; i < hi; ++i) {
            const int s = Ap[i], e = Ap[i + 1], len = e - s;
            uint64_t hsh = 1469598103934665603ULL ^ (uint64_t)len;
            for (int k = s; k < e; ++k) { hsh ^= (uint64_t)(uint32_t)(Aj[k] - i); hsh *= 1099511628211ULL; }
            int slot = (int)(hsh & (TBL - 1)), id = -1;
            for (int probe = 0; probe < TBL; ++probe, slot = (slot + 1) & (TBL - 1)) {
                if (tbl[slot] < 0) break;
                if (thash[slot] != hsh) continue;
                const int c = tbl[slot];
                if (Lc.dptr[c + 1] - Lc.dptr[c] != len) continue;
                bool same = true;
                for (int q = 0; q < len && same; ++q) same = (Lc.doff[Lc.dptr[c] + q] == Aj[s + q] - i);
                if (same) { id = c; break; }
            }
            if (id < 0) {
                id = (int)Lc.dptr.size() - 1;
                if (id >= PAT_MAX || (int)Lc.doff.size() + len > PAT_DICT_MAX) { Lc.overflow = true; return; }   // not a stencil operator
                for (int k = s; k < e; ++k) Lc.doff.push_back(Aj[k] - i);
                Lc.dptr.push_back((int)Lc.doff.size());
                tbl[slot] = id;
                thash[slot] = hsh;
                Lc.count.push_back(0);
            }
            pat[i] = id;
            ++Lc.count[(size_t)id];
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(analyse, t);
        analyse(0);
        for (auto &th : pool) th.join();
    }
    for (const Local &Lc : loc) if (Lc.overflow) return 0;
    // merge in range order
    std::vector<std::vector<int>> remap((size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) {
        const Local &Lc = loc[(size_t)t];
        const int np = (int)Lc.dptr.size() - 1;
        remap[(size_t)t].resize((size_t)np);
        for (int c = 0; c < np; ++c) {
            const int len = Lc.dptr[c + 1] - Lc.dptr[c];
            int id = -1;
            for (int g = 0; g + 1 < (int)dptr.size() && id < 0; ++g) {
                if (dptr[g + 1] - dptr[g] != len) continue;
                bool same = true;
                for (int q = 0; q < len && same; ++q) same = (doff[dptr[g] + q] == Lc.doff[Lc.dptr[c] + q]);
                if (same) id = g;
            }
            if (id < 0) {
                id = (int)dptr.size() - 1;
                if (id >= PAT_MAX || (int)doff.size() + len > PAT_DICT_MAX) return 0;
                for (int q = 0; q < len; ++q) doff.push_back(Lc.doff[Lc.dptr[c] + q]);
                dptr.push_back((int)doff.size());
                pcount.push_back(0);
            }
            remap[(size_t)t][(size_t)c] = id;
            pcount[(size_t)id] += Lc.count[(size_t)c];
        }
    }
    if (nthreads > 1) {
        auto renumber = [&](int t) {
            const int lo = (int)((long)n * t / nthreads), hi = (int)((long)n * (t + 1) / nthreads);
            const std::vector<int> &rm = remap[(size_t)t];
            for (int i = lo; i < hi; ++i) pat[i] = rm[(size_t)pat[i]];
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(renumber, t);
        renumber(0);
        for (auto &th : pool) th.join();
    }
    M.npat = (int)dptr.size() - 1;
    M.ndict = (int)doff.size();
    // stride of the slowest grid axis = the widest offset of the pattern most rows share (halo
    // offsets of a rank-local operator are larger but belong to boundary rows only)
    M.period_rows = 0;
    {
        const int top = (int)(std::max_element(pcount.begin(), pcount.end()) - pcount.begin());
        for (int q = dptr[top]; q < dptr[top + 1]; ++q) M.period_rows = std::max(M.period_rows, std::abs(doff[q]));
    }
    CHK(dev_alloc(&M.pat, n, acct));
    CHK(dev_alloc(&M.dict_ptr, M.npat + 1, acct));
    CHK(dev_alloc(&M.dict_off, M.ndict, acct));
    AMG_HIP(hipMemcpy(M.pat, pat.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(M.dict_ptr, dptr.data(), sizeof(int) * dptr.size(), hipMemcpyHostToDevice));
    if (M.ndict) AMG_HIP(hipMemcpy(M.dict_off, doff.data(), sizeof(int) * doff.size(), hipMemcpyHostToDevice));
    return try_stencil(M, dptr, doff, pcount, pat, acct);
}

int upload_bsr(DevBsr &M, int nbrows, int bs, const int *Ap, const int *Aj, const double *Ax,
               long *acct)
{
    long nb = Ap[nbrows];
    M.nbrows = nbrows; M.bs = bs; M.nblocks = nb;
    CHK(dev_alloc(&M.Ap, nbrows + 1, acct));
    CHK(dev_alloc(&M.Aj, nb, acct));
    CHK(dev_alloc(&M.Ax, nb * bs * bs, acct));
    AMG_HIP(hipMemcpy(M.Ap, Ap, sizeof(int) * (size_t)(nbrows + 1), hipMemcpyHostToDevice));
    if (nb) {
        AMG_HIP(hipMemcpy(M.Aj, Aj, sizeof(int) * (size_t)nb, hipMemcpyHostToDevice));
        AMG_HIP(hipMemcpy(M.Ax, Ax, sizeof(double) * (size_t)(nb * bs * bs), hipMemcpyHostToDevice));
    }
    return 0;
}

// BSR (R x C blocks, row-major) -> scalar CSR keeping every stored entry, in
// the order scipy's bsr_matvec accumulates them (block by block, then bj).
void expand_bsr(int nbrows, int R, int C, const int *Ap, const int *Aj, const double *Ax,
                std::vector<int> &cp, std::vector<int> &cj, std::vector<double> &cx)
{
    long nb = Ap[nbrows];
    cp.assign((size_t)nbrows * R + 1, 0);
    cj.resize((size_t)nb * R * C);
    cx.resize((size_t)nb * R * C);
    long pos = 0;
    for (int i = 0; i < nbrows; ++i) {
        for (int r = 0; r < R; ++r) {
            cp[(size_t)i * R + r] = (int)pos;
            for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
                const double *blk = Ax + (long)jj * R * C + (long)r * C;
                for (int c = 0; c < C; ++c) {
                    cj[pos] = Aj[jj] * C + c;
                    cx[pos] = blk[c];
                    ++pos;
                }
            }
        }
    }
    cp[(size_t)nbrows * R] = (int)pos;
}

// ------------------------------------------------------------------ schedules
// host-side loops of the schedule builders over up to 16 threads: fn(first, last) on contiguous ranges of [0, n)
template <class F> static void host_parallel(long n, long grain, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    static const int cap = std::getenv("AMG_SETUP_THREADS") ? std::atoi(std::getenv("AMG_SETUP_THREADS")) : 16;
    long nt = std::min<long>(std::min<long>(hw ? hw : 1, std::max(1, cap)), (n + grain - 1) / std::max<long>(grain, 1));
    if (nt <= 1) { fn(0L, n); return; }
    std::vector<std::thread> th;
    const long step = (n + nt - 1) / nt;
    for (long t = 0; t < nt; ++t) {
        const long lo = t * step, hi = std::min(n, lo + step);
        if (lo >= hi) break;
        th.emplace_back([=]() { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

void Schedule::release()
{
    free_csr(G);
    if (rowmap) hipFree(rowmap);
    if (diagpos) hipFree(diagpos);
    if (rows) hipFree(rows);
    if (level_ptr_dev) hipFree(level_ptr_dev);
    free_bsr(Gb);
    flow.release();
    bflow.release();
    rowmap = diagpos = rows = level_ptr_dev = nullptr;
    for (int *p : {c2_code_f, c2_code_b, c2_off, perm_Aj, cl_code_f, cl_code_b}) if (p) hipFree(p);
    cl_code_f = cl_code_b = nullptr;
    for (double *p : {c2_diag, c2_val, c2_dummy, xp, bp, bd}) if (p) hipFree(p);
    c2_code_f = c2_code_b = c2_off = perm_Aj = nullptr;
    c2_diag = c2_val = c2_dummy = xp = bp = bd = nullptr;
    chain2 = perm = false;
    gb_level_slice.clear();
    chains.clear();
    chain_width.clear();
}

void Schedule::drop_level_copies()
{
    free_csr(G);
    G = DevCsr();
    for (int *p : {rowmap, diagpos, level_ptr_dev, c2_code_f, c2_code_b, c2_off, perm_Aj, cl_code_f, cl_code_b}) if (p) hipFree(p);
    for (double *p : {c2_diag, c2_val, c2_dummy, xp, bp, bd}) if (p) hipFree(p);
    rowmap = diagpos = level_ptr_dev = c2_code_f = c2_code_b = c2_off = perm_Aj = cl_code_f = cl_code_b = nullptr;
    c2_diag = c2_val = c2_dummy = xp = bp = bd = nullptr;
    chain2 = perm = chain_long = false;
    chains.clear(); chain_width.clear();
    std::vector<int>().swap(gp_host);
    level_copy_bytes = 0;
}

int build_levels(int n, const int *Ap, const int *Aj, const int *tasks, int ntasks,
                 std::vector<int> &level_ptr, std::vector<int> &order)
{
    std::vector<int> lastW((size_t)n, 0), lastR((size_t)n, 0), lvl((size_t)ntasks);
    int maxl = 0;
    for (int t = 0; t < ntasks; ++t) {
        int i = tasks ? tasks[t] : t;
        if (i < 0 || i >= n) { set_error("schedule: row index out of range"); return AMG_EINVAL; }
        int l = std::max(lastW[i], lastR[i]);
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            int j = Aj[jj];
            if (j != i && j >= 0 && j < n) l = std::max(l, lastW[j]);
        }
        l += 1;
        lvl[t] = l;
        lastW[i] = l;
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            int j = Aj[jj];
            if (j != i && j >= 0 && j < n) lastR[j] = std::max(lastR[j], l);
        }
        maxl = std::max(maxl, l);
    }
    level_ptr.assign((size_t)maxl + 1, 0);
    for (int t = 0; t < ntasks; ++t) level_ptr[lvl[t]]++;   // counts at index level (1-based)
    // prefix: level_ptr[l] = start of level l+1 ... convert counts to offsets
    {
        int run = 0;
        for (int l = 1; l <= maxl; ++l) { int c = level_ptr[l]; level_ptr[l - 1] = run; run += c; }
        level_ptr[maxl] = run;
    }
    order.resize((size_t)ntasks);
    std::vector<int> cur(level_ptr.begin(), level_ptr.end() - 1);
    for (int t = 0; t < ntasks; ++t) order[cur[lvl[t] - 1]++] = t;
    return 0;
}

int build_csr_schedule(const int *Ap, const int *Aj, const double *Ax, int n, const int *tasks,
                       int ntasks, Schedule &S, hipStream_t st, bool allow_flow, int ncols)
{
    (void)st;
    std::vector<int> order;
    CHK(build_levels(n, Ap, Aj, tasks, ntasks, S.level_ptr, order));
    S.ntasks = ntasks;
    std::vector<int> gp((size_t)ntasks + 1), rowmap((size_t)ntasks), dpos((size_t)ntasks);
    long nnz = 0;
    for (int k = 0; k < ntasks; ++k) {
        int t = order[k];
        int i = tasks ? tasks[t] : t;
        gp[k] = (int)nnz;
        nnz += Ap[i + 1] - Ap[i];
        rowmap[k] = i;
    }
    if (nnz > 2147483647L) { set_error("schedule: nnz exceeds int32"); return AMG_EINVAL; }
    gp[ntasks] = (int)nnz;
    std::vector<int> gj((size_t)nnz);
    std::vector<double> gx((size_t)nnz);
    host_parallel(ntasks, 1 << 15, [&](long klo, long khi) {
        for (long k = klo; k < khi; ++k) {
            int i = rowmap[(size_t)k];
            int len = Ap[i + 1] - Ap[i];
            std::memcpy(gj.data() + gp[(size_t)k], Aj + Ap[i], sizeof(int) * (size_t)len);
            std::memcpy(gx.data() + gp[(size_t)k], Ax + Ap[i], sizeof(double) * (size_t)len);
            int d = -1;
            for (int q = 0; q < len; ++q)
                if (Aj[Ap[i] + q] == i) d = gp[(size_t)k] + q;   // last diagonal entry wins, as relaxation.h:51-52
            dpos[(size_t)k] = d;
        }
    });
    CHK(upload_csr(S.G, ntasks, n, gp.data(), gj.data(), gx.data(), nullptr));
    S.gp_host = gp;
    CHK(dev_alloc(&S.rowmap, ntasks, (long *)nullptr));
    CHK(dev_alloc(&S.diagpos, ntasks, (long *)nullptr));
    AMG_HIP(hipMemcpy(S.rowmap, rowmap.data(), sizeof(int) * (size_t)ntasks, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(S.diagpos, dpos.data(), sizeof(int) * (size_t)ntasks, hipMemcpyHostToDevice));
    // runs of narrow levels for the chained sweep
    CHK(dev_alloc(&S.level_ptr_dev, (long)S.level_ptr.size(), (long *)nullptr));
    AMG_HIP(hipMemcpy(S.level_ptr_dev, S.level_ptr.data(), sizeof(int) * S.level_ptr.size(), hipMemcpyHostToDevice));
    S.chains.clear();
    S.chain_width.clear();
    // Measured (tools/gs_chain_ab.py): one workgroup beats a launch per level only when the levels are
    // really small -- a few hundred short rows (2-D 5/9-point operators: -12 % per cycle); on levels of
    // ~1000 rows or rows of 30 entries (SA coarse levels) it is 1.5-2.5x slower.  Hence the narrow window.
    const int nl = S.nlevels();
    const bool short_rows = ntasks > 0 && (double)S.G.nnz <= 10.0 * (double)ntasks;
    const int max_rows = std::min(512, gs_chain_max_rows());
    for (int l = 0; short_rows && l < nl;) {
        if (S.level_ptr[l + 1] - S.level_ptr[l] > max_rows) { ++l; continue; }
        int e = l;
        while (e < nl && S.level_ptr[e + 1] - S.level_ptr[e] <= max_rows) ++e;
        if (e - l >= 4) {
            // Pieces of one width class each (64 / 128 / 256 / 512 threads): the workgroup of a piece is sized to its
            // widest level, because idle waves still issue the whole pipeline.  A piece shorter than 16 levels is not
            // worth a launch of its own and joins its wider neighbour.
            auto cls = [&](int q) { const int w = S.level_ptr[q + 1] - S.level_ptr[q]; return w <= 64 ? 64 : (w <= 128 ? 128 : (w <= 256 ? 256 : 512)); };
            std::vector<std::array<int, 3>> pc;                              // first, last + 1, class
            for (int q = l; q < e; ++q) {
                const int c = cls(q);
                if (!pc.empty() && pc.back()[2] == c) pc.back()[1] = q + 1;
                else pc.push_back({q, q + 1, c});
            }
            for (bool merged = true; merged && pc.size() > 1;) {
                merged = false;
                for (size_t k = 0; k < pc.size(); ++k) {
                    if (pc[k][1] - pc[k][0] >= 16) continue;
                    // join the neighbour whose class is closer from above (or the only one)
                    size_t j;
                    if (k == 0) j = 1;
                    else if (k + 1 == pc.size()) j = k - 1;
                    else j = (pc[k - 1][2] >= pc[k][2] && (pc[k + 1][2] < pc[k][2] || pc[k - 1][2] <= pc[k + 1][2])) ? k - 1 : k + 1;
                    const size_t a0 = std::min(j, k), a1 = std::max(j, k);
                    pc[a0] = {pc[a0][0], pc[a1][1], std::max(pc[a0][2], pc[a1][2])};
                    pc.erase(pc.begin() + (long)a1);
                    merged = true;
                    break;
                }
                // neighbours that ended up in the same class are one piece
                for (size_t k = 0; k + 1 < pc.size();)
                    if (pc[k][2] == pc[k + 1][2]) { pc[k][1] = pc[k + 1][1]; pc.erase(pc.begin() + (long)k + 1); merged = true; }
                    else ++k;
            }
            for (const auto &q : pc)
                for (int p = q[0]; p < q[1]; p += CHAIN2_LMAX) {             // one launch each
                    S.chains.emplace_back(p, std::min(q[1], p + CHAIN2_LMAX));
                    S.chain_width.push_back(q[2]);
                }
        }
        l = e;
    }
    // Operators with LONG rows (coarse levels of a smoothed-aggregation hierarchy: 30-60 entries per row, a dozen to a
    // few hundred rows per dependency level, hundreds of levels): runs of levels that fit one workgroup entry-parallel
    // (gs_chainl_kernel).  Pieces of one workgroup size each (128 / 256 / 512 threads by the widest level's rows and
    // entries), pieces under 8 levels join a neighbour.
    S.chain_long = false;
    if (!short_rows && ntasks > 0 && gs_chain_max_rows() > 0) {
        const int KE = gs_chainl_entries_per_lane(), WMAX = gs_chainl_max_rows();
        auto need = [&](int q) {                                            // threads level q needs (> WMAX: not chained)
            const int r = S.level_ptr[q + 1] - S.level_ptr[q];
            const long e = (long)gp[(size_t)S.level_ptr[q + 1]] - gp[(size_t)S.level_ptr[q]];
            const long w = std::max<long>(r, (e + KE - 1) / KE);
            if (e < 4) return WMAX + 1;                                     // (the kernel requests whole quads of entries)
            return w <= 64 ? 64 : (w <= 128 ? 128 : (w <= 256 ? 256 : (w <= WMAX ? 512 : WMAX + 1)));
        };
        for (int l = 0; l < nl;) {
            if (need(l) > WMAX) { ++l; continue; }
            int e = l;
            while (e < nl && need(e) <= WMAX) ++e;
            if (e - l >= 4) {
                std::vector<std::array<int, 3>> pc;                          // first, last + 1, class
                for (int q = l; q < e; ++q) {
                    const int c = need(q);
                    if (!pc.empty() && pc.back()[2] == c) pc.back()[1] = q + 1;
                    else pc.push_back({q, q + 1, c});
                }
                for (bool merged = true; merged && pc.size() > 1;) {
                    merged = false;
                    for (size_t k = 0; k < pc.size(); ++k) {
                        if (pc[k][1] - pc[k][0] >= 8) continue;
                        const size_t j = (k == 0) ? 1 : ((k + 1 == pc.size()) ? k - 1 : (pc[k - 1][2] <= pc[k + 1][2] ? k - 1 : k + 1));
                        const size_t a0 = std::min(j, k), a1 = std::max(j, k);
                        pc[a0] = {pc[a0][0], pc[a1][1], std::max(pc[a0][2], pc[a1][2])};
                        pc.erase(pc.begin() + (long)a1);
                        merged = true;
                        break;
                    }
                    for (size_t k = 0; k + 1 < pc.size();)
                        if (pc[k][2] == pc[k + 1][2]) { pc[k][1] = pc[k + 1][1]; pc.erase(pc.begin() + (long)k + 1); merged = true; }
                        else ++k;
                }
                for (const auto &q : pc)
                    for (int p = q[0]; p < q[1]; p += gs_chainl_max_levels()) {      // one launch each
                        S.chains.emplace_back(p, std::min(q[1], p + gs_chainl_max_levels()));
                        S.chain_width.push_back(q[2]);
                    }
                S.chain_long = true;
            }
            l = e;
        }
        // LDS hand-off variant (gs_chainl2_kernel): per-entry codes -- the column, or ~(ring slot) for an operand produced
        // one or two levels earlier in the same launch.  Needs every unknown listed exactly once, columns inside, and no
        // zero diagonal in a chained row; otherwise the memory hand-off kernel sweeps the pieces.
        bool ring_ok = S.chain_long && ntasks == n;
        std::vector<int> lvl_of, pos_of, piece_of;
        if (ring_ok) {
            lvl_of.assign((size_t)n, -1); pos_of.assign((size_t)n, -1); piece_of.assign((size_t)nl, -1);
            for (int l = 0; l < nl && ring_ok; ++l)
                for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
                    const int i = rowmap[(size_t)k];
                    if (i < 0 || i >= n || lvl_of[(size_t)i] >= 0) { ring_ok = false; break; }
                    lvl_of[(size_t)i] = l; pos_of[(size_t)i] = k - S.level_ptr[l];
                }
            for (size_t c = 0; c < S.chains.size(); ++c)
                for (int l = S.chains[c].first; l < S.chains[c].second; ++l) piece_of[(size_t)l] = (int)c;
        }
        if (ring_ok) {
            std::vector<int> cf(gj), cb(gj);                                // default: the column (read from memory)
            const int RW = gs_chainl_max_rows();
            for (int l = 0; l < nl && ring_ok; ++l) {
                if (piece_of[(size_t)l] < 0) continue;
                for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1] && ring_ok; ++k) {
                    if (dpos[(size_t)k] < 0 || gx[(size_t)dpos[(size_t)k]] == 0.0) { ring_ok = false; break; }   // a row that keeps its value
                    for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) {
                        const int c = gj[(size_t)q];
                        if (c < 0 || c >= n) { ring_ok = false; break; }
                        const int lc = lvl_of[(size_t)c];
                        if (piece_of[(size_t)lc] != piece_of[(size_t)l]) continue;
                        const int dl = l - lc;                              // > 0: produced earlier in a forward sweep
                        const int slot = (lc % 3) * RW + pos_of[(size_t)c];
                        if (dl >= 1 && dl <= 2) cf[(size_t)q] = ~slot;
                        if (-dl >= 1 && -dl <= 2) cb[(size_t)q] = ~slot;
                    }
                }
            }
            if (ring_ok) {
                CHK(dev_alloc(&S.cl_code_f, (long)cf.size(), (long *)nullptr));
                CHK(dev_alloc(&S.cl_code_b, (long)cb.size(), (long *)nullptr));
                AMG_HIP(hipMemcpy(S.cl_code_f, cf.data(), sizeof(int) * cf.size(), hipMemcpyHostToDevice));
                AMG_HIP(hipMemcpy(S.cl_code_b, cb.data(), sizeof(int) * cb.size(), hipMemcpyHostToDevice));
            }
        }
    }
    // ---- the chained sweep's padded copy (Schedule::c2_*, gs_chain2_kernel)
    S.chain2 = false;
    S.perm = false;
    // The second-generation chain runs in LEVEL-ORDER numbering (unknown k = the k-th row of the schedule), which
    // needs the schedule to be a permutation of the unknowns of a square operator: every row listed exactly once and
    // no column outside (a partitioned level's halo columns are).
    bool permutation = !S.chains.empty() && !S.chain_long && max_rows <= CHAIN2_WG && ntasks == n;
    for (size_t q = 0; permutation && q < gj.size(); ++q) permutation = gj[q] >= 0 && gj[q] < n;
    if (permutation) {
        std::vector<int> lvl_of((size_t)n, -1), pos_of((size_t)n, -1), piece_of((size_t)nl, -1), inv((size_t)n, -1);
        bool ok = true;
        for (int l = 0; l < nl && ok; ++l)
            for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
                const int i = rowmap[(size_t)k];
                if (lvl_of[(size_t)i] >= 0) { ok = false; break; }          // a row listed twice: keep the first-generation chain
                lvl_of[(size_t)i] = l; pos_of[(size_t)i] = k - S.level_ptr[l]; inv[(size_t)i] = k;
            }
        for (size_t c = 0; c < S.chains.size(); ++c)
            for (int l = S.chains[c].first; l < S.chains[c].second; ++l) piece_of[(size_t)l] = (int)c;
        std::vector<int> coff((size_t)nl + 1, 0);
        for (int l = 0; l < nl; ++l) coff[(size_t)l + 1] = coff[(size_t)l] + (piece_of[(size_t)l] >= 0 ? S.level_ptr[l + 1] - S.level_ptr[l] : 0);
        const long total = coff[(size_t)nl];
        // slots per row: the longest off-diagonal row of the chained levels, rounded up to 4, 8 or 12
        int longest = 0;
        for (int l = 0; l < nl && ok; ++l) {
            if (piece_of[(size_t)l] < 0) continue;
            for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
                int cnt = 0;
                for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) cnt += (gj[(size_t)q] != rowmap[(size_t)k]) ? 1 : 0;
                longest = std::max(longest, cnt);
            }
        }
        if (longest > CHAIN2_PF) ok = false;
        const int PFs = longest <= 4 ? 4 : (longest <= 8 ? 8 : 12);
        S.c2_pf = PFs;
        if ((double)total * PFs * 8.0 >= 4.0e9 || (double)n * 8.0 >= 4.0e9) ok = false;     // the kernel addresses with 32-bit byte offsets
        // padded slots: value 0, operand = the permanent 0.0 kept behind the last unknown of xp
        std::vector<int> cf((size_t)total * PFs, n), cb((size_t)total * PFs, n);
        std::vector<double> cd((size_t)n, 0.0), cv((size_t)total * PFs, 0.0);        // cd: the diagonal by level-order position
        for (int l = 0; l < nl && ok; ++l) {
            if (piece_of[(size_t)l] < 0) continue;
            const int base = coff[(size_t)l], cnt = S.level_ptr[l + 1] - S.level_ptr[l];
            for (int tt = 0; tt < cnt && ok; ++tt) {
                const int k = S.level_ptr[l] + tt, i = rowmap[(size_t)k];
                cd[(size_t)k] = dpos[(size_t)k] >= 0 ? gx[(size_t)dpos[(size_t)k]] : 0.0;
                if (cd[(size_t)k] == 0.0) { ok = false; break; }    // a row that keeps its value (relaxation.h:58-60): first-generation chain
                int u = 0;
                for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) {
                    const int c = gj[(size_t)q];
                    if (c == i) continue;                                   // the diagonal is not part of the sum
                    if (u >= PFs) { ok = false; break; }
                    // codes: QUADS of slots side by side (one 16-byte load fetches four): quad-major, then row, then slot % 4
                    const size_t at = (size_t)PFs * base + (size_t)(u / 4) * 4 * cnt + 4 * (size_t)tt + (size_t)(u & 3);
                    // values: PAIRS of slots side by side (one 16-byte load fetches two): pair-major, then row, then slot parity
                    cv[(size_t)PFs * base + (size_t)(u / 2) * 2 * cnt + 2 * (size_t)tt + (size_t)(u & 1)] = gx[(size_t)q];
                    int f = inv[(size_t)c], bk = inv[(size_t)c];             // default: settled in memory, read at its level-order position
                    if (piece_of[(size_t)lvl_of[(size_t)c]] == piece_of[(size_t)l]) {
                        const int dl = l - lvl_of[(size_t)c];                // > 0: produced earlier in a forward sweep
                        // produced within the last CHAIN2_D levels of the sweep: the operand is read from the ring
                        // buffer its level wrote, (level % (D + 1)) * 512 + position in the level
                        const int slot = (lvl_of[(size_t)c] % (CHAIN2_D + 1)) * CHAIN2_WG + pos_of[(size_t)c];
                        if (dl >= 1 && dl <= CHAIN2_D) f = ~(slot * 8);      // ~(byte offset into the ring)
                        if (-dl >= 1 && -dl <= CHAIN2_D) bk = ~(slot * 8);
                    }
                    cf[at] = f; cb[at] = bk;
                    ++u;
                }
            }
        }
        if (ok && total > 0) {
            CHK(dev_alloc(&S.c2_diag, n, (long *)nullptr));
            CHK(dev_alloc(&S.c2_val, total * PFs, (long *)nullptr));
            CHK(dev_alloc(&S.c2_code_f, total * PFs, (long *)nullptr));
            CHK(dev_alloc(&S.c2_code_b, total * PFs, (long *)nullptr));
            CHK(dev_alloc(&S.c2_off, nl + 1, (long *)nullptr));
            CHK(dev_alloc(&S.c2_dummy, CHAIN2_WG, (long *)nullptr));
            AMG_HIP(hipMemcpy(S.c2_diag, cd.data(), sizeof(double) * cd.size(), hipMemcpyHostToDevice));
            AMG_HIP(hipMemcpy(S.c2_val, cv.data(), sizeof(double) * cv.size(), hipMemcpyHostToDevice));
            AMG_HIP(hipMemcpy(S.c2_code_f, cf.data(), sizeof(int) * cf.size(), hipMemcpyHostToDevice));
            AMG_HIP(hipMemcpy(S.c2_code_b, cb.data(), sizeof(int) * cb.size(), hipMemcpyHostToDevice));
            AMG_HIP(hipMemcpy(S.c2_off, coff.data(), sizeof(int) * coff.size(), hipMemcpyHostToDevice));
            // the level-ordered copy's columns in level-order numbering (for the per-level launches and the first-
            // generation chain of the same sweep) and the gathered x / b the sweep works on
            std::vector<int> pj(gj.size());
            for (size_t q = 0; q < gj.size(); ++q) pj[q] = inv[(size_t)gj[q]];
            CHK(dev_alloc(&S.perm_Aj, (long)pj.size(), (long *)nullptr));
            CHK(dev_alloc(&S.xp, (long)n + 1, (long *)nullptr));
            AMG_HIP(hipMemset(S.xp, 0, sizeof(double) * ((size_t)n + 1)));       // xp[n] stays 0.0: the operand of padded slots
            CHK(dev_alloc(&S.bp, n, (long *)nullptr));
            CHK(dev_alloc(&S.bd, 2 * (long)n, (long *)nullptr));
            AMG_HIP(hipMemcpy(S.perm_Aj, pj.data(), sizeof(int) * pj.size(), hipMemcpyHostToDevice));
            S.chain2 = true;
            S.perm = true;
        }
    }
    // ---- dataflow form (gsflow.hip): one persistent launch per smoother application
    S.flow_auto = false;
    S.level_copy_bytes = 12L * (long)nnz + 12L * ntasks;
    if (allow_flow && gs_flow_mode() != 0 && ntasks == n) {
        // by default only where levels would otherwise be launches of their own or long-row chains: the sweeps of
        // narrow short-row levels (2-D operators) hand their values on through LDS faster than any memory hand-off
        int chained = 0;
        for (const auto &c : S.chains) chained += c.second - c.first;
        // ... and not where a level holds hundreds of thousands of rows (multicolour orderings, the 500^3 level): a launch
        // per level then runs at streaming speed and the second copy of the operator would only cost memory
        const bool wanted = (S.chain_long || chained * 10 < nl * 9) && (long)ntasks < (long)nl * 200000L;
        if (wanted || gs_flow_mode() == 2) {
            CHK(build_flow_form(S.flow, n, ntasks, S.level_ptr, rowmap, gp, gj, gx, ncols));
            S.flow_auto = wanted && S.flow.ready;
            // the default serves this schedule from the dataflow form alone: the level-ordered copies of the other
            // paths (12-28 B per entry) are not kept beside it
            if (S.flow_auto && gs_flow_mode() == 1) S.drop_level_copies();
        }
    }
    return 0;
}

int build_block_schedule(const int *Ap, const int *Aj, int nb, const int *tasks, int ntasks,
                         Schedule &S, hipStream_t st, const double *Ax, int bs, bool independent, bool block_flow)
{
    (void)st;
    std::vector<int> order;
    if (independent) {
        // Jacobi-type passes: the listed block rows do not depend on each other -- one level, in list order
        S.level_ptr.assign({0, ntasks});
        order.resize((size_t)ntasks);
        for (int t = 0; t < ntasks; ++t) order[(size_t)t] = t;
        for (int t = 0; t < ntasks; ++t) {
            const int i = tasks ? tasks[t] : t;
            if (i < 0 || i >= nb) { set_error("schedule: row index out of range"); return AMG_EINVAL; }
        }
    } else {
        CHK(build_levels(nb, Ap, Aj, tasks, ntasks, S.level_ptr, order));
    }
    S.ntasks = ntasks;
    std::vector<int> rows((size_t)ntasks);
    for (int k = 0; k < ntasks; ++k) rows[k] = tasks ? tasks[order[k]] : order[k];
    CHK(dev_alloc(&S.rows, ntasks, (long *)nullptr));
    AMG_HIP(hipMemcpy(S.rows, rows.data(), sizeof(int) * (size_t)ntasks, hipMemcpyHostToDevice));
    if (Ax && bs > 0) {
        // block rows copied in level order so that every level is one contiguous, streamable slice
        const long B2 = (long)bs * bs;
        std::vector<int> gp((size_t)ntasks + 1);
        long nblk = 0;
        for (int k = 0; k < ntasks; ++k) { gp[k] = (int)nblk; nblk += Ap[rows[k] + 1] - Ap[rows[k]]; }
        gp[ntasks] = (int)nblk;
        std::vector<int> gj((size_t)nblk);
        std::vector<double> gx((size_t)(nblk * B2));
        host_parallel(ntasks, 1 << 14, [&](long klo, long khi) {
            for (long k = klo; k < khi; ++k) {
                int i = rows[(size_t)k], len = Ap[i + 1] - Ap[i];
                std::memcpy(gj.data() + gp[(size_t)k], Aj + Ap[i], sizeof(int) * (size_t)len);
                std::memcpy(gx.data() + (long)gp[(size_t)k] * B2, Ax + (long)Ap[i] * B2, sizeof(double) * (size_t)(len * B2));
            }
        });
        // block Gauss-Seidel: the dataflow form (one persistent launch per smoother application) serves the schedule
        // alone; otherwise the level-ordered copy, and for the levels of a large operator its sliced block form
        if (block_flow && !independent && gs_flow_mode() != 0 && (bs == 2 || bs == 3))
            CHK(build_block_flow_form(S.bflow, nb, bs, ntasks, S.level_ptr, rows, gp, gj, gx));
        if (!S.bflow.ready) {
            long acct = 0;
            CHK(upload_bsr(S.Gb, ntasks, bs, gp.data(), gj.data(), gx.data(), &acct));
            if (!independent) CHK(build_bsell_levels(S.Gb, S.level_ptr, S.rows, S.gb_level_slice, &acct));
            S.level_copy_bytes = acct;
        }
    }
    return 0;
}

// ------------------------------------------------------------------ operator application
static StreamArgs base_args(const DevCsr &M)
{
    StreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Ap = M.Ap; a.Aj = M.Aj; a.Ax = M.Ax;
    a.row_lo = 0; a.row_hi = M.nrows;
    a.nnz_total = M.nnz;
    a.rows_per_wg = rows_per_wg_for(M.nnz, M.nrows);
    if (M.Aj16 && M.i16_rpb == a.rows_per_wg) { a.Aj16 = M.Aj16; a.wg_base = M.wg_base; a.wg_flag = M.wg_flag; }
    return a;
}

// the storage form an operator application runs from: stencil, offset-pattern or plain CSR
static BsrStreamArgs block_spmv_args(const DevBsr &B, StreamMode mode, const StreamArgs &a)
{
    BsrStreamArgs q;
    std::memset(&q, 0, sizeof(q));
    q.Ap = B.Ap; q.Aj = B.Aj; q.Ax = B.Ax; q.bs = B.bs;
    q.brow_lo = 0; q.brow_hi = B.nbrows;
    q.xin = a.xg; q.xout = a.out; q.b = a.b; q.v2 = a.v2; q.c0 = a.c0; q.out2 = a.out2;
    q.gscale = (a.gscale == 0.0) ? 1.0 : a.gscale;
    q.smode = (int)mode;
    return q;
}

// an operator kept by its square blocks only (no expanded CSR copy) must be applied from them
static bool applies_from_blocks(const DevCsr &M, StreamMode mode, const StreamArgs &a)
{
    if (!(M.blk && M.blk->Ap)) return false;
    if (!M.Ap) return true;
    return bsr_spmv_enabled(M.blk->bs) && bsr_spmv_supports(mode) && a.row_lo == 0 && a.row_hi == M.nrows;
}

int apply_operator(const DevCsr &M, StreamMode mode, const StreamArgs &a, hipStream_t st)
{
    if (applies_from_blocks(M, mode, a)) {
        if (!bsr_spmv_supports(mode) || a.row_lo != 0 || a.row_hi != M.nrows) {
            set_error("this operator is stored by blocks only: the requested application needs its scalar expansion");
            return AMG_ENOTIMPL;
        }
        const BsrStreamArgs q = block_spmv_args(*M.blk, mode, a);
        if (bsell_applies(*M.blk, BM_SPMV, q)) return launch_bsell(*M.blk, BM_SPMV, q, st);
        return launch_bsr_stream(BM_SPMV, q, M.blk->nblocks, st);
    }
    if (M.st_vals && (stencil_enabled() || !M.Ap) && pattern_supports(mode)) return launch_stencil(mode, a, M, st);
    if (M.pat && pattern_supports(mode)) return launch_pattern(mode, a, M, st);
    if (M.sl_val && (sell_enabled() || !M.Ap) && sell_supports(mode) && a.row_lo == M.sl_lo && a.row_hi == M.sl_hi && !a.rowmap && a.Aj == M.Aj)
        return launch_sell(mode, a, M, st);
    if (!M.Ap || !a.Ap) {
        set_error("this operator's CSR arrays were released (amg_hier_release_sources): the requested application needs them");
        return AMG_ESTATE;
    }
    return launch_stream(mode, a, st);
}

int spmv(const DevCsr &M, StreamMode mode, const double *xg, const double *b, const double *v2,
         double *out, double *out2, double c0, hipStream_t st)
{
    StreamArgs a = base_args(M);
    a.xg = xg; a.b = b; a.v2 = v2; a.out = out; a.out2 = out2; a.c0 = c0;
    return apply_operator(M, mode, a, st);
}

// directional sweeps of a scheduled (CSR flavour) Gauss-Seidel, in the order given (seq[k] != 0: reversed).  With the
// second-generation chain the sweeps run in LEVEL-ORDER numbering on gathered copies of x and b (one gather before,
// one scatter after the whole sequence): every access of a level is then contiguous over the lanes.
int gs_sweep_csr(const Schedule &S, bool bsr1, double *x, const double *b, const unsigned char *seq, int nseq,
                 hipStream_t st, bool allow_flow)
{
    if (nseq <= 0) return 0;
    if (S.flow.ready && (!S.G.Ap || (allow_flow && (gs_flow_mode() == 2 || (gs_flow_mode() == 1 && S.flow_auto)))))
        return gs_flow_sweep(S.flow, bsr1, x, b, seq, nseq, st);
    const bool perm = S.perm && gs_chain_enabled() && gs_chain_generation() == 2;
    DevCsr G = S.G;
    G.owned = false;
    const int *rowmap = S.rowmap;
    double *xs = x;
    const double *bs = b;
    if (perm) {
        CHK(launch_perm_gather(S.rowmap, x, b, S.c2_diag, S.xp, S.bp, S.bd, S.ntasks, st));
        G.Aj = S.perm_Aj; rowmap = nullptr; xs = S.xp; bs = S.bp;
    }
    StreamArgs a = base_args(G);
    a.xg = xs; a.b = bs; a.out = xs; a.rowmap = rowmap; a.diagpos = S.diagpos;
    const int nl = S.nlevels();
    for (int k = 0; k < nseq; ++k) {
        const bool reverse = seq[k] != 0;
        auto launches = [&](int l0, int l1) -> int {          // levels [l0, l1), one launch each, in sweep order
            for (int q = 0; q < l1 - l0; ++q) {
                const int l = reverse ? l1 - 1 - q : l0 + q;
                a.row_lo = S.level_ptr[l];
                a.row_hi = S.level_ptr[l + 1];
                CHK(launch_gs_level(a, bsr1, S.gp_host.empty() ? nullptr : S.gp_host.data(), st));
            }
            return 0;
        };
        if (!gs_chain_enabled() || S.chains.empty()) { CHK(launches(0, nl)); continue; }
        // segments in sweep order: wide levels by launches, runs of narrow levels by one chained launch each
        const int nc = (int)S.chains.size();
        int pos = reverse ? nl : 0;
        for (int c = 0; c < nc; ++c) {
            const auto &ch = S.chains[(size_t)(reverse ? nc - 1 - c : c)];
            const int width = S.chain_width[(size_t)(reverse ? nc - 1 - c : c)];
            if (!reverse) { CHK(launches(pos, ch.first)); pos = ch.second; }
            else { CHK(launches(ch.second, pos)); pos = ch.first; }
            if (perm)
                CHK(launch_gs_chain2(S.level_ptr_dev, S.c2_val, reverse ? S.c2_code_b : S.c2_code_f, S.c2_off, S.c2_dummy, S.c2_pf,
                                     ch.first, ch.second - ch.first, width, reverse, bsr1, xs, S.bd, S.ntasks, st));
            else if (S.chain_long)
                CHK(launch_gs_chain_long(G, rowmap, S.diagpos, S.level_ptr_dev, ch.first, ch.second - ch.first, width, reverse, bsr1, xs, bs, st,
                                         gs_chain_generation() == 2 ? (reverse ? S.cl_code_b : S.cl_code_f) : nullptr));
            else
                CHK(launch_gs_chain(G, rowmap, S.diagpos, S.level_ptr_dev, ch.first, ch.second - ch.first, width, reverse, bsr1, xs, bs, st));
        }
        if (!reverse) CHK(launches(pos, nl));
        else CHK(launches(0, pos));
    }
    if (perm) CHK(launch_perm_scatter(S.rowmap, S.xp, x, S.ntasks, st));
    return 0;
}

int gs_sweep_csr(const Schedule &S, bool bsr1, double *x, const double *b, bool reverse, hipStream_t st, bool allow_flow)
{
    const unsigned char r = reverse ? 1 : 0;
    return gs_sweep_csr(S, bsr1, x, b, &r, 1, st, allow_flow);
}

// one directional pass over the block rows of a schedule built WITH values (Schedule::Gb: the rows copied in level
// order, so that every level is one contiguous slice streamed by bsr_stream_kernel); xin != x for Jacobi-type
// passes (the operand vector is then the frozen copy), omega for the Jacobi blends
int sweep_block_schedule(const Schedule &S, BlockMode mode, const double *Dinv, const double *xin, double *x, const double *b,
                         double omega, bool reverse, hipStream_t st)
{
    if (!S.Gb.Ap) { set_error("block schedule holds no operator copy"); return AMG_ESTATE; }
    const int nl = S.nlevels();
    BsrStreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Ap = S.Gb.Ap; a.Aj = S.Gb.Aj; a.Ax = S.Gb.Ax; a.bs = S.Gb.bs;
    a.rowmap = S.rows; a.intra_reverse = reverse ? 1 : 0;
    a.xin = xin; a.xout = x; a.b = b; a.Dinv = Dinv; a.omega = omega;
    const bool sliced = mode == BM_BLOCK_GS && !S.gb_level_slice.empty() && S.Gb.bsl_val && bsell_level_enabled();
    for (int q = 0; q < nl; ++q) {
        int l = reverse ? nl - 1 - q : q;
        a.brow_lo = S.level_ptr[l];
        a.brow_hi = S.level_ptr[l + 1];
        if (sliced) CHK(launch_bsell_level(S.Gb, mode, a, S.gb_level_slice[(size_t)l], S.gb_level_slice[(size_t)l + 1], st));
        else CHK(launch_bsr_stream(mode, a, (long)S.Gb.nblocks * (a.brow_hi - a.brow_lo) / (S.ntasks ? S.ntasks : 1), st));
    }
    return 0;
}

static int gs_sweep_block(const Schedule &S, const DevBsr &Ab, BlockMode mode, const double *Dinv,
                          double *x, const double *b, bool reverse, hipStream_t st)
{
    (void)Ab;
    return sweep_block_schedule(S, mode, Dinv, x, x, b, 1.0, reverse, st);
}

int block_gs_sweeps(const Schedule &S, const double *Dinv, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st,
                    bool allow_flow)
{
    (void)allow_flow;                  // (a schedule built with the dataflow form holds nothing else)
    if (S.bflow.ready) return block_flow_sweep(S.bflow, Dinv, x, b, seq, nseq, st);
    for (int k = 0; k < nseq; ++k) CHK(sweep_block_schedule(S, BM_BLOCK_GS, Dinv, x, x, b, 1.0, seq[k] != 0, st));
    return 0;
}

static int bsr_stream_all(const DevBsr &Ab, BlockMode mode, const double *Dinv, const double *xin, double *xout,
                          const double *b, double omega, hipStream_t st)
{
    BsrStreamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Ap = Ab.Ap; a.Aj = Ab.Aj; a.Ax = Ab.Ax; a.bs = Ab.bs;
    a.brow_lo = 0; a.brow_hi = Ab.nbrows;
    a.xin = xin; a.xout = xout; a.b = b; a.Dinv = Dinv; a.omega = omega;
    if (bsell_applies(Ab, mode, a)) return launch_bsell(Ab, mode, a, st);
    return launch_bsr_stream(mode, a, Ab.nblocks, st);
}

}  // namespace amg

using namespace amg;

// Dataflow Gauss-Seidel sweeps are persistent launches whose waves wait for each other: fine for one process per device.
// A partitioned hierarchy uses them when every rank has a device of its own (world <= devices of the node; halo columns
// are frozen operands of the form), not when ranks share one.  AMG_DIST_FLOW=0 | 1 overrides.
static bool flow_allowed(const amg_hier *h)
{
    if (!h->comm) return true;
    const char *e = std::getenv("AMG_DIST_FLOW");
    if (e) return std::atoi(e) != 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return false;
    return h->comm->world <= ndev;
}

// ------------------------------------------------------------------ row-partitioned levels
// refresh the halo part of v (a level-lvl vector of n_own + n_halo entries) from its owners
static int exchange(amg_hier *h, Level &L, double *v)
{
    if (!h->comm || L.part.channel < 0) return 0;
    double *halo = v + L.part.n_own;
    CHK(comm_exchange_begin(h->comm, L.part.channel, v, L.part.send_idx, halo, h->stream));
    return comm_exchange_end(h->comm, L.part.channel, halo, h->stream);
}

// A_l applied with xg as the gathered operand.  On a partitioned level the halo of xg is refreshed first; when the
// level has a large halo-free row window, that window is computed while the halo is in flight.
static int level_apply(amg_hier *h, Level &L, StreamMode mode, const double *xg, double gscale, const double *b,
                       const double *v2, double *out, double c0)
{
    hipStream_t st = h->stream;
    StreamArgs a = base_args(L.A);
    a.xg = xg; a.b = b; a.v2 = v2; a.out = out; a.c0 = c0; a.gscale = gscale;
    if (!h->comm || L.part.channel < 0) return apply_operator(L.A, mode, a, st);
    double *v = const_cast<double *>(xg);          // every gathered vector is a level work vector
    if (!(h->overlap && L.part.overlap)) {
        CHK(exchange(h, L, v));
        return apply_operator(L.A, mode, a, st);
    }
    double *halo = v + L.part.n_own;
    CHK(comm_exchange_begin(h->comm, L.part.channel, v, L.part.send_idx, halo, st));
    a.row_lo = L.part.i0; a.row_hi = L.part.i1;
    CHK(apply_operator(L.A, mode, a, st));
    CHK(comm_exchange_end(h->comm, L.part.channel, halo, st));
    if (L.part.i0 > 0) { a.row_lo = 0; a.row_hi = L.part.i0; CHK(apply_operator(L.A, mode, a, st)); }
    if (L.part.i1 < L.A.nrows) { a.row_lo = L.part.i1; a.row_hi = L.A.nrows; CHK(apply_operator(L.A, mode, a, st)); }
    return 0;
}

// sqrt of the sum over ALL ranks of the squares of v[0..n) into *slot (single GPU: the plain 2-norm)
static int global_norm(amg_hier *h, const double *v, long n, double *slot)
{
    if (!h->comm) return launch_norm2(v, n, h->norm_scratch, slot, h->stream);
    double *partial = h->norm_scratch + 1031;
    CHK(launch_dot(v, v, n, h->norm_scratch, partial, h->stream));
    return comm_allreduce_sqrt(h->comm, h->reduce_channel, partial, slot, h->stream);
}

// ------------------------------------------------------------------ relaxation on a level
// x may be swapped with the level's alternate buffer (Jacobi writes out of place).
static int relax(amg_hier *h, Level &L, Smoother &s, double *&x, double *&xalt, const double *b,
                 bool x_zero, bool r_ready = false)
{
    hipStream_t st = h->stream;
    const int n = L.A.nrows;
    const bool bsr = (L.fmt == AMG_FMT_BSR);
    switch (s.kind) {
    case AMG_SM_NONE:
        return 0;
    case AMG_SM_CALLBACK: {
        if (!s.cb) { set_error("callback smoother without a callback"); return AMG_ESTATE; }
        const int lvl = (int)(&L - &h->lv[0]);
        if (s.cb(s.cb_user, lvl, x, b) != 0) { if (last_error().empty()) set_error("callback smoother failed"); return AMG_ESTATE; }
        return 0;
    }
    case AMG_SM_JACOBI:
        // relaxation.py:357-427
        for (int it = 0; it < s.iterations; ++it) {
            if (bsr && L.R > 1) {
                // bsr_jacobi: temp = x (relaxation.h:303-305), then x updated in place
                AMG_HIP(hipMemcpyAsync(xalt, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
                CHK(bsr_stream_all(L.Ab, BM_BSR_JACOBI, nullptr, xalt, x, b, s.omega, st));
            } else {
                CHK(level_apply(h, L, bsr ? SM_JACOBI_BSR1 : SM_JACOBI, x, 0.0, b, x, xalt, s.omega));
                std::swap(x, xalt);
            }
        }
        return 0;
    case AMG_SM_GAUSS_SEIDEL:
    case AMG_SM_GAUSS_SEIDEL_INDEXED:
    case AMG_SM_SOR: {
        // relaxation.py:280-354 (gauss_seidel), :108-169 (sor), :671-741 (indexed)
        const bool point_block = bsr && L.R > 1 && s.kind != AMG_SM_GAUSS_SEIDEL_INDEXED;
        auto sweep_once = [&](bool reverse) -> int {
            // partitioned level: HYBRID sweep -- the halo is refreshed once per directional sweep and frozen during
            // it (Gauss-Seidel inside a rank, Jacobi across ranks: BASELINE configuration C4)
            CHK(exchange(h, L, x));
            if (point_block)
                return gs_sweep_block(*s.sched, L.Ab, BM_BSR_GS, nullptr, x, b, reverse, st);
            // (ranks that SHARE a device -- tests, rehearsals -- keep the level-scheduled sweeps: persistent dataflow
            //  launches of different processes could keep each other's waves from becoming resident; flow_allowed())
            return gs_sweep_csr(*s.sched, bsr && s.kind != AMG_SM_GAUSS_SEIDEL_INDEXED, x, b, reverse, st, flow_allowed(h));
        };
        auto gs = [&](int iterations, int sweep) -> int {
            if (!point_block && !h->comm) {
                // all directional sweeps of this application in one call (level-order numbering entered once)
                std::vector<unsigned char> seq;
                for (int it = 0; it < iterations; ++it) {
                    if (sweep == AMG_SWEEP_FORWARD) seq.push_back(0);
                    else if (sweep == AMG_SWEEP_BACKWARD) seq.push_back(1);
                    else { seq.push_back(0); seq.push_back(1); }
                }
                return gs_sweep_csr(*s.sched, bsr && s.kind != AMG_SM_GAUSS_SEIDEL_INDEXED, x, b, seq.data(), (int)seq.size(), st, true);
            }
            for (int it = 0; it < iterations; ++it) {
                if (sweep == AMG_SWEEP_FORWARD) CHK(sweep_once(false));
                else if (sweep == AMG_SWEEP_BACKWARD) CHK(sweep_once(true));
                else { CHK(sweep_once(false)); CHK(sweep_once(true)); }
            }
            return 0;
        };
        if (s.kind != AMG_SM_SOR) return gs(s.iterations, s.sweep);
        for (int it = 0; it < s.iterations; ++it) {
            AMG_HIP(hipMemcpyAsync(xalt, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
            CHK(gs(1, s.sweep));
            CHK(launch_sor_combine(x, xalt, s.omega, n, st));
        }
        return 0;
    }
    case AMG_SM_POLYNOMIAL: {
        // relaxation.py:593-668.  r = b when x == 0 is bitwise identical to b - A*0.
        // h = c0*r is never stored: the next operator application gathers c0*r[j] from r on the fly
        // (the same single rounding), which saves one vector write per application (and the whole
        // scale pass when x == 0).
        const int nc = (int)s.coef.size();
        for (int it = 0; it < s.iterations; ++it) {
            const double *r;
            if (x_zero) {
                r = b;
            } else if (r_ready && it == 0) {
                r = L.r;                       // b - A x of this very x, left there by residual_norm_to
            } else {
                CHK(level_apply(h, L, SM_RESIDUAL, x, 0.0, b, nullptr, L.r, 0.0));
                r = L.r;
            }
            if (nc == 1) {
                CHK(launch_axpy_scaled(x, r, s.coef[0], n, st));
            } else if (nc == 2) {
                CHK(level_apply(h, L, SM_POLY_LAST, r, s.coef[0], r, x, x, s.coef[1]));
            } else {
                double *hh = L.h, *hn = L.h2;
                CHK(level_apply(h, L, SM_POLY_STEP, r, s.coef[0], r, nullptr, hh, s.coef[1]));
                for (int c = 2; c < nc - 1; ++c) {
                    CHK(level_apply(h, L, SM_POLY_STEP, hh, 0.0, r, nullptr, hn, s.coef[c]));
                    std::swap(hh, hn);
                }
                CHK(level_apply(h, L, SM_POLY_LAST, hh, 0.0, r, x, x, s.coef[nc - 1]));
            }
            x_zero = false;
        }
        return 0;
    }
    default:
        break;
    }
    if (h->comm && L.part.channel >= 0) {
        set_error("this smoother has no row-partitioned form (offered: jacobi, polynomial, gauss_seidel / indexed as hybrid sweeps)");
        return AMG_ENOTIMPL;
    }
    switch (s.kind) {
    case AMG_SM_GAUSS_SEIDEL_NE: {
        // relaxation.py:821-908 (Kaczmarz sweep over the rows)
        const std::vector<int> &lp = s.sw_level_ptr;
        const int nl = (int)lp.size() - 1;
        auto sweep_once = [&](bool reverse) -> int {
            for (int q = 0; q < nl; ++q) {
                const int l = reverse ? nl - 1 - q : q;
                CHK(launch_gs_ne_level(L.A.Ap, L.A.Aj, L.A.Ax, x, b, s.Dinv, s.omega, s.sw_order + lp[l], lp[l + 1] - lp[l], st));
            }
            return 0;
        };
        for (int it = 0; it < s.iterations; ++it) {
            if (s.sweep == AMG_SWEEP_FORWARD || s.sweep == AMG_SWEEP_SYMMETRIC) CHK(sweep_once(false));
            if (s.sweep == AMG_SWEEP_BACKWARD || s.sweep == AMG_SWEEP_SYMMETRIC) CHK(sweep_once(true));
        }
        return 0;
    }
    case AMG_SM_GAUSS_SEIDEL_NR: {
        // relaxation.py:911-997: every call starts from r = b - A x (a CSC product in the reference:
        // terms in ascending column order = a row of A with sorted indices), then sweeps the columns,
        // keeping r current; a symmetric sweep is a forward CALL followed by a backward CALL
        const std::vector<int> &lp = s.sw_level_ptr;
        const int nl = (int)lp.size() - 1;
        const DevCsr &Arow = s.aux[1].Ap ? s.aux[1] : L.A;
        const DevCsr &Acol = s.aux[0];
        auto call = [&](bool reverse, int iterations) -> int {
            CHK(spmv(Arow, SM_RESIDUAL, x, b, nullptr, L.r, nullptr, 0.0, st));
            for (int it = 0; it < iterations; ++it)
                for (int q = 0; q < nl; ++q) {
                    const int l = reverse ? nl - 1 - q : q;
                    CHK(launch_gs_nr_level(Acol.Ap, Acol.Aj, Acol.Ax, x, L.r, s.Dinv, s.omega, s.sw_order + lp[l],
                                           lp[l + 1] - lp[l], st));
                }
            return 0;
        };
        if (s.sweep == AMG_SWEEP_SYMMETRIC) {
            for (int it = 0; it < s.iterations; ++it) { CHK(call(false, 1)); CHK(call(true, 1)); }
        } else {
            CHK(call(s.sweep == AMG_SWEEP_BACKWARD, s.iterations));
        }
        return 0;
    }
    case AMG_SM_JACOBI_NE: {
        // relaxation.py:744-818: delta = (b - A x) .* Dinv; x += A^H (omega delta), the contributions to
        // an entry added in (row, position) order = down the column of A
        const DevCsr &Acol = s.aux[0];
        for (int it = 0; it < s.iterations; ++it) {
            CHK(spmv(L.A, SM_RESIDUAL, x, b, nullptr, L.r, nullptr, 0.0, st));
            CHK(launch_mul_elem(L.r, s.Dinv, n, st));
            CHK(launch_jacobi_ne_gather(Acol.Ap, Acol.Aj, Acol.Ax, L.r, s.omega, L.h, n, st));
            CHK(launch_axpy_inplace(x, L.h, n, st));
        }
        return 0;
    }
    case AMG_SM_SCHWARZ: {
        // relaxation.py:254-277: subdomains by dependency levels; descending levels = backward sweep
        const std::vector<int> &lp = s.sw_level_ptr;
        const int nl = (int)lp.size() - 1;
        auto sweep_once = [&](bool reverse) -> int {
            for (int q = 0; q < nl; ++q) {
                const int l = reverse ? nl - 1 - q : q;
                CHK(launch_schwarz_level(L.A.Ap, L.A.Aj, L.A.Ax, x, b, s.sw_Tx, s.sw_Tp, s.sw_Sj, s.sw_Sp,
                                         s.sw_scratch, s.sw_order + lp[l], lp[l + 1] - lp[l], st));
            }
            return 0;
        };
        for (int it = 0; it < s.iterations; ++it) {
            if (s.sweep == AMG_SWEEP_FORWARD || s.sweep == AMG_SWEEP_SYMMETRIC) CHK(sweep_once(false));
            if (s.sweep == AMG_SWEEP_BACKWARD || s.sweep == AMG_SWEEP_SYMMETRIC) CHK(sweep_once(true));
        }
        return 0;
    }
    case AMG_SM_BLOCK_JACOBI: {
        // relaxation.py:430-506
        const DevBsr &Ab = s.Ablk_owned ? s.Ablk : L.Ab;
        for (int it = 0; it < s.iterations; ++it) {
            CHK(bsr_stream_all(Ab, BM_BLOCK_JACOBI, s.Dinv, x, xalt, b, s.omega, st));
            std::swap(x, xalt);
        }
        return 0;
    }
    case AMG_SM_BLOCK_GAUSS_SEIDEL: {
        // relaxation.py:509-590
        std::vector<unsigned char> seq;
        for (int it = 0; it < s.iterations; ++it) {
            if (s.sweep == AMG_SWEEP_FORWARD || s.sweep == AMG_SWEEP_SYMMETRIC) seq.push_back(0);
            if (s.sweep == AMG_SWEEP_BACKWARD || s.sweep == AMG_SWEEP_SYMMETRIC) seq.push_back(1);
        }
        return block_gs_sweeps(*s.sched, s.Dinv, x, b, seq.data(), (int)seq.size(), st, flow_allowed(h));
    }
    }
    set_error("unknown smoother kind");
    return AMG_EINVAL;
}

// coarse_grid_solver (multilevel.py:554-720) on the last level
static int coarse_solve(amg_hier *h, const double *b, double *&x, double *&xalt)
{
    Level &L = h->lv[h->nlevels - 1];
    const int n = L.A.nrows;
    if (L.A.nnz == 0 || h->coarse_kind == 0) {
        // generic_solver: A.nnz == 0 -> zeros (multilevel.py:699-701); solver None -> 0*b
        AMG_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, h->stream));
        return 0;
    }
    if (h->coarse_kind == 1 && h->comm && h->coarse_gather_channel >= 0) {
        CHK(comm_exchange_begin(h->comm, h->coarse_gather_channel, b, h->coarse_gather_idx, h->coarse_full_b, h->stream));
        CHK(comm_exchange_end(h->comm, h->coarse_gather_channel, h->coarse_full_b, h->stream));
        CHK(launch_dense_apply(h->coarse_Mt, h->coarse_full_b, h->coarse_full_x, h->coarse_n, h->stream));
        AMG_HIP(hipMemcpyAsync(x, h->coarse_full_x + h->coarse_lo, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
        return 0;
    }
    if (h->coarse_kind == 1) return launch_dense_apply(h->coarse_Mt, b, x, n, h->stream);
    if (h->coarse_kind == 3) {
        // host coarse solver (multilevel.py:642-692 Krylov names / callables): the coarsest right-hand side (a few
        // hundred entries) goes to the host, the correction comes back
        h->coarse_hb.resize((size_t)n); h->coarse_hx.assign((size_t)n, 0.0);
        AMG_HIP(hipMemcpyAsync(h->coarse_hb.data(), b, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        AMG_HIP(hipStreamSynchronize(h->stream));
        if (h->coarse_cb(h->coarse_cb_user, n, h->coarse_hb.data(), h->coarse_hx.data()) != 0) { set_error("coarse solver callback failed"); return AMG_ESTATE; }
        AMG_HIP(hipMemcpyAsync(x, h->coarse_hx.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->stream));
        AMG_HIP(hipStreamSynchronize(h->stream));
        return 0;
    }
    if (h->comm && L.part.channel >= 0) { set_error("relaxation as coarse solver has no row-partitioned form"); return AMG_ENOTIMPL; }
    AMG_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, h->stream));   // multilevel.py:675
    return relax(h, L, h->coarse_sm, x, xalt, b, true);
}

// multilevel_solver.__solve (multilevel.py:473-548)
static int cycle(amg_hier *h, int lvl, double *&x, double *&xalt, const double *b, int cyc, bool x_zero,
                 bool r_ready = false)
{
    Level &L = h->lv[lvl];
    Level &Lc = h->lv[lvl + 1];
    hipStream_t st = h->stream;
    const int nc = Lc.A.nrows;

    CHK(relax(h, L, L.sm[AMG_PRE], x, xalt, b, x_zero, r_ready));                          // :494
    CHK(level_apply(h, L, SM_RESIDUAL, x, 0.0, b, nullptr, L.r, 0.0));                     // :496
    CHK(exchange(h, L, L.r));                                                              // R gathers fine residuals
    if (h->comm && L.part.gather_channel >= 0) {
        // entering the replicated levels: every rank restricts its slice of the coarse rows, all ranks gather
        CHK(spmv(L.Rm, SM_MATVEC, L.r, nullptr, nullptr, L.part.rslice, nullptr, 0.0, st));
        CHK(comm_exchange_begin(h->comm, L.part.gather_channel, L.part.rslice, L.part.gather_idx, Lc.b, st));
        CHK(comm_exchange_end(h->comm, L.part.gather_channel, Lc.b, st));
    } else {
        CHK(spmv(L.Rm, SM_MATVEC, L.r, nullptr, nullptr, Lc.b, nullptr, 0.0, st));         // :498
    }
    AMG_HIP(hipMemsetAsync(Lc.x, 0, sizeof(double) * (size_t)nc, st));                     // :499

    if (lvl == h->nlevels - 2) {
        CHK(coarse_solve(h, Lc.b, Lc.x, Lc.xalt));                                         // :501-502
    } else if (cyc == AMG_CYCLE_V) {
        CHK(cycle(h, lvl + 1, Lc.x, Lc.xalt, Lc.b, AMG_CYCLE_V, true));                    // :504-505
    } else if (cyc == AMG_CYCLE_W) {
        CHK(cycle(h, lvl + 1, Lc.x, Lc.xalt, Lc.b, cyc, true));                            // :506-508
        CHK(cycle(h, lvl + 1, Lc.x, Lc.xalt, Lc.b, cyc, false));
    } else if (cyc == AMG_CYCLE_F) {
        CHK(cycle(h, lvl + 1, Lc.x, Lc.xalt, Lc.b, cyc, true));                            // :509-511
        CHK(cycle(h, lvl + 1, Lc.x, Lc.xalt, Lc.b, AMG_CYCLE_V, false));
    } else if (h->comm) {
        set_error("AMLI cycles have no row-partitioned form");
        return AMG_ENOTIMPL;
    } else {
        // AMLI (multilevel.py:512-540): two coarse cycles from an all-ones guess, A-orthogonalised and combined with
        // optimal step lengths.  The inner products stay in device memory and the updates read them from there
        // (launch_axmy_ratio): no host round trip inside the cycle, so AMLI iterations are graph-replayed like the others.
        const int nAMLI = 2;
        const size_t bytes = sizeof(double) * (size_t)nc;
        for (int q = 0; q < 4; ++q)
            if (!Lc.amli[q]) CHK(dev_alloc(&Lc.amli[q], nc, &h->dev_bytes));
        double *p[2] = {Lc.amli[0], Lc.amli[1]};
        double *Apk = Lc.amli[2], *Apj = Lc.amli[3];
        double *num = h->norm_scratch + 1027, *den = h->norm_scratch + 1029;
        for (int k = 0; k < nAMLI; ++k) {
            CHK(launch_fill(p[k], 1.0, nc, st));                           // p[k,:] = 1
            double *pk = p[k], *alt = Lc.xalt;
            CHK(cycle(h, lvl + 1, pk, alt, Lc.b, cyc, false));
            if (pk != p[k]) {                                              // Jacobi swapped the buffers
                AMG_HIP(hipMemcpyAsync(p[k], pk, bytes, hipMemcpyDeviceToDevice, st));
            }
            for (int j = 0; j < k; ++j) {
                CHK(spmv(Lc.A, SM_MATVEC, p[k], nullptr, nullptr, Apk, nullptr, 0.0, st));
                CHK(spmv(Lc.A, SM_MATVEC, p[j], nullptr, nullptr, Apj, nullptr, 0.0, st));
                CHK(launch_dot(p[j], Apk, nc, h->norm_scratch, num, st));
                CHK(launch_dot(p[j], Apj, nc, h->norm_scratch, den, st));
                CHK(launch_axmy_ratio(p[k], p[j], num, den, 1.0, nc, st));  // p[k] -= beta*p[j], beta = num / den
            }
            CHK(spmv(Lc.A, SM_MATVEC, p[k], nullptr, nullptr, Apk, nullptr, 0.0, st));
            CHK(launch_dot(p[k], Lc.b, nc, h->norm_scratch, num, st));
            CHK(launch_dot(p[k], Apk, nc, h->norm_scratch, den, st));
            CHK(launch_axmy_ratio(Lc.x, p[k], num, den, -1.0, nc, st));     // coarse_x += alpha*p[k], alpha = num / den
            CHK(launch_axmy_ratio(Lc.b, Apk, num, den, 1.0, nc, st));       // coarse_b -= alpha*Ap
        }
    }

    CHK(exchange(h, Lc, Lc.x));                                                            // P gathers coarse corrections
    CHK(spmv(L.P, SM_MATVEC_ACC, Lc.x, nullptr, nullptr, x, nullptr, 0.0, st));            // :544
    CHK(relax(h, L, L.sm[AMG_POST], x, xalt, b, false));                                   // :545
    return 0;
}

static int one_iteration(amg_hier *h, int cyc, bool x_zero, bool r_ready = false)
{
    Level &L0 = h->lv[0];
    if (h->nlevels == 1) return coarse_solve(h, L0.b, L0.x, L0.xalt);                      // :455-457
    return cycle(h, 0, L0.x, L0.xalt, L0.b, cyc, x_zero, r_ready);                         // :459
}

static int residual_norm_to(amg_hier *h, double *slot)
{
    // util/linalg.py:109-112: ||b - A x||.  Every workgroup of the operator application reduces its
    // rows' squares, a second kernel adds the partials in order.  The residual vector itself is
    // stored only when the next cycle's pre-smoother starts by forming exactly this b - A x (the
    // polynomial smoother, relaxation.py:655): it then reads it from L0.r instead of applying A
    // a second time to the same x -- the same kernel on the same operands, hence the same bits.
    Level &L0 = h->lv[0];
    StreamArgs a = base_args(L0.A);
    a.xg = L0.x; a.b = L0.b; a.out2 = h->sumsq_partials;
    const bool keep = h->keep_residual && h->nlevels > 1 && L0.sm[AMG_PRE].kind == AMG_SM_POLYNOMIAL &&
                      L0.sm[AMG_PRE].iterations >= 1;
    if (h->comm) {
        // partitioned: r = b - A x over the owned rows (halo of x refreshed, interior rows overlapped), then the
        // all-reduced norm; r stays in L0.r for a polynomial pre-smoother as on one GPU
        CHK(level_apply(h, L0, SM_RESIDUAL, L0.x, 0.0, L0.b, nullptr, L0.r, 0.0));
        h->r_kept = keep;
        return global_norm(h, L0.r, L0.A.nrows, slot);
    }
    a.out = keep ? L0.r : nullptr;
    h->r_kept = keep;
    const bool stencil = L0.A.st_vals && stencil_enabled();
    if (applies_from_blocks(L0.A, SM_RESIDUAL_SUMSQ, a)) {
        const BsrStreamArgs q = block_spmv_args(*L0.A.blk, SM_RESIDUAL_SUMSQ, a);
        const int nbq = bsr_stream_blocks(q, L0.A.blk->nblocks);
        if (nbq > h->sumsq_cap) { set_error("sumsq partial buffer too small"); return AMG_ESTATE; }
        CHK(launch_bsr_stream(BM_SPMV, q, L0.A.blk->nblocks, h->stream));
        return launch_sum_sqrt(h->sumsq_partials, nbq, h->sumsq_partials + h->sumsq_cap, slot, h->stream);
    }
    const int nb = stencil ? stencil_blocks(a, L0.A) : stream_blocks(a);
    if (nb > h->sumsq_cap) { set_error("sumsq partial buffer too small"); return AMG_ESTATE; }
    if (stencil) CHK(launch_stencil(SM_RESIDUAL_SUMSQ, a, L0.A, h->stream));
    else if (L0.A.pat) CHK(launch_pattern(SM_RESIDUAL_SUMSQ, a, L0.A, h->stream));
    else CHK(launch_stream(SM_RESIDUAL_SUMSQ, a, h->stream));
    return launch_sum_sqrt(h->sumsq_partials, nb, h->sumsq_partials + h->sumsq_cap, slot, h->stream);
}

// One solve() iteration = cycle + residual norm into `slot`, replayed from a hipGraph once the
// same buffer state has been seen before (the first occurrence runs eagerly, the second is
// captured, later ones are replays).  Everything inside is kernel / memset / D2D-copy nodes on
// h->stream; no host synchronisation.
static std::vector<double *> buffer_state(amg_hier *h)
{
    std::vector<double *> s;
    for (auto &L : h->lv) { s.push_back(L.x); s.push_back(L.xalt); }
    return s;
}

static int iteration_with_norm(amg_hier *h, int cyc, bool x_zero, double *slot)
{
    const bool r_ready = h->r_kept;
    h->r_kept = false;
    CHK(one_iteration(h, cyc, x_zero, r_ready));
    return residual_norm_to(h, slot);
}

static void drop_graphs(amg_hier *h);

static int graph_iteration(amg_hier *h, int cyc, bool x_zero, double *dst)
{
    hipStream_t st = h->stream;
    double *slot = h->norm_scratch + 1028;
    if (!h->use_graphs || h->has_callbacks) return iteration_with_norm(h, cyc, x_zero, dst);
    if (h->graph_epoch != config_epoch()) {      // a launch knob changed: the captured launches are stale
        drop_graphs(h);
        h->graph_epoch = config_epoch();
    }
    std::vector<double *> state = buffer_state(h);
    GraphEntry *ge = nullptr;
    for (auto &g : h->graphs)
        if (g.cyc == cyc && g.x_zero == x_zero && g.kept_in == h->r_kept && g.state_in == state) { ge = &g; break; }
    if (!ge) {
        h->graphs.emplace_back();
        ge = &h->graphs.back();
        ge->cyc = cyc; ge->x_zero = x_zero; ge->kept_in = h->r_kept; ge->state_in = state;
    }
    if (!ge->exec && ge->seen < 1) {           // first encounter: eager
        ge->seen++;
        return iteration_with_norm(h, cyc, x_zero, dst);
    }
    if (!ge->exec) {                            // second encounter: capture
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { h->use_graphs = 0; return iteration_with_norm(h, cyc, x_zero, dst); }
        int rc = iteration_with_norm(h, cyc, x_zero, slot);
        hipGraph_t graph = nullptr;
        e = hipStreamEndCapture(st, &graph);
        if (rc != 0 || e != hipSuccess || !graph) {
            // capture failed: the pointer swaps already happened but nothing ran -> restore and go eager
            if (graph) hipGraphDestroy(graph);
            for (size_t l = 0; l < h->lv.size(); ++l) { h->lv[l].x = state[2 * l]; h->lv[l].xalt = state[2 * l + 1]; }
            h->use_graphs = 0;
            (void)hipGetLastError();
            return iteration_with_norm(h, cyc, x_zero, dst);
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            hipGraphDestroy(graph);
            for (size_t l = 0; l < h->lv.size(); ++l) { h->lv[l].x = state[2 * l]; h->lv[l].xalt = state[2 * l + 1]; }
            h->use_graphs = 0;
            (void)hipGetLastError();
            return iteration_with_norm(h, cyc, x_zero, dst);
        }
        ge->graph = graph; ge->exec = exec;
        ge->state_out = buffer_state(h);
        ge->kept_out = h->r_kept;
    } else {
        for (size_t l = 0; l < h->lv.size(); ++l) { h->lv[l].x = ge->state_out[2 * l]; h->lv[l].xalt = ge->state_out[2 * l + 1]; }
        h->r_kept = ge->kept_out;
    }
    AMG_HIP(hipGraphLaunch(ge->exec, st));
    AMG_HIP(hipMemcpyAsync(dst, slot, sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
}

static void drop_graphs(amg_hier *h)
{
    for (auto &g : h->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
}

// ------------------------------------------------------------------ C API
extern "C" {

const char *amg_last_error(void) { return amg::last_error().c_str(); }

int amg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *amg_device_name(int device)
{
    static thread_local std::string name;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return "";
    name = std::string(p.gcnArchName) + " " + p.name;
    return name.c_str();
}

amg_hier *amg_hier_create(int nlevels, int device)
{
    if (nlevels < 1) { set_error("nlevels < 1"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (amgcore_hip has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("bad device index"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    amg_hier *h = new amg_hier();
    h->device = device;
    h->nlevels = nlevels;
    h->lv.resize((size_t)nlevels);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        set_error("stream/event creation failed");
        delete h;
        return nullptr;
    }
    return h;
}

static void free_smoother(Smoother &s)
{
    if (s.sw_Sj) hipFree(s.sw_Sj);
    if (s.sw_Sp) hipFree(s.sw_Sp);
    if (s.sw_Tp) hipFree(s.sw_Tp);
    if (s.sw_order) hipFree(s.sw_order);
    if (s.sw_Tx) hipFree(s.sw_Tx);
    if (s.sw_scratch) hipFree(s.sw_scratch);
    s.sw_Sj = s.sw_Sp = s.sw_Tp = s.sw_order = nullptr;
    s.sw_Tx = s.sw_scratch = nullptr;
    free_csr(s.aux[0]);
    free_csr(s.aux[1]);
    if (s.Dinv) hipFree(s.Dinv);
    s.Dinv = nullptr;
    if (s.Ablk_owned) free_bsr(s.Ablk);
    if (s.sched && s.sched.use_count() == 1) s.sched->release();
    s.sched.reset();
}

void amg_hier_destroy(amg_hier *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    drop_graphs(h);
    for (auto &L : h->lv) {
        free_smoother(L.sm[0]);
        free_smoother(L.sm[1]);
        if (L.sched_csr && L.sched_csr.use_count() == 1) L.sched_csr->release();
        if (L.sched_blk && L.sched_blk.use_count() == 1) L.sched_blk->release();
        if (L.sched_bgs && L.sched_bgs.use_count() == 1) L.sched_bgs->release();
        free_csr(L.A); free_csr(L.P); free_csr(L.Rm); free_bsr(L.Ab);
        if (L.part.send_idx) hipFree(L.part.send_idx);
        if (L.part.gather_idx) hipFree(L.part.gather_idx);
        if (L.part.rslice) hipFree(L.part.rslice);
        for (double *p : {L.x, L.xalt, L.b, L.r, L.h, L.h2, L.amli[0], L.amli[1], L.amli[2], L.amli[3]}) if (p) hipFree(p);
    }
    free_smoother(h->coarse_sm);
    if (h->coarse_Mt) hipFree(h->coarse_Mt);
    if (h->coarse_gather_idx) hipFree(h->coarse_gather_idx);
    if (h->coarse_full_b) hipFree(h->coarse_full_b);
    if (h->coarse_full_x) hipFree(h->coarse_full_x);
    if (h->arn_V) hipFree(h->arn_V);
    if (h->arn_dinv) hipFree(h->arn_dinv);
    if (h->arn_coef) hipFree(h->arn_coef);
    if (h->norm_scratch) hipFree(h->norm_scratch);
    if (h->sumsq_partials) hipFree(h->sumsq_partials);
    for (double *q : h->pcg) if (q) hipFree(q);
    if (h->res_dev) hipFree(h->res_dev);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

#define ENTER(h)                                                        \
    if (!(h)) { set_error("null hierarchy"); return AMG_EINVAL; }       \
    AMG_HIP(hipSetDevice((h)->device))

/* Value index of an operator in stencil form (amg_dev.hpp, DevCsr::st_codes): scans the padded values on the device and,
 * when there are at most 255 distinct ones, stores one-byte codes into a dictionary.  Returns the number of distinct values,
 * 0 when not applicable (no stencil form, too many values), < 0 on error. */
static int value_index_build(amg_hier *h, DevCsr &M)
{
    if (!M.st_vals) return 0;
    if (M.st_codes) { M.st_vi_on = true; return M.st_ndict; }
    const long count = (long)((M.nrows + 255) / 256) * M.st_nu * 256;
    unsigned long long *table = nullptr;
    int *ovf = nullptr;
    if (dev_alloc(&table, 1024, nullptr) != 0 || dev_alloc(&ovf, 1, nullptr) != 0) return AMG_ENOMEM;
    AMG_HIP(hipMemset(table, 0xFF, sizeof(unsigned long long) * 1024));
    AMG_HIP(hipDeviceSynchronize());
    int rc = launch_value_scan(M.st_vals, count, table, ovf, h->stream);
    std::vector<unsigned long long> tb(1024);
    int overflow = 0;
    AMG_HIP(hipStreamSynchronize(h->stream));
    AMG_HIP(hipMemcpy(tb.data(), table, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost));
    AMG_HIP(hipMemcpy(&overflow, ovf, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(table); hipFree(ovf);
    if (rc != 0) return AMG_ESTATE;
    std::vector<long long> bits;
    for (unsigned long long v : tb) if (v != ~0ULL) bits.push_back((long long)v);
    if (overflow || bits.empty() || bits.size() > 255) return 0;          // not a few-valued operator
    std::sort(bits.begin(), bits.end());
    std::vector<double> dict(256, 0.0);
    for (size_t k = 0; k < bits.size(); ++k) std::memcpy(&dict[k], &bits[k], sizeof(double));
    if (dev_alloc(&M.st_dict, 256, &h->dev_bytes) != 0) return AMG_ENOMEM;
    AMG_HIP(hipMemcpy(M.st_dict, dict.data(), sizeof(double) * 256, hipMemcpyHostToDevice));
    const long code_bytes = (long)((M.nrows + 255) / 256) * ((M.st_nu + 7) / 8) * 256 * 8;
    if (dev_alloc(&M.st_codes, code_bytes, &h->dev_bytes) != 0) return AMG_ENOMEM;
    M.st_ndict = (int)bits.size();
    AMG_HIP(hipMemsetAsync(M.st_codes, 0xFF, (size_t)code_bytes, h->stream));      // slots the stencil does not have: absent (255)
    rc = launch_value_encode(M.st_vals, count, M.st_dict, M.st_ndict, M.st_codes, M.st_nu, h->stream,
                             M.st_nu <= 7 ? static_cast<const unsigned char *>(M.st_mask) : nullptr, M.nrows);
    AMG_HIP(hipStreamSynchronize(h->stream));
    if (rc != 0) return AMG_ESTATE;
    M.st_vi_on = true;
    return M.st_ndict;
}

// r3: applied when an operator is set (amg_hier_set_matrix) unless AMG_VALUE_INDEX=0 / amg_set_value_index(0): an operator
// whose values do not compress is left as it is (one scan of its values, ~1 ms per 5 GB)
static int g_value_index_auto = std::getenv("AMG_VALUE_INDEX") ? std::atoi(std::getenv("AMG_VALUE_INDEX")) : 1;
void amg_set_value_index(int on) { g_value_index_auto = on ? 1 : 0; }
int amg_value_index_enabled(void) { return g_value_index_auto; }

int amg_hier_set_matrix(amg_hier *h, int lvl, int which, int fmt, int nrows, int ncols, int R, int C,
                        const int *Ap, const int *Aj, const double *Ax, int on_device)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2) { set_error("bad level/which"); return AMG_EINVAL; }
    if (fmt == AMG_FMT_CSR) { R = 1; C = 1; }
    if (R < 1 || C < 1 || nrows % R || ncols % C) { set_error("bad block size"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    DevCsr &M = (which == AMG_MAT_A) ? L.A : (which == AMG_MAT_P ? L.P : L.Rm);
    free_csr(M);
    if (on_device) {
        // device-resident source (e.g. an operator assembled on the GPU): copied device-to-device
        if (fmt != AMG_FMT_CSR && !(R == 1 && C == 1)) { set_error("device sources: CSR / BSR(1,1) only"); return AMG_ENOTIMPL; }
        int last = 0;
        AMG_HIP(hipMemcpy(&last, Ap + nrows, sizeof(int), hipMemcpyDeviceToHost));
        if (last < 0) { set_error("bad device row pointer"); return AMG_EINVAL; }
        M.nrows = nrows; M.ncols = ncols; M.nnz = last;
        CHK(dev_alloc(&M.Ap, nrows + 1, &h->dev_bytes));
        CHK(dev_alloc(&M.Aj, last, &h->dev_bytes));
        CHK(dev_alloc(&M.Ax, last, &h->dev_bytes));
        AMG_HIP(hipMemcpy(M.Ap, Ap, sizeof(int) * (size_t)(nrows + 1), hipMemcpyDeviceToDevice));
        if (last) {
            AMG_HIP(hipMemcpy(M.Aj, Aj, sizeof(int) * (size_t)last, hipMemcpyDeviceToDevice));
            AMG_HIP(hipMemcpy(M.Ax, Ax, sizeof(double) * (size_t)last, hipMemcpyDeviceToDevice));
        }
        if (which == AMG_MAT_A && nrows >= 1024) {
            // the structure analysis runs on the host: fetch the index arrays (not the values)
            std::vector<int> hp((size_t)nrows + 1), hj((size_t)last);
            AMG_HIP(hipMemcpy(hp.data(), Ap, sizeof(int) * hp.size(), hipMemcpyDeviceToHost));
            if (last) AMG_HIP(hipMemcpy(hj.data(), Aj, sizeof(int) * hj.size(), hipMemcpyDeviceToHost));
            CHK(try_patterns(M, hp.data(), hj.data(), &h->dev_bytes));
            CHK(build_index16(M, hp.data(), &h->dev_bytes));
            if (M.st_vals && g_value_index_auto && value_index_build(h, M) < 0) return AMG_ESTATE;
        }
    } else if (fmt == AMG_FMT_CSR || (R == 1 && C == 1)) {
        CHK(upload_csr(M, nrows, ncols, Ap, Aj, Ax, &h->dev_bytes));
        if (which == AMG_MAT_A) CHK(try_patterns(M, Ap, Aj, &h->dev_bytes));
        CHK(build_index16(M, Ap, &h->dev_bytes));
        if (which == AMG_MAT_A && M.st_vals && g_value_index_auto && value_index_build(h, M) < 0) return AMG_ESTATE;
    } else if (which == AMG_MAT_A && R == C && bsr_spmv_enabled(R)) {
        // a level operator with square blocks that every application streams from the blocks themselves (8 B per
        // entry + 4 B per block): no scalar expansion -- at 3x3 blocks and 5*10^7 rows it would hold more than
        // 2^31 entries and cost 12 B per entry.  The block copy is uploaded below.
        M.nrows = nrows; M.ncols = ncols;
        M.nnz = (long)Ap[nrows / R] * R * C;
    } else {
        if ((double)Ap[nrows / R] * R * C > 2147483647.0) {
            set_error("the scalar expansion of this BSR operator exceeds int32 entries");
            return AMG_EINVAL;
        }
        std::vector<int> cp, cj;
        std::vector<double> cx;
        expand_bsr(nrows / R, R, C, Ap, Aj, Ax, cp, cj, cx);
        CHK(upload_csr(M, nrows, ncols, cp.data(), cj.data(), cx.data(), &h->dev_bytes));
    }
    // operators without grid structure: the sliced form (sell.hip) for whole-operator applications
    // (a partitioned level's A gets it in amg_hier_set_partition, for the interior rows that run beside the exchange)
    if (M.Ap && !M.pat && !M.st_vals && !(h->comm && which == AMG_MAT_A)) CHK(build_sell(M, &h->dev_bytes));
    if (which == AMG_MAT_A) {
        if (nrows != ncols && !(h->comm && ncols > nrows)) { set_error("A must be square (row-partitioned: owned rows x [owned | halo] columns)"); return AMG_EINVAL; }
        L.fmt = fmt; L.R = R; L.C = C; L.hasA = true;
        free_bsr(L.Ab);
        if (fmt == AMG_FMT_BSR && R == C && R > 1) {
            CHK(upload_bsr(L.Ab, nrows / R, R, Ap, Aj, Ax, &h->dev_bytes));
            if (!h->comm) CHK(build_bsell(L.Ab, &h->dev_bytes));        // whole passes run from the sliced block form (sell.hip)
            L.A.blk = &L.Ab;           // applications of A stream the blocks (levels live in a vector sized once)
        }
    } else if (which == AMG_MAT_P) {
        L.hasP = true;
    } else {
        L.hasR = true;
    }
    h->finalized = false;
    return 0;
}

static int fill_smoother(amg_hier *h, Smoother &s, const amg_smoother_desc *d, int n)
{
    free_smoother(s);
    s = Smoother();
    if (!d) return 0;
    s.kind = d->kind;
    s.iterations = d->iterations > 0 ? d->iterations : 1;
    s.sweep = d->sweep;
    s.omega = d->omega;
    if (d->kind == AMG_SM_POLYNOMIAL) {
        if (d->ncoef < 1 || !d->coef) { set_error("polynomial smoother needs coefficients"); return AMG_EINVAL; }
        s.coef.assign(d->coef, d->coef + d->ncoef);
    }
    if (d->kind == AMG_SM_BLOCK_JACOBI || d->kind == AMG_SM_BLOCK_GAUSS_SEIDEL) {
        if (d->blocksize < 1 || n % d->blocksize || !d->Dinv) { set_error("block smoother needs blocksize and Dinv"); return AMG_EINVAL; }
        s.bs = d->blocksize;
        long cnt = (long)n * s.bs;
        CHK(dev_alloc(&s.Dinv, cnt, &h->dev_bytes));
        AMG_HIP(hipMemcpy(s.Dinv, d->Dinv, sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice));
    }
    if (d->kind == AMG_SM_GAUSS_SEIDEL_INDEXED) {
        if (d->nindices < 0 || (d->nindices && !d->indices)) { set_error("indexed GS needs indices"); return AMG_EINVAL; }
        s.indices.assign(d->indices, d->indices + d->nindices);
    }
    if (d->kind == AMG_SM_GAUSS_SEIDEL_NE || d->kind == AMG_SM_GAUSS_SEIDEL_NR || d->kind == AMG_SM_JACOBI_NE) {
        if (!d->Dinv) { set_error("normal-equation smoother needs the inverse diagonal of A A^H / A^H A"); return AMG_EINVAL; }
        CHK(dev_alloc(&s.Dinv, n, &h->dev_bytes));
        AMG_HIP(hipMemcpy(s.Dinv, d->Dinv, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        CHK(dev_alloc(&s.sw_order, n, &h->dev_bytes));
    }
    if (d->kind == AMG_SM_SCHWARZ) {
        const int nsd = d->nsdomains;
        if (nsd < 0 || !d->Sp || !d->Tp || (nsd && (!d->Sj || !d->Tx))) { set_error("schwarz smoother needs subdomains and inverse blocks"); return AMG_EINVAL; }
        for (int k = 0; k < nsd; ++k) {
            const long m = (long)d->Sp[k + 1] - d->Sp[k];
            if (m < 0 || m * m != (long)d->Tp[k + 1] - d->Tp[k]) { set_error("schwarz: inverse block size does not match its subdomain"); return AMG_EINVAL; }
        }
        s.nsd = nsd;
        s.hSp.assign(d->Sp, d->Sp + nsd + 1);
        s.hSj.assign(d->Sj, d->Sj + (nsd ? d->Sp[nsd] : 0));
        for (int v : s.hSj) if (v < 0 || v >= n) { set_error("schwarz: subdomain index out of range"); return AMG_EINVAL; }
        CHK(dev_alloc(&s.sw_Sp, nsd + 1, &h->dev_bytes));
        CHK(dev_alloc(&s.sw_Tp, nsd + 1, &h->dev_bytes));
        CHK(dev_alloc(&s.sw_Sj, (long)s.hSj.size(), &h->dev_bytes));
        CHK(dev_alloc(&s.sw_Tx, nsd ? d->Tp[nsd] : 0, &h->dev_bytes));
        CHK(dev_alloc(&s.sw_scratch, (long)s.hSj.size(), &h->dev_bytes));
        CHK(dev_alloc(&s.sw_order, nsd, &h->dev_bytes));
        AMG_HIP(hipMemcpy(s.sw_Sp, d->Sp, sizeof(int) * (size_t)(nsd + 1), hipMemcpyHostToDevice));
        AMG_HIP(hipMemcpy(s.sw_Tp, d->Tp, sizeof(int) * (size_t)(nsd + 1), hipMemcpyHostToDevice));
        if (nsd) {
            AMG_HIP(hipMemcpy(s.sw_Sj, d->Sj, sizeof(int) * s.hSj.size(), hipMemcpyHostToDevice));
            AMG_HIP(hipMemcpy(s.sw_Tx, d->Tx, sizeof(double) * (size_t)d->Tp[nsd], hipMemcpyHostToDevice));
        }
    }
    return 0;
}

int amg_hier_set_smoother(amg_hier *h, int lvl, int which, const amg_smoother_desc *d)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 1) { set_error("bad level/which"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    if (!L.hasA) { set_error("set A before its smoothers"); return AMG_ESTATE; }
    h->finalized = false;
    return fill_smoother(h, L.sm[which], d, L.A.nrows);
}

/* re-blocked copy of A for a block smoother: relaxation.py:471,563 A.tobsr(blocksize) */
int amg_hier_set_block_matrix(amg_hier *h, int lvl, int which, int nbrows, int bs, const int *Ap,
                              const int *Aj, const double *Ax)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2) { set_error("bad level/which"); return AMG_EINVAL; }
    Smoother &s = (which == 2) ? h->coarse_sm : h->lv[lvl].sm[which];
    if (s.Ablk_owned) free_bsr(s.Ablk);
    CHK(upload_bsr(s.Ablk, nbrows, bs, Ap, Aj, Ax, &h->dev_bytes));
    s.Ablk_owned = true;
    h->finalized = false;
    return 0;
}

int amg_hier_set_aux_matrix(amg_hier *h, int lvl, int which, int slot, int nmajor, int nminor, const int *Ap,
                            const int *Aj, const double *Ax)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2 || slot < 0 || slot > 1) { set_error("bad level/which/slot"); return AMG_EINVAL; }
    if (nmajor < 0 || nminor < 0 || !Ap) { set_error("bad matrix"); return AMG_EINVAL; }
    Smoother &s = (which == 2) ? h->coarse_sm : h->lv[lvl].sm[which];
    free_csr(s.aux[slot]);
    CHK(upload_csr(s.aux[slot], nmajor, nminor, Ap, Aj, Ax, &h->dev_bytes));
    h->finalized = false;
    return 0;
}

int amg_hier_set_coarse_dense(amg_hier *h, const double *M, int n)
{
    ENTER(h);
    if (n < 0 || (n && !M)) { set_error("bad coarse matrix"); return AMG_EINVAL; }
    if (h->coarse_Mt) hipFree(h->coarse_Mt);
    h->coarse_Mt = nullptr;
    std::vector<double> Mt((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) Mt[(size_t)k * n + i] = M[(size_t)i * n + k];
    CHK(dev_alloc(&h->coarse_Mt, (long)n * n, &h->dev_bytes));
    AMG_HIP(hipMemcpy(h->coarse_Mt, Mt.data(), sizeof(double) * Mt.size(), hipMemcpyHostToDevice));
    h->coarse_n = n;
    h->coarse_kind = 1;
    return 0;
}

int amg_hier_set_coarse_smoother(amg_hier *h, const amg_smoother_desc *d)
{
    ENTER(h);
    Level &L = h->lv[h->nlevels - 1];
    if (!L.hasA) { set_error("set the coarsest A first"); return AMG_ESTATE; }
    CHK(fill_smoother(h, h->coarse_sm, d, L.A.nrows));
    h->coarse_kind = d ? 2 : 0;
    h->finalized = false;
    return 0;
}

static int need_schedule(amg_hier *h, Level &L, Smoother &s)
{
    const int n = L.A.nrows;
    if (!L.A.Ap && L.hasA && !(L.fmt == AMG_FMT_BSR && L.R > 1) &&
        !(s.kind == AMG_SM_NONE || s.kind == AMG_SM_POLYNOMIAL || s.kind == AMG_SM_JACOBI || s.kind == AMG_SM_CALLBACK)) {
        set_error("this level's CSR arrays were released (amg_hier_release_sources): rebuild the hierarchy to attach this smoother");
        return AMG_ESTATE;
    }
    const bool bsr_pt = (L.fmt == AMG_FMT_BSR && L.R > 1);
    if (s.kind == AMG_SM_GAUSS_SEIDEL || s.kind == AMG_SM_SOR) {
        if (bsr_pt) {
            if (!L.sched_blk) {
                std::vector<int> bp((size_t)L.Ab.nbrows + 1), bj((size_t)L.Ab.nblocks);
                std::vector<double> bx((size_t)L.Ab.nblocks * L.Ab.bs * L.Ab.bs);
                AMG_HIP(hipMemcpy(bp.data(), L.Ab.Ap, sizeof(int) * bp.size(), hipMemcpyDeviceToHost));
                if (!bj.empty()) {
                    AMG_HIP(hipMemcpy(bj.data(), L.Ab.Aj, sizeof(int) * bj.size(), hipMemcpyDeviceToHost));
                    AMG_HIP(hipMemcpy(bx.data(), L.Ab.Ax, sizeof(double) * bx.size(), hipMemcpyDeviceToHost));
                }
                L.sched_blk = std::make_shared<Schedule>();
                CHK(build_block_schedule(bp.data(), bj.data(), L.Ab.nbrows, nullptr, L.Ab.nbrows, *L.sched_blk, h->stream,
                                         bx.data(), L.Ab.bs));
                h->dev_bytes += L.sched_blk->level_copy_bytes;
            }
            s.sched = L.sched_blk;
        } else {
            if (!L.sched_csr) {
                std::vector<int> ap((size_t)n + 1), aj((size_t)L.A.nnz);
                std::vector<double> ax((size_t)L.A.nnz);
                AMG_HIP(hipMemcpy(ap.data(), L.A.Ap, sizeof(int) * ap.size(), hipMemcpyDeviceToHost));
                if (L.A.nnz) {
                    AMG_HIP(hipMemcpy(aj.data(), L.A.Aj, sizeof(int) * aj.size(), hipMemcpyDeviceToHost));
                    AMG_HIP(hipMemcpy(ax.data(), L.A.Ax, sizeof(double) * ax.size(), hipMemcpyDeviceToHost));
                }
                L.sched_csr = std::make_shared<Schedule>();
                CHK(build_csr_schedule(ap.data(), aj.data(), ax.data(), n, nullptr, n, *L.sched_csr, h->stream, flow_allowed(h), L.A.ncols));
                h->dev_bytes += L.sched_csr->level_copy_bytes + L.sched_csr->flow.bytes;
            }
            s.sched = L.sched_csr;
        }
    } else if (s.kind == AMG_SM_GAUSS_SEIDEL_NE || s.kind == AMG_SM_GAUSS_SEIDEL_NR || s.kind == AMG_SM_JACOBI_NE) {
        if (!(L.fmt == AMG_FMT_CSR || (L.R == 1 && L.C == 1))) { set_error("normal-equation smoother: level operator must be CSR or BSR(1,1)"); return AMG_ENOTIMPL; }
        if (s.kind != AMG_SM_GAUSS_SEIDEL_NE && (!s.aux[0].Ap || s.aux[0].nrows != n || s.aux[0].nnz != L.A.nnz)) {
            set_error("gauss_seidel_nr / jacobi_ne: pass A by columns with amg_hier_set_aux_matrix(slot 0)");
            return AMG_ESTATE;
        }
        if (s.kind != AMG_SM_JACOBI_NE) {
            // tasks = rows of A (ne) / columns of A (nr); tasks sharing a vector entry keep their order
            const DevCsr &T = (s.kind == AMG_SM_GAUSS_SEIDEL_NE) ? L.A : s.aux[0];
            std::vector<int> ap((size_t)n + 1), aj((size_t)T.nnz), order;
            AMG_HIP(hipMemcpy(ap.data(), T.Ap, sizeof(int) * ap.size(), hipMemcpyDeviceToHost));
            if (T.nnz) AMG_HIP(hipMemcpy(aj.data(), T.Aj, sizeof(int) * aj.size(), hipMemcpyDeviceToHost));
            CHK(ne_touch_levels(n, ap.data(), aj.data(), n, s.sw_level_ptr, order));
            if (n) AMG_HIP(hipMemcpy(s.sw_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
        }
    } else if (s.kind == AMG_SM_SCHWARZ) {
        if (!(L.fmt == AMG_FMT_CSR || (L.R == 1 && L.C == 1))) { set_error("schwarz smoother: level operator must be CSR or BSR(1,1)"); return AMG_ENOTIMPL; }
        std::vector<int> ap((size_t)n + 1), aj((size_t)L.A.nnz);
        AMG_HIP(hipMemcpy(ap.data(), L.A.Ap, sizeof(int) * ap.size(), hipMemcpyDeviceToHost));
        if (L.A.nnz) AMG_HIP(hipMemcpy(aj.data(), L.A.Aj, sizeof(int) * aj.size(), hipMemcpyDeviceToHost));
        std::vector<int> tasks((size_t)s.nsd), order;
        for (int k = 0; k < s.nsd; ++k) tasks[k] = k;
        CHK(schwarz_levels(n, ap.data(), aj.data(), s.hSj.data(), s.hSp.data(), tasks, s.sw_level_ptr, order));
        if (s.nsd) AMG_HIP(hipMemcpy(s.sw_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
    } else if (s.kind == AMG_SM_GAUSS_SEIDEL_INDEXED) {
        if (!L.A.Ap) { set_error("gauss_seidel_indexed needs the scalar form of the level operator (BSR level kept by blocks only)"); return AMG_ENOTIMPL; }
        std::vector<int> ap((size_t)n + 1), aj((size_t)L.A.nnz);
        std::vector<double> ax((size_t)L.A.nnz);
        AMG_HIP(hipMemcpy(ap.data(), L.A.Ap, sizeof(int) * ap.size(), hipMemcpyDeviceToHost));
        if (L.A.nnz) {
            AMG_HIP(hipMemcpy(aj.data(), L.A.Aj, sizeof(int) * aj.size(), hipMemcpyDeviceToHost));
            AMG_HIP(hipMemcpy(ax.data(), L.A.Ax, sizeof(double) * ax.size(), hipMemcpyDeviceToHost));
        }
        s.sched = std::make_shared<Schedule>();
        CHK(build_csr_schedule(ap.data(), aj.data(), ax.data(), n, s.indices.data(), (int)s.indices.size(), *s.sched, h->stream, flow_allowed(h), L.A.ncols));
        h->dev_bytes += s.sched->level_copy_bytes + s.sched->flow.bytes;
    } else if (s.kind == AMG_SM_BLOCK_GAUSS_SEIDEL || s.kind == AMG_SM_BLOCK_JACOBI) {
        const DevBsr *Ab = s.Ablk_owned ? &s.Ablk : &L.Ab;
        if (!Ab->Ap || Ab->bs != s.bs) {
            set_error("block smoother: no BSR copy of A with the smoother's blocksize (amg_hier_set_block_matrix)");
            return AMG_ESTATE;
        }
        if (s.kind == AMG_SM_BLOCK_GAUSS_SEIDEL && !s.Ablk_owned && L.sched_bgs) {
            s.sched = L.sched_bgs;                                     // pre- and post-smoother sweep the same blocks
        } else if (s.kind == AMG_SM_BLOCK_GAUSS_SEIDEL) {
            std::vector<int> bp((size_t)Ab->nbrows + 1), bj((size_t)Ab->nblocks);
            std::vector<double> bx((size_t)Ab->nblocks * Ab->bs * Ab->bs);
            AMG_HIP(hipMemcpy(bp.data(), Ab->Ap, sizeof(int) * bp.size(), hipMemcpyDeviceToHost));
            if (!bj.empty()) {
                AMG_HIP(hipMemcpy(bj.data(), Ab->Aj, sizeof(int) * bj.size(), hipMemcpyDeviceToHost));
                AMG_HIP(hipMemcpy(bx.data(), Ab->Ax, sizeof(double) * bx.size(), hipMemcpyDeviceToHost));
            }
            s.sched = std::make_shared<Schedule>();
            CHK(build_block_schedule(bp.data(), bj.data(), Ab->nbrows, nullptr, Ab->nbrows, *s.sched, h->stream,
                                     bx.data(), Ab->bs, false, flow_allowed(h)));
            h->dev_bytes += s.sched->level_copy_bytes + s.sched->bflow.bytes;
            if (!s.Ablk_owned) L.sched_bgs = s.sched;
        }
    }
    return 0;
}

int amg_hier_set_callback_smoother(amg_hier *h, int lvl, int which, amg_relax_callback cb, void *user)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 1 || !cb) { set_error("bad callback smoother"); return AMG_EINVAL; }
    Smoother &s = h->lv[lvl].sm[which];
    free_smoother(s);
    s = Smoother();
    s.kind = AMG_SM_CALLBACK; s.cb = cb; s.cb_user = user;
    h->has_callbacks = true;
    h->finalized = false;
    return 0;
}

int amg_hier_set_coarse_callback(amg_hier *h, amg_coarse_callback cb, void *user)
{
    ENTER(h);
    if (!cb) { set_error("null coarse callback"); return AMG_EINVAL; }
    h->coarse_cb = cb; h->coarse_cb_user = user; h->coarse_kind = 3;
    h->has_callbacks = true;
    h->finalized = false;
    return 0;
}

int amg_hier_apply(amg_hier *h, int lvl, int which, const double *x_dev, double *y_dev)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2) { set_error("bad level/which"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    const DevCsr &M = (which == AMG_MAT_A) ? L.A : (which == AMG_MAT_P ? L.P : L.Rm);
    if (!M.Ap && !(M.blk && M.blk->Ap) && !M.st_vals && !M.sl_val) { set_error("operator not set"); return AMG_ESTATE; }
    return spmv(M, SM_MATVEC, x_dev, nullptr, nullptr, y_dev, nullptr, 0.0, h->stream);
}

int amg_hier_apply_aux(amg_hier *h, int lvl, int which, int slot, const double *x_dev, double *y_dev)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2 || slot < 0 || slot > 1) { set_error("bad level/which/slot"); return AMG_EINVAL; }
    Smoother &s = (which == 2) ? h->coarse_sm : h->lv[lvl].sm[which];
    if (!s.aux[slot].Ap) { set_error("auxiliary operator not set"); return AMG_ESTATE; }
    return spmv(s.aux[slot], SM_MATVEC, x_dev, nullptr, nullptr, y_dev, nullptr, 0.0, h->stream);
}

double *amg_hier_scratch(amg_hier *h)
{
    if (!h) return nullptr;
    if (!h->norm_scratch) {
        hipSetDevice(h->device);
        if (dev_alloc(&h->norm_scratch, 1024 + 8, &h->dev_bytes) != 0) return nullptr;
    }
    return h->norm_scratch;
}

/* ---- row-partitioned hierarchies (one process per GPU): see hier.hpp Partition, comm.hpp ---- */
int amg_hier_set_comm(amg_hier *h, amg_comm *comm, int reduce_channel)
{
    ENTER(h);
    if (!comm || comm->device != h->device) { set_error("communicator missing or on another device"); return AMG_EINVAL; }
    if (reduce_channel < 0 || reduce_channel >= (int)comm->ch.size()) { set_error("bad all-reduce channel"); return AMG_EINVAL; }
    h->comm = comm;
    h->reduce_channel = reduce_channel;
    h->finalized = false;
    return 0;
}

/* Level lvl of this rank: n_own owned entries followed by n_halo halo entries; `channel` is the exchange plan whose
 * receive side fills the halo (-1: the level needs no exchange: a replicated level, or one rank), send_idx (host,
 * the channel's send total) lists the owned entries the peers need, grouped by destination rank; rows [i0, i1) of
 * the local A read no halo entry. */
int amg_hier_set_partition(amg_hier *h, int lvl, int n_own, int n_halo, int channel, const int *send_idx, int i0, int i1)
{
    ENTER(h);
    if (!h->comm) { set_error("amg_hier_set_comm first"); return AMG_ESTATE; }
    if (lvl < 0 || lvl >= h->nlevels || n_own < 0 || n_halo < 0) { set_error("bad partition"); return AMG_EINVAL; }
    Partition &P = h->lv[lvl].part;
    if (P.send_idx) { hipFree(P.send_idx); P.send_idx = nullptr; }
    P.n_own = n_own; P.n_halo = n_halo; P.channel = channel;
    if (channel >= 0) {
        if (channel >= (int)h->comm->ch.size()) { set_error("bad channel"); return AMG_EINVAL; }
        const Channel &C = h->comm->ch[(size_t)channel];
        if (C.recv_total != n_halo) { set_error("channel does not deliver this level's halo"); return AMG_EINVAL; }
        for (int k = 0; k < C.send_total; ++k)
            if (send_idx[k] < 0 || send_idx[k] >= n_own) { set_error("send index outside the owned range"); return AMG_EINVAL; }
        CHK(dev_alloc(&P.send_idx, C.send_total, &h->dev_bytes));
        if (C.send_total) AMG_HIP(hipMemcpy(P.send_idx, send_idx, sizeof(int) * (size_t)C.send_total, hipMemcpyHostToDevice));
    }
    if (i0 < 0 || i1 > n_own || i1 < i0) { i0 = 0; i1 = 0; }
    P.i0 = i0; P.i1 = i1;
    P.overlap = channel >= 0 && n_halo > 0 && (double)(i1 - i0) >= 0.5 * (double)std::max(n_own, 1);
    {
        // the sliced form of this level's A: the rows one application covers in ONE launch -- the interior rows when
        // they run beside the exchange, else all of them
        DevCsr &A = h->lv[lvl].A;
        free_sell(A);
        if (A.Ap && !A.pat && !A.st_vals) {
            if (h->overlap && P.overlap) CHK(build_sell(A, &h->dev_bytes, i0, i1));
            else CHK(build_sell(A, &h->dev_bytes));
        }
    }
    h->finalized = false;
    return 0;
}

/* Level lvl is the last partitioned level: its restriction R computes `rows` coarse entries (this rank's slice) and
 * `channel` (every rank receives every rank's slice, its own included) assembles the whole coarse right-hand side
 * on every rank. */
int amg_hier_set_gather(amg_hier *h, int lvl, int channel, int rows)
{
    ENTER(h);
    if (!h->comm) { set_error("amg_hier_set_comm first"); return AMG_ESTATE; }
    if (lvl < 0 || lvl >= h->nlevels - 1 || channel < 0 || channel >= (int)h->comm->ch.size() || rows < 0) { set_error("bad gather"); return AMG_EINVAL; }
    Partition &P = h->lv[lvl].part;
    const Channel &C = h->comm->ch[(size_t)channel];
    if (C.send_total != rows * h->comm->world) { set_error("gather channel does not send this rank's slice to every rank"); return AMG_EINVAL; }
    if (P.gather_idx) hipFree(P.gather_idx);
    if (P.rslice) hipFree(P.rslice);
    P.gather_idx = nullptr; P.rslice = nullptr;
    std::vector<int> idx((size_t)C.send_total);
    for (int k = 0; k < C.send_total; ++k) idx[(size_t)k] = rows ? k % rows : 0;
    CHK(dev_alloc(&P.gather_idx, C.send_total, &h->dev_bytes));
    if (C.send_total) AMG_HIP(hipMemcpy(P.gather_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    CHK(dev_alloc(&P.rslice, rows, &h->dev_bytes));
    P.gather_channel = channel; P.gather_rows = rows;
    h->finalized = false;
    return 0;
}

/* The coarsest level is partitioned too and the coarse solver is a dense operator: `channel` gathers every rank's
 * slice of the coarse right-hand side on every rank; this rank's slice starts at entry `lo` of the whole vector. */
int amg_hier_set_coarse_gather(amg_hier *h, int channel, int lo)
{
    ENTER(h);
    if (!h->comm || channel < 0 || channel >= (int)h->comm->ch.size()) { set_error("bad coarse gather"); return AMG_EINVAL; }
    const Channel &C = h->comm->ch[(size_t)channel];
    const int rows = h->comm->world ? C.send_total / h->comm->world : 0;
    std::vector<int> idx((size_t)C.send_total);
    for (int k = 0; k < C.send_total; ++k) idx[(size_t)k] = rows ? k % rows : 0;
    if (h->coarse_gather_idx) hipFree(h->coarse_gather_idx);
    h->coarse_gather_idx = nullptr;
    CHK(dev_alloc(&h->coarse_gather_idx, C.send_total, &h->dev_bytes));
    if (C.send_total) AMG_HIP(hipMemcpy(h->coarse_gather_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    h->coarse_gather_channel = channel;
    h->coarse_lo = lo;
    h->finalized = false;
    return 0;
}

/* after a partitioned solve: non-zero (and an error message) if a peer never delivered */
int amg_hier_comm_check(amg_hier *h)
{
    ENTER(h);
    return h->comm ? comm_check(h->comm) : 0;
}

// Frees the CSR arrays of operators whose derived form (stencil / sliced) serves every application this hierarchy's
// cycles make of them: A_l applied from a complete stencil form under polynomial / Jacobi smoothers, A_l (l >= 1), P_l,
// R_l applied from a complete sliced form under polynomial smoothers.  Gauss-Seidel-type smoothers, partitioned
// hierarchies and operators with fall-back row ranges keep theirs.  The forms are lossless, so iterates are unchanged;
// afterwards the forms are used whatever the amg_set_stencil_form / amg_set_sell_form knobs say, and a smoother that
// needs the CSR arrays cannot be attached any more (AMG_ESTATE).
static bool csr_free_smoother(const Smoother &s, bool jacobi_ok)
{
    return s.kind == AMG_SM_NONE || s.kind == AMG_SM_POLYNOMIAL || (jacobi_ok && s.kind == AMG_SM_JACOBI);
}

static long release_csr_arrays(DevCsr &M)
{
    long freed = 0;
    if (M.Ap) { hipFree(M.Ap); freed += 4L * (M.nrows + 1 + PAD); }
    if (M.Aj) { hipFree(M.Aj); freed += 4L * (M.nnz + PAD); }
    if (M.Ax) { hipFree(M.Ax); freed += 8L * (M.nnz + PAD); }
    if (M.pat) { hipFree(M.pat); freed += 4L * M.nrows; }
    if (M.dict_ptr) hipFree(M.dict_ptr);
    if (M.dict_off) hipFree(M.dict_off);
    if (M.Aj16) { hipFree(M.Aj16); freed += 2L * M.nnz; }
    if (M.wg_base) hipFree(M.wg_base);
    if (M.wg_flag) hipFree(M.wg_flag);
    M.Ap = M.Aj = nullptr; M.Ax = nullptr; M.pat = nullptr; M.dict_ptr = M.dict_off = nullptr; M.npat = M.ndict = 0;
    M.Aj16 = nullptr; M.wg_base = nullptr; M.wg_flag = nullptr;
    return freed;
}

long amg_hier_release_sources(amg_hier *h)
{
    if (!h) return -1;
    if (hipSetDevice(h->device) != hipSuccess || !h->finalized || h->comm) return 0;
    if (h->stream) hipStreamSynchronize(h->stream);
    drop_graphs(h);
    long freed = 0;
    for (int l = 0; l < h->nlevels; ++l) {
        Level &L = h->lv[l];
        const bool last = l == h->nlevels - 1;
        const bool coarse_uses_A = last && h->coarse_kind != 1 && h->coarse_kind != 0;      // relaxation / callback coarse solvers
        if (L.fmt == AMG_FMT_BSR && L.R > 1) continue;                                      // block levels keep their arrays
        const bool sm_poly = last ? true : (csr_free_smoother(L.sm[0], false) && csr_free_smoother(L.sm[1], false));
        const bool sm_jac = last ? true : (csr_free_smoother(L.sm[0], true) && csr_free_smoother(L.sm[1], true));
        DevCsr &A = L.A;
        if (A.Ap && !coarse_uses_A && !A.blk) {
            const bool stencil_all = A.st_vals && A.st_nranges == 0;
            const bool sliced_all = A.sl_val && A.sl_lo == 0 && A.sl_hi == A.nrows && l > 0;    // (level 0 needs the fused norm)
            if ((stencil_all && sm_jac) || (sliced_all && sm_poly)) freed += release_csr_arrays(A);
        }
        if (!last) {
            for (DevCsr *M : {&L.P, &L.Rm})
                if (M->Ap && M->sl_val && M->sl_lo == 0 && M->sl_hi == M->nrows && !M->blk) freed += release_csr_arrays(*M);
        }
    }
    h->dev_bytes -= freed;
    return freed;
}

int amg_hier_finalize(amg_hier *h)
{
    ENTER(h);
    for (int l = 0; l < h->nlevels; ++l) {
        Level &L = h->lv[l];
        if (!L.hasA) { set_error("level " + std::to_string(l) + " has no A"); return AMG_ESTATE; }
        if (l < h->nlevels - 1) {
            if (!L.hasP || !L.hasR) { set_error("level " + std::to_string(l) + " lacks P or R"); return AMG_ESTATE; }
            Level &Lc = h->lv[l + 1];
            const int r_rows = (h->comm && L.part.gather_channel >= 0) ? L.part.gather_rows : Lc.A.nrows;
            if (!Lc.hasA || L.P.nrows != L.A.nrows || L.P.ncols != Lc.A.ncols || L.Rm.nrows != r_rows ||
                L.Rm.ncols != L.A.ncols) {
                set_error("operator shapes of level " + std::to_string(l) + " are inconsistent");
                return AMG_EINVAL;
            }
        }
        if (h->comm && (L.part.n_own != L.A.nrows || L.part.n_own + L.part.n_halo != L.A.ncols)) {
            set_error("level " + std::to_string(l) + ": partition does not match the local operator");
            return AMG_EINVAL;
        }
        const long n = L.A.ncols;                 // [owned | halo] on a partitioned level, = nrows otherwise
        double **vecs[] = {&L.x, &L.xalt, &L.b, &L.r, &L.h, &L.h2};
        for (double **p : vecs)
            if (!*p) CHK(dev_alloc(p, n, &h->dev_bytes));
        if (l < h->nlevels - 1) {
            CHK(need_schedule(h, L, L.sm[0]));
            CHK(need_schedule(h, L, L.sm[1]));
        }
    }
    if (h->coarse_kind == 2) CHK(need_schedule(h, h->lv[h->nlevels - 1], h->coarse_sm));
    if (h->coarse_kind == 1 && h->comm && h->coarse_gather_channel >= 0) {
        if (!h->coarse_full_b) CHK(dev_alloc(&h->coarse_full_b, h->coarse_n, &h->dev_bytes));
        if (!h->coarse_full_x) CHK(dev_alloc(&h->coarse_full_x, h->coarse_n, &h->dev_bytes));
        if (h->coarse_lo < 0 || h->coarse_lo + h->lv[h->nlevels - 1].A.nrows > h->coarse_n) {
            set_error("coarse slice outside the dense operator");
            return AMG_EINVAL;
        }
    } else if (h->coarse_kind == 1 && h->coarse_n != h->lv[h->nlevels - 1].A.nrows) {
        set_error("coarse dense operator has the wrong size");
        return AMG_EINVAL;
    }
    if (!h->norm_scratch) CHK(dev_alloc(&h->norm_scratch, 1024 + 8, &h->dev_bytes));
    if (h->comm) {
        if (h->reduce_channel < 0) { set_error("partitioned hierarchy without an all-reduce channel"); return AMG_ESTATE; }
        const char *env = getenv("AMG_DIST_OVERLAP");
        if (env) h->overlap = atoi(env);
    }
    {
        StreamArgs a0 = base_args(h->lv[0].A);
        long need = std::max(stream_blocks(a0), stencil_blocks(a0, h->lv[0].A)) + 8;
        if (h->lv[0].A.blk && h->lv[0].A.blk->Ap)
            need = std::max(need, (long)bsr_stream_blocks(block_spmv_args(*h->lv[0].A.blk, SM_RESIDUAL_SUMSQ, a0), h->lv[0].A.blk->nblocks) + 8);
        if (need > h->sumsq_cap) {
            if (h->sumsq_partials) hipFree(h->sumsq_partials);
            h->sumsq_partials = nullptr;
            CHK(dev_alloc(&h->sumsq_partials, need + 512, &h->dev_bytes));   // + second-stage scratch
            h->sumsq_cap = need;
        }
    }
    drop_graphs(h);
    {
        const char *env = getenv("AMG_HIP_GRAPHS");
        if (env) h->use_graphs = atoi(env);
        if (h->comm && h->comm->transport == 1) h->use_graphs = 0;      // library calls inside the iteration: run eagerly
    }
    h->finalized = true;
    return 0;
}

static int ensure_res(amg_hier *h, int cap)
{
    if (cap <= h->res_cap) return 0;
    if (h->res_dev) hipFree(h->res_dev);
    h->res_dev = nullptr;
    CHK(dev_alloc(&h->res_dev, cap, &h->dev_bytes));
    h->res_cap = cap;
    return 0;
}

static int load_vectors(amg_hier *h, const double *b, const double *x, int flags)
{
    Level &L0 = h->lv[0];
    const size_t bytes = sizeof(double) * (size_t)L0.A.nrows;
    hipMemcpyKind kind = (flags & AMG_SOLVE_DEVICE_VECTORS) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (b != L0.b) AMG_HIP(hipMemcpyAsync(L0.b, b, bytes, kind, h->stream));
    if (flags & AMG_SOLVE_X0_ZERO) AMG_HIP(hipMemsetAsync(L0.x, 0, bytes, h->stream));
    else if (x != L0.x) AMG_HIP(hipMemcpyAsync(L0.x, x, bytes, kind, h->stream));
    return 0;
}

static int store_x(amg_hier *h, double *x, int flags)
{
    Level &L0 = h->lv[0];
    const size_t bytes = sizeof(double) * (size_t)L0.A.nrows;
    hipMemcpyKind kind = (flags & AMG_SOLVE_DEVICE_VECTORS) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (x != L0.x) AMG_HIP(hipMemcpyAsync(x, L0.x, bytes, kind, h->stream));
    AMG_HIP(hipStreamSynchronize(h->stream));
    // a dataflow Gauss-Seidel sweep whose waves gave up waiting left unusable iterates behind: say so
    if (gs_flow_status() != 0) { set_error("a dataflow Gauss-Seidel sweep ran out of its time budget (amg_set_gs_flow(0) selects the level-scheduled sweeps)"); return AMG_ESTATE; }
    return 0;
}

int amg_hier_solve(amg_hier *h, const double *b, double *x, double tol, int maxiter, int cyc,
                   double *residuals, int *nres, int flags)
{
    ENTER(h);
    if (!h->finalized) { set_error("hierarchy not finalised"); return AMG_ESTATE; }
    if (!b || !x || !residuals || !nres || maxiter < 0) { set_error("bad solve arguments"); return AMG_EINVAL; }
    Level &L0 = h->lv[0];
    hipStream_t st = h->stream;
    CHK(ensure_res(h, maxiter + 2));
    CHK(load_vectors(h, b, x, flags));

    // tol *= norm(b)  (multilevel.py:427-429)
    double normb = 0.0;
    CHK(global_norm(h, L0.b, L0.A.nrows, h->res_dev + maxiter + 1));
    AMG_HIP(hipMemcpyAsync(&normb, h->res_dev + maxiter + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    CHK(residual_norm_to(h, h->res_dev));                                         // :450
    AMG_HIP(hipMemcpyAsync(&residuals[0], h->res_dev, sizeof(double), hipMemcpyDeviceToHost, st));
    AMG_HIP(hipStreamSynchronize(st));
    if (normb != 0.0) tol = tol * normb;

    int k = 1;
    bool x_zero = (flags & AMG_SOLVE_X0_ZERO) != 0;
    const bool fixed = (flags & AMG_SOLVE_NO_EARLY_STOP) != 0;
    AMG_HIP(hipEventRecord(h->ev0, st));
    while (k <= maxiter && (fixed || residuals[k - 1] > tol)) {                   // :454
        CHK(graph_iteration(h, cyc, x_zero, h->res_dev + k));                     // :459-461
        x_zero = false;
        if (!fixed) {
            AMG_HIP(hipMemcpyAsync(&residuals[k], h->res_dev + k, sizeof(double), hipMemcpyDeviceToHost, st));
            AMG_HIP(hipStreamSynchronize(st));
        }
        ++k;
    }
    AMG_HIP(hipEventRecord(h->ev1, st));
    if (fixed && k > 1)
        AMG_HIP(hipMemcpyAsync(residuals + 1, h->res_dev + 1, sizeof(double) * (size_t)(k - 1), hipMemcpyDeviceToHost, st));
    *nres = k;
    CHK(store_x(h, x, flags));
    float ms = 0.f;
    AMG_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_ms = ms;
    return 0;
}

// Preconditioned conjugate gradients with one multigrid cycle (from a zero guess) as M:
// multilevel_solver.solve(accel='cg') = pyamg/krylov/_cg.py:84-179 with
// M = aspreconditioner(cycle) (pyamg/multilevel.py:306-314), all vectors in HBM.  residuals[k] is the
// preconditioner norm sqrt(<r_k, M r_k>) as in the reference.  *info: 0 converged / maxiter reached,
// -1 indefinite operator or preconditioner (the reference warns and stops).
int amg_hier_pcg(amg_hier *h, const double *b, double *x, double tol, int maxiter, int cyc,
                 double *residuals, int *nres, int *info, int flags)
{
    ENTER(h);
    if (!h->finalized) { set_error("hierarchy not finalised"); return AMG_ESTATE; }
    if (!b || !x || !residuals || !nres || !info || maxiter < 1) { set_error("bad pcg arguments"); return AMG_EINVAL; }
    if (cyc == AMG_CYCLE_AMLI) { set_error("AMLI cycles require fgmres or no acceleration"); return AMG_EINVAL; }
    if (h->comm) { set_error("the device PCG has no row-partitioned form yet"); return AMG_ENOTIMPL; }
    Level &L0 = h->lv[0];
    hipStream_t st = h->stream;
    const long n = L0.A.nrows;
    const size_t bytes = sizeof(double) * (size_t)n;
    for (int q = 0; q < 4; ++q)
        if (!h->pcg[q]) CHK(dev_alloc(&h->pcg[q], n, &h->dev_bytes));
    double *xk = h->pcg[0], *r = h->pcg[1], *p = h->pcg[2], *Ap = h->pcg[3];
    double *slot = h->norm_scratch + 1026;
    hipMemcpyKind in_kind = (flags & AMG_SOLVE_DEVICE_VECTORS) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipMemcpyKind out_kind = (flags & AMG_SOLVE_DEVICE_VECTORS) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    // b is kept in L0.b (the cycle is always called with rhs = r, so L0.b is free; L0.r is the
    // cycle's own residual scratch and must not be used here)
    double *bdev = L0.b;
    AMG_HIP(hipMemcpyAsync(bdev, b, bytes, in_kind, st));
    if (flags & AMG_SOLVE_X0_ZERO) AMG_HIP(hipMemsetAsync(xk, 0, bytes, st));
    else AMG_HIP(hipMemcpyAsync(xk, x, bytes, in_kind, st));
    auto fetch = [&](double *dst) -> int {
        AMG_HIP(hipMemcpyAsync(dst, slot, sizeof(double), hipMemcpyDeviceToHost, st));
        AMG_HIP(hipStreamSynchronize(st));
        return 0;
    };
    auto dot = [&](const double *u, const double *v, double *out) -> int {
        CHK(launch_dot(u, v, n, h->norm_scratch, slot, st));
        return fetch(out);
    };
    auto precond = [&](const double *rhs) -> int {      // z = M rhs -> L0.x
        AMG_HIP(hipMemsetAsync(L0.x, 0, bytes, st));
        if (h->nlevels == 1) return coarse_solve(h, rhs, L0.x, L0.xalt);
        return cycle(h, 0, L0.x, L0.xalt, rhs, cyc, true);
    };
    int k = 0;
    *info = 0;
    CHK(spmv(L0.A, SM_RESIDUAL, xk, bdev, nullptr, r, nullptr, 0.0, st));          // r = b - A x
    CHK(precond(r));                                                              // z = M r
    AMG_HIP(hipMemcpyAsync(p, L0.x, bytes, hipMemcpyDeviceToDevice, st));          // p = z.copy()
    double rz = 0.0, normb = 0.0;
    CHK(dot(r, L0.x, &rz));
    double normr = std::sqrt(rz);
    residuals[k++] = normr;
    CHK(launch_norm2(bdev, n, h->norm_scratch, slot, st));
    CHK(fetch(&normb));
    if (normb == 0.0) normb = 1.0;
    bool done = normr < tol * normb;
    if (!done && normr != 0.0) tol = tol * normr;
    const int recompute_r = 8;
    int iter = 0;
    AMG_HIP(hipEventRecord(h->ev0, st));
    while (!done) {
        CHK(spmv(L0.A, SM_MATVEC, p, nullptr, nullptr, Ap, nullptr, 0.0, st));     // Ap = A p
        const double rz_old = rz;
        double pAp = 0.0;
        CHK(dot(Ap, p, &pAp));
        if (pAp < 0.0) { *info = -1; break; }
        const double alpha = rz / pAp;
        CHK(launch_axmy(xk, p, -alpha, n, st));                                    // x += alpha p
        if ((iter % recompute_r) && iter > 0) CHK(launch_axmy(r, Ap, alpha, n, st));   // r -= alpha Ap
        else CHK(spmv(L0.A, SM_RESIDUAL, xk, bdev, nullptr, r, nullptr, 0.0, st));
        CHK(precond(r));
        CHK(dot(r, L0.x, &rz));
        if (rz < 0.0) { *info = -1; break; }
        const double beta = rz / rz_old;
        CHK(launch_scale_add(p, beta, L0.x, n, st));                               // p = beta p + z
        ++iter;
        normr = std::sqrt(rz);
        if (k <= maxiter) residuals[k++] = normr;
        if (normr < tol) break;
        if (rz == 0.0) { *info = -1; break; }
        if (iter >= maxiter) break;                                               // scipy-style cap (the reference has none)
    }
    AMG_HIP(hipEventRecord(h->ev1, st));
    *nres = k;
    AMG_HIP(hipMemcpyAsync(x, xk, bytes, out_kind, st));
    AMG_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    AMG_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_ms = ms;
    return 0;
}

int amg_hier_cycle(amg_hier *h, const double *b, double *x, int cyc, int flags)
{
    ENTER(h);
    if (!h->finalized) { set_error("hierarchy not finalised"); return AMG_ESTATE; }
    CHK(load_vectors(h, b, x, flags));
    CHK(one_iteration(h, cyc, (flags & AMG_SOLVE_X0_ZERO) != 0));
    return store_x(h, x, flags);
}

int amg_hier_relax(amg_hier *h, int lvl, int which, const double *b, double *x)
{
    ENTER(h);
    if (!h->finalized) { set_error("hierarchy not finalised"); return AMG_ESTATE; }
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2) { set_error("bad level/which"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    Smoother &s = (which == 2) ? h->coarse_sm : L.sm[which];
    const size_t bytes = sizeof(double) * (size_t)L.A.nrows;
    AMG_HIP(hipMemcpyAsync(L.b, b, bytes, hipMemcpyHostToDevice, h->stream));
    AMG_HIP(hipMemcpyAsync(L.x, x, bytes, hipMemcpyHostToDevice, h->stream));
    CHK(relax(h, L, s, L.x, L.xalt, L.b, false));
    AMG_HIP(hipMemcpyAsync(x, L.x, bytes, hipMemcpyDeviceToHost, h->stream));
    AMG_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

int amg_hier_matvec(amg_hier *h, int lvl, int which, const double *x, double *y)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2) { set_error("bad level/which"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    const DevCsr &M = (which == AMG_MAT_A) ? L.A : (which == AMG_MAT_P ? L.P : L.Rm);
    if (!M.Ap && !(M.blk && M.blk->Ap) && !M.st_vals && !M.sl_val) { set_error("operator not set"); return AMG_ESTATE; }
    double *dx = nullptr, *dy = nullptr;
    CHK(dev_alloc(&dx, M.ncols, (long *)nullptr));
    CHK(dev_alloc(&dy, M.nrows, (long *)nullptr));
    AMG_HIP(hipMemcpy(dx, x, sizeof(double) * (size_t)M.ncols, hipMemcpyHostToDevice));
    int rc = spmv(M, SM_MATVEC, dx, nullptr, nullptr, dy, nullptr, 0.0, h->stream);
    if (rc == 0) {
        hipStreamSynchronize(h->stream);
        hipMemcpy(y, dy, sizeof(double) * (size_t)M.nrows, hipMemcpyDeviceToHost);
    }
    hipFree(dx); hipFree(dy);
    return rc;
}

// algorithmic bytes, SURVEY.md 8(d): bytes_spmv(M) = 12 nnz + 4 (rows+1) + 8 cols + 8 rows
static double bytes_spmv(const DevCsr &M)
{
    if (M.blk && M.blk->Ap) {
        // SURVEY 8(d), BSR(bs x bs): 8 nnz + 4 nnz / bs^2 for the matrix stream, 4 (rows / bs + 1) for the row pointer
        const double nb = (double)M.blk->nblocks, b2 = (double)M.blk->bs * M.blk->bs;
        return 8.0 * nb * b2 + 4.0 * nb + 4.0 * (M.blk->nbrows + 1.0) + 8.0 * M.ncols + 8.0 * M.nrows;
    }
    return 12.0 * (double)M.nnz + 4.0 * (M.nrows + 1.0) + 8.0 * M.ncols + 8.0 * M.nrows;
}

static double smoother_apps(const Smoother &s, bool x_zero)
{
    // number of "A-applications" (bytes_spmv(A)+8n each) of one smoother call
    switch (s.kind) {
    case AMG_SM_JACOBI: case AMG_SM_BLOCK_JACOBI: return s.iterations;
    case AMG_SM_GAUSS_SEIDEL: case AMG_SM_GAUSS_SEIDEL_INDEXED: case AMG_SM_BLOCK_GAUSS_SEIDEL: case AMG_SM_SOR:
        return s.iterations * (s.sweep == AMG_SWEEP_SYMMETRIC ? 2.0 : 1.0);
    case AMG_SM_POLYNOMIAL: {
        double d = (double)s.coef.size();
        return s.iterations * d - (x_zero ? 1.0 : 0.0);
    }
    default: return 0.0;
    }
}

// Bytes of one operator application as THIS library stores the operator: the offset-pattern form
// streams the values, one pattern id per row and the row pointer; the column indices are implied.
static double bytes_spmv_moved(const DevCsr &M)
{
    if (M.blk && M.blk->Ap && (!M.Ap || bsr_spmv_enabled(M.blk->bs))) return bytes_spmv(M);
    if (M.st_vals && (stencil_enabled() || !M.Ap))     // padded values (or one-byte codes) + one mask word per row; no row pointer
    {
        const bool coded = M.st_vi_on && M.st_codes;
        // (the coded kernel of stencils with <= 7 offsets finds "absent" in the codes and reads no row mask unless rows are left to the pattern kernel)
        const double mask_bytes = (coded && M.st_nu <= 7 && M.st_nranges == 0) ? 0.0 : (M.st_nu <= 7 ? 1.0 : 4.0);
        return (coded ? 8.0 * (double)((M.st_nu + 7) / 8) : 8.0 * (double)M.st_nu) * 256.0 * (double)((M.nrows + 255) / 256) + mask_bytes * M.nrows +
               8.0 * M.ncols + 8.0 * M.nrows;
    }
    if (!M.pat && M.sl_val && (sell_enabled() || !M.Ap) && M.sl_lo == 0 && M.sl_hi == M.nrows)    // padded entries, slot bookkeeping (row id + length), slice offsets; no row pointer
    {
        // (16-bit column codes: 2 B instead of 4 B per entry of a coded slice, + its 16 window origins)
        const double f16 = (M.sl_code && sell_index16_enabled()) ? M.sl_frac16 : 0.0;
        return (12.0 - 2.0 * f16) * (double)M.sl_entries + (6.0 * 64.0 + 64.0 * f16 + 1.0) * M.sl_nslices + 8.0 * (M.sl_nslices + 1.0) + 8.0 * M.ncols + 8.0 * M.nrows;
    }
    if (!M.pat) {
        double idx = (M.Aj16 && index16_enabled()) ? (2.0 * M.i16_frac + 4.0 * (1.0 - M.i16_frac)) : 4.0;
        return (8.0 + idx) * (double)M.nnz + 4.0 * (M.nrows + 1.0) + 8.0 * M.ncols + 8.0 * M.nrows;
    }
    return 8.0 * (double)M.nnz + 4.0 * M.nrows + 4.0 * (M.nrows + 1.0) + 8.0 * M.ncols + 8.0 * M.nrows;
}

static double cycle_bytes_impl(amg_hier *h, int cyc, bool moved)
{
    if (!h) return 0.0;
    // visits per level for V/W/F
    std::vector<double> visits((size_t)h->nlevels, 0.0);
    visits[0] = 1.0;
    for (int l = 1; l < h->nlevels; ++l) {
        if (cyc == AMG_CYCLE_W) visits[l] = 2.0 * visits[l - 1];
        else if (cyc == AMG_CYCLE_F) visits[l] = visits[l - 1] + 1.0;
        else visits[l] = 1.0;
    }
    double total = 0.0;
    for (int l = 0; l < h->nlevels - 1; ++l) {
        Level &L = h->lv[l];
        double n = L.A.nrows;
        double app = (moved ? bytes_spmv_moved(L.A) : bytes_spmv(L.A)) + 8.0 * n;
        // every visit: pre + residual + post; coarse levels enter with x == 0 on first of
        // each pair of visits -- counted as zero-start for all coarse presmooths of a V cycle
        double k = smoother_apps(L.sm[0], l > 0) + 1.0 + smoother_apps(L.sm[1], false);
        double extra = 0.0;
        if (l == 0) {
            k += 1.0;   // outer residual norm (multilevel.py:461)
            if (moved && h->keep_residual && L.sm[0].kind == AMG_SM_POLYNOMIAL && L.sm[0].iterations >= 1) {
                k -= 1.0;            // the pre-smoother starts from the kept residual ...
                extra = 8.0 * n;     // ... which the norm pass wrote
            }
        }
        total += visits[l] * (k * app + extra + bytes_spmv(L.Rm) + bytes_spmv(L.P) + 8.0 * n);
    }
    double ncs = h->lv[h->nlevels - 1].A.nrows;
    total += visits[h->nlevels - 1] * (8.0 * ncs * ncs + 16.0 * ncs);
    return total;
}

/* on > 0: build (if need be) and use level lvl's value index, returns the number of distinct values (0: not applicable);
 * on == 0: back to the 8-byte values; on < 0: query -- the number of distinct values when the index is in use, else 0. */
int amg_hier_value_index(amg_hier *h, int lvl, int on)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels) { set_error("bad level"); return AMG_EINVAL; }
    DevCsr &M = h->lv[lvl].A;
    if (on < 0) return (M.st_vi_on && M.st_codes) ? M.st_ndict : 0;
    drop_graphs(h);
    if (!on) { M.st_vi_on = false; return 0; }
    return value_index_build(h, M);
}

int amg_hier_operator_form(amg_hier *h, int lvl)
{
    if (!h || lvl < 0 || lvl >= h->nlevels) return -1;
    const DevCsr &M = h->lv[lvl].A;
    if (M.st_vals && (stencil_enabled() || !M.Ap)) return 2;
    if (M.pat) return 1;
    return (M.sl_val && (sell_enabled() || !M.Ap)) ? 3 : 0;      // 3: sliced form (sell.hip)
}
double amg_hier_operator_bytes(amg_hier *h, int lvl, int moved)
{
    if (!h || lvl < 0 || lvl >= h->nlevels) return 0.0;
    const DevCsr &M = h->lv[lvl].A;
    return (moved ? bytes_spmv_moved(M) : bytes_spmv(M)) + 8.0 * M.nrows;     // + the right-hand side of r = b - A x
}
double amg_hier_cycle_bytes(amg_hier *h, int cyc) { return cycle_bytes_impl(h, cyc, false); }
double amg_hier_cycle_bytes_moved(amg_hier *h, int cyc) { return cycle_bytes_impl(h, cyc, true); }

double amg_hier_last_solve_ms(amg_hier *h) { return h ? h->last_ms : 0.0; }
long amg_hier_device_bytes(amg_hier *h) { return h ? h->dev_bytes : 0; }
void *amg_hier_stream(amg_hier *h) { return h ? (void *)h->stream : nullptr; }
double *amg_hier_dev_x(amg_hier *h) { return h && h->finalized ? h->lv[0].x : nullptr; }
double *amg_hier_dev_b(amg_hier *h) { return h && h->finalized ? h->lv[0].b : nullptr; }

int amg_hier_time_spmv(amg_hier *h, int lvl, int which, int mode, int reps, double *ms)
{
    ENTER(h);
    if (!h->finalized) { set_error("hierarchy not finalised"); return AMG_ESTATE; }
    if (lvl < 0 || lvl >= h->nlevels || reps < 1 || !ms) { set_error("bad arguments"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    const DevCsr &M = (which == AMG_MAT_A) ? L.A : (which == AMG_MAT_P ? L.P : L.Rm);
    if (!M.Ap && !(M.blk && M.blk->Ap) && !M.st_vals && !M.sl_val) { set_error("operator not set"); return AMG_ESTATE; }
    // vectors: A: x -> r ; P: coarse x -> h ; R: r -> coarse b
    const double *in; double *out;
    if (which == AMG_MAT_A) { in = L.x; out = L.r; }
    else if (which == AMG_MAT_P) { in = h->lv[lvl + 1].x; out = L.h; }
    else { in = L.r; out = h->lv[lvl + 1].b; }
    StreamMode sm = ((mode & 1) && which == AMG_MAT_A) ? SM_RESIDUAL : SM_MATVEC;
    if ((mode & 6) && !M.Ap) { set_error("operator is stored by blocks only"); return AMG_ENOTIMPL; }
    DevCsr Mplain = M;                     // mode bit 1 (value 2): time the plain CSR kernel even if
    if (mode & 2) { Mplain.pat = nullptr; Mplain.st_vals = nullptr; Mplain.Aj16 = nullptr; Mplain.blk = nullptr; }   // the operator has derived forms;
    if (mode & 4) Mplain.st_vals = nullptr;                             // bit 2 (value 4): the pattern kernel
    const DevCsr &Mu = (mode & 6) ? Mplain : M;
    CHK(spmv(Mu, sm, in, L.b, nullptr, out, nullptr, 0.0, h->stream));   // warm-up
    AMG_HIP(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) CHK(spmv(Mu, sm, in, L.b, nullptr, out, nullptr, 0.0, h->stream));
    AMG_HIP(hipEventRecord(h->ev1, h->stream));
    AMG_HIP(hipStreamSynchronize(h->stream));
    float t = 0.f;
    AMG_HIP(hipEventElapsedTime(&t, h->ev0, h->ev1));
    *ms = t / reps;
    return 0;
}

/* Setup-side: `nsweeps` Gauss-Seidel sweeps of A x = b (b == NULL: zero right-hand side) over level lvl's operator in ITS OWN
 * row order, from the CSR arrays in HBM (gsflow.hip: gs_natural_kernel) -- relaxation.h:34-62 bit for bit; what the
 * candidate improvement of aggregation.py:313-320 runs.  dirs[k] != 0: descending rows.  x (host, n doubles) is updated in
 * place.  AMG_ENOTIMPL when the operator has a row of more than 8 entries or no CSR arrays on the device (the caller keeps
 * its host sweep). */
int amg_hier_gs_natural(amg_hier *h, int lvl, double *x, const double *b, const unsigned char *dirs, int nsweeps)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || !x || !dirs || nsweeps < 0) { set_error("bad natural-order Gauss-Seidel arguments"); return AMG_EINVAL; }
    DevCsr &M = h->lv[lvl].A;
    if (!M.Ap || !M.Aj || !M.Ax || M.nrows != M.ncols) { set_error("natural-order Gauss-Seidel: no square CSR operator on the device"); return AMG_ENOTIMPL; }
    const int n = M.nrows;
    if (M.longest_row < 0) {
        std::vector<int> ap((size_t)n + 1);
        AMG_HIP(hipMemcpy(ap.data(), M.Ap, sizeof(int) * ap.size(), hipMemcpyDeviceToHost));
        int lr = 0;
        for (int i = 0; i < n; ++i) lr = std::max(lr, ap[(size_t)i + 1] - ap[(size_t)i]);
        M.longest_row = lr;
    }
    if (M.longest_row > 8) { set_error("natural-order Gauss-Seidel: rows of more than 8 entries"); return AMG_ENOTIMPL; }
    // Tasks and their order.  An operator in stencil form (every row's columns are i + off[u], the row mask says which are
    // stored): a task is a run of at most TASK_ROWS consecutive positions of the sweep, cut where a row does NOT take the value of
    // its predecessor (offset -1 forward, +1 backward: a grid line starts there) -- so that the only chain between
    // neighbouring tasks is the one inside a grid line.  The dependency level of a task then follows from the offsets:
    // its rows minus a distance d >= 64 lie in a few earlier tasks (found with a pointer that only moves forward).  Tasks
    // sorted by level (stable) = an order in which dependencies come first and the tasks of a level are independent.
    // Without that structure only small operators are swept (ascending order offers too little to do, gsflow.hip).
    const int ntasks64 = (n + 63) / 64;
    // rows per task of a structured operator: a task is one chain of hand-offs (0.7 us each at 500^3), a sweep is
    // (tasks per grid line + lines per plane + planes) of them; 32 rows leave about one task per resident wave and level
    constexpr long TASK_ROWS = 32;
    int *order_dev[2] = {nullptr, nullptr}, *tstart_dev[2] = {nullptr, nullptr};
    int ntasks_dir[2] = {0, 0};
    const bool structured = M.st_vals && M.st_nranges == 0 && M.st_nu <= 7;
    if (!structured && ntasks64 > 4096) { set_error("natural-order Gauss-Seidel: large operator without stencil structure"); return AMG_ENOTIMPL; }
    if (structured && ntasks64 > 64) {
        std::vector<unsigned char> mk((size_t)n);
        AMG_HIP(hipMemcpy(mk.data(), M.st_mask, (size_t)n, hipMemcpyDeviceToHost));
        auto release = [&]() { for (int k = 0; k < 2; ++k) { if (order_dev[k]) hipFree(order_dev[k]); if (tstart_dev[k]) hipFree(tstart_dev[k]); order_dev[k] = tstart_dev[k] = nullptr; } };
        for (int dir = 0; dir < 2; ++dir) {
            bool wanted = false;
            for (int k = 0; k < nsweeps; ++k) wanted |= ((dirs[k] != 0) == (dir == 1));
            if (!wanted) continue;
            int u_near = -1;                                       // the slot of offset -1 (forward) / +1 (backward)
            std::vector<long> dist;                                // distances >= 2 to operands the sweep has already renewed
            std::vector<int> dist_u;                               // ... and their slots
            bool supported = true;
            for (int u = 0; u < M.st_nu; ++u) {
                const long o = M.st_off[u];
                const long d = dir == 0 ? -o : o;
                if (d <= 0) continue;
                if (d == 1) u_near = u;
                else {
                    // a second short-range dependency (a grid whose lines are shorter than a task) chains neighbouring tasks
                    // whatever the cuts: correct, but only small operators get through in reasonable time
                    if (d < TASK_ROWS && ntasks64 > 4096) supported = false;
                    dist.push_back(d);
                    dist_u.push_back(u);
                }
            }
            if (!supported) { release(); set_error("natural-order Gauss-Seidel: large stencil operator with several short-range offsets"); return AMG_ENOTIMPL; }
            auto row_of = [&](long p) { return dir == 0 ? p : (long)n - 1 - p; };
            std::vector<int> ts;
            std::vector<unsigned char> any;                         // per task: the OR of its rows' masks (which offsets occur at all)
            ts.reserve(2 * (size_t)ntasks64 + (size_t)ntasks64 / 4 + 2);
            any.reserve(2 * (size_t)ntasks64 + (size_t)ntasks64 / 4 + 2);
            for (long p = 0; p < n;) {
                const long begin = p;
                ts.push_back((int)begin);
                unsigned char acc = mk[(size_t)row_of(p)];
                ++p;
                while (p < n && p - begin < TASK_ROWS) {
                    const unsigned char m = mk[(size_t)row_of(p)];
                    if (u_near >= 0 && !((m >> u_near) & 1u)) break;
                    acc |= m;
                    ++p;
                }
                any.push_back(acc);
            }
            const int nt = (int)ts.size();
            ts.push_back(n);
            std::vector<int> level((size_t)nt, 0);
            std::vector<int> ptr(dist.size(), 0);
            int top = 0;
            for (int q = 0; q < nt; ++q) {
                int lv = 0;
                const long b0 = ts[(size_t)q], b1 = (long)ts[(size_t)q + 1] - 1;
                if (q > 0 && u_near >= 0 && ((mk[(size_t)row_of(b0)] >> u_near) & 1u)) lv = level[(size_t)q - 1] + 1;
                for (size_t k = 0; k < dist.size(); ++k) {
                    // (a task whose rows do not store that offset -- the first line of a grid plane, the first plane -- does not
                    //  depend on what lies that far back: assuming it did would chain the planes one after the other)
                    if (!((any[(size_t)q] >> dist_u[k]) & 1u)) continue;
                    const long lo = b0 - dist[k], hi = std::min(b1 - dist[k], b0 - 1);      // (positions >= b0 are lanes of this task)
                    if (hi < 0) continue;
                    int &t = ptr[k];
                    const long lo0 = std::max(lo, 0L);
                    while (ts[(size_t)t + 1] <= lo0) ++t;          // the task that holds position lo0
                    for (int tt = t; tt < q && ts[(size_t)tt] <= hi; ++tt) lv = std::max(lv, level[(size_t)tt] + 1);
                }
                level[(size_t)q] = lv;
                top = std::max(top, lv);
            }
            std::vector<int> first((size_t)top + 2, 0), order((size_t)nt);
            for (int q = 0; q < nt; ++q) ++first[(size_t)level[(size_t)q] + 1];
            for (int l = 0; l <= top; ++l) first[(size_t)l + 1] += first[(size_t)l];
            for (int q = 0; q < nt; ++q) order[(size_t)first[(size_t)level[(size_t)q]]++] = q;
            hipError_t eo = hipMalloc((void **)&order_dev[dir], sizeof(int) * (size_t)nt);
            if (eo == hipSuccess) eo = hipMalloc((void **)&tstart_dev[dir], sizeof(int) * ((size_t)nt + 1));
            if (eo == hipSuccess) eo = hipMemcpy(order_dev[dir], order.data(), sizeof(int) * (size_t)nt, hipMemcpyHostToDevice);
            if (eo == hipSuccess) eo = hipMemcpy(tstart_dev[dir], ts.data(), sizeof(int) * ((size_t)nt + 1), hipMemcpyHostToDevice);
            if (eo != hipSuccess) { release(); return hip_fail(eo, "natural-order Gauss-Seidel: task order to the device", __FILE__, __LINE__); }
            ntasks_dir[dir] = nt;
        }
    }
    double *dx = nullptr, *db = nullptr;
    int rc = 0;
    hipError_t e = hipMalloc((void **)&dx, sizeof(double) * (size_t)std::max(n, 1));
    if (e == hipSuccess) e = hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) {
        e = hipMalloc((void **)&db, sizeof(double) * (size_t)std::max(n, 1));
        if (e == hipSuccess) e = hipMemcpy(db, b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) rc = hip_fail(e, "natural-order Gauss-Seidel: vectors to the device", __FILE__, __LINE__);
    if (rc == 0) rc = gs_natural_sweeps(M.Ap, M.Aj, M.Ax, n, M.longest_row, dx, db, dirs, nsweeps, h->stream, order_dev[0], order_dev[1],
                                        tstart_dev[0], tstart_dev[1], ntasks_dir[0], ntasks_dir[1]);
    if (rc == 0 && gs_flow_status() != 0) { set_error("natural-order Gauss-Seidel gave up waiting for an operand"); rc = AMG_ESTATE; }
    if (rc == 0) {
        e = hipMemcpy(x, dx, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "natural-order Gauss-Seidel: result to the host", __FILE__, __LINE__);
    }
    if (dx) hipFree(dx);
    if (db) hipFree(db);
    for (int k = 0; k < 2; ++k) { if (order_dev[k]) hipFree(order_dev[k]); if (tstart_dev[k]) hipFree(tstart_dev[k]); }
    return rc == -40 ? AMG_ENOTIMPL : rc;
}

// ---- setup-time helper: Arnoldi on a stored operator (pyamg/util/linalg.py:173-279) ----------
// Krylov basis of M = diag(dinv) * A_lvl (dinv == NULL: M = A_lvl) from the start vector v0,
// modified Gram-Schmidt exactly in the reference's order; H is (maxiter+1) x maxiter row-major on
// the host.  The basis stays on the device for amg_arnoldi_combine.  *steps = columns computed;
// *breakdown = 1 if H[j+1][j] fell below `breakdown_tol`.
int amg_arnoldi(amg_hier *h, int lvl, const double *dinv, const double *v0, int maxiter,
                double breakdown_tol, double *H, int *steps, int *breakdown)
{
    ENTER(h);
    if (lvl < 0 || lvl >= h->nlevels || maxiter < 1 || !v0 || !H || !steps || !breakdown) { set_error("bad arnoldi arguments"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    if (!L.hasA) { set_error("level has no A"); return AMG_ESTATE; }
    const long n = L.A.nrows;
    hipStream_t st = h->stream;
    if (maxiter > n) maxiter = (int)n;
    if (h->arn_m < maxiter || h->arn_n != n) {
        if (h->arn_V) hipFree(h->arn_V);
        if (h->arn_dinv) hipFree(h->arn_dinv);
        if (h->arn_coef) hipFree(h->arn_coef);
        h->arn_V = h->arn_dinv = h->arn_coef = nullptr;
        h->dev_bytes -= h->arn_bytes;
        h->arn_bytes = 0;
        CHK(dev_alloc(&h->arn_V, (long)(maxiter + 1) * n, &h->arn_bytes));
        CHK(dev_alloc(&h->arn_dinv, n, &h->arn_bytes));
        CHK(dev_alloc(&h->arn_coef, maxiter + 8, &h->arn_bytes));
        h->dev_bytes += h->arn_bytes;                   // counted while it lives (amg_arnoldi_free gives it back)
        h->arn_m = maxiter; h->arn_n = n;
    }
    if (!h->norm_scratch) CHK(dev_alloc(&h->norm_scratch, 1024 + 8, &h->dev_bytes));
    double *V = h->arn_V;
    double *slot = h->norm_scratch + 1026;
    if (dinv) AMG_HIP(hipMemcpyAsync(h->arn_dinv, dinv, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    AMG_HIP(hipMemcpyAsync(V, v0, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    auto fetch = [&](double *dst) -> int {
        AMG_HIP(hipMemcpyAsync(dst, slot, sizeof(double), hipMemcpyDeviceToHost, st));
        AMG_HIP(hipStreamSynchronize(st));
        return 0;
    };
    double nv = 0.0;
    CHK(launch_norm2(V, n, h->norm_scratch, slot, st));
    CHK(fetch(&nv));
    CHK(launch_divide(V, nv, n, st));                               // v0 /= norm(v0)
    for (long q = 0; q < (long)(maxiter + 1) * maxiter; ++q) H[q] = 0.0;
    *breakdown = 0;
    int j = 0;
    for (j = 0; j < maxiter; ++j) {
        double *w = V + (long)(j + 1) * n;
        CHK(spmv(L.A, SM_MATVEC, V + (long)j * n, nullptr, nullptr, w, nullptr, 0.0, st));
        if (dinv) CHK(launch_mul_elem(w, h->arn_dinv, n, st));
        for (int i = 0; i <= j; ++i) {
            double hij = 0.0;
            CHK(launch_dot(V + (long)i * n, w, n, h->norm_scratch, slot, st));
            CHK(fetch(&hij));
            H[(long)i * maxiter + j] = hij;
            CHK(launch_axmy(w, V + (long)i * n, hij, n, st));
        }
        double beta = 0.0;
        CHK(launch_norm2(w, n, h->norm_scratch, slot, st));
        CHK(fetch(&beta));
        H[(long)(j + 1) * maxiter + j] = beta;
        if (beta < breakdown_tol) {
            *breakdown = 1;
            if (beta != 0.0) CHK(launch_divide(w, beta, n, st));
            ++j;
            break;
        }
        CHK(launch_divide(w, beta, n, st));
    }
    *steps = j;
    AMG_HIP(hipStreamSynchronize(st));
    return 0;
}

// v = V[:, :m] @ coef  (the restart vector, util/linalg.py:397-404); v may be NULL to keep it
// on the device only.  The result also becomes the next amg_arnoldi start if v0_from_device.
int amg_arnoldi_combine(amg_hier *h, const double *coef, int m, double *v)
{
    ENTER(h);
    if (!h->arn_V || m < 1 || m > h->arn_m || !coef) { set_error("bad combine arguments"); return AMG_EINVAL; }
    const long n = h->arn_n;
    double *tmp = nullptr;
    CHK(dev_alloc(&tmp, n, (long *)nullptr));
    AMG_HIP(hipMemcpyAsync(h->arn_coef, coef, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    int rc = launch_combine(tmp, h->arn_V, h->arn_coef, m, n, n, h->stream);
    if (rc == 0 && v) {
        hipMemcpyAsync(v, tmp, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream);
    }
    hipStreamSynchronize(h->stream);
    hipFree(tmp);
    return rc;
}

void amg_arnoldi_free(amg_hier *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->arn_V) hipFree(h->arn_V);
    if (h->arn_dinv) hipFree(h->arn_dinv);
    if (h->arn_coef) hipFree(h->arn_coef);
    h->arn_V = h->arn_dinv = h->arn_coef = nullptr;
    h->arn_m = 0; h->arn_n = 0;
    h->dev_bytes -= h->arn_bytes;
    h->arn_bytes = 0;
}

// average device time (ms, hipEvents on the hierarchy stream) of one application of a stored
// smoother to the level's resident x and b; which = AMG_PRE / AMG_POST / 2 (coarse smoother)
int amg_hier_time_relax(amg_hier *h, int lvl, int which, int reps, double *ms)
{
    ENTER(h);
    if (!h->finalized) { set_error("hierarchy not finalised"); return AMG_ESTATE; }
    if (lvl < 0 || lvl >= h->nlevels || which < 0 || which > 2 || reps < 1 || !ms) { set_error("bad arguments"); return AMG_EINVAL; }
    Level &L = h->lv[lvl];
    Smoother &s = (which == 2) ? h->coarse_sm : L.sm[which];
    CHK(relax(h, L, s, L.x, L.xalt, L.b, false));
    AMG_HIP(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < reps; ++r) CHK(relax(h, L, s, L.x, L.xalt, L.b, false));
    AMG_HIP(hipEventRecord(h->ev1, h->stream));
    AMG_HIP(hipStreamSynchronize(h->stream));
    float t = 0.f;
    AMG_HIP(hipEventElapsedTime(&t, h->ev0, h->ev1));
    *ms = t / reps;
    return 0;
}

void amg_set_stream_variant(int v) { amg::set_stream_variant(v); }
void amg_set_stream_pipe(int on) { amg::set_stream_pipe(on); }
void amg_set_xcd_chunk(int c) { amg::set_xcd_chunk(c); }
void amg_set_xcd_period(int on) { amg::set_xcd_period(on); }
void amg_set_stencil_form(int on) { amg::set_stencil_form(on); }
void amg_set_stencil_pairs(int on) { amg::set_stencil_pairs(on); }
void amg_set_sell_form(int on) { amg::set_sell_form(on); }
void amg_set_sell_index16(int on) { amg::set_sell_index16(on); }
void amg_set_gs_chain(int on) { amg::set_gs_chain(on); }
void amg_set_gs_level_hint(int on) { amg::set_gs_level_hint(on); }
void amg_set_gs_flow(int mode) { amg::set_gs_flow(mode); }
void amg_set_gs_flow_lookahead(int levels) { amg::set_gs_flow_lookahead(levels); }
int amg_gs_flow_status(void) { return amg::gs_flow_status(); }
void amg_set_bsr_spmv(int on) { amg::set_bsr_spmv(on); }
void amg_set_index16(int on) { amg::set_index16(on); }
void amg_set_tile_target(int t) { amg::set_tile_target(t); }
void amg_hier_use_graphs(amg_hier *h, int on) { if (h) { h->use_graphs = on; if (!on) drop_graphs(h); } }
void amg_hier_keep_residual(amg_hier *h, int on) { if (h) { h->keep_residual = on; h->r_kept = false; drop_graphs(h); } }

}  // extern "C"
