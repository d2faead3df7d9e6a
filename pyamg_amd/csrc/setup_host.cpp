// Host-side (CPU) setup helpers for the smoothed-aggregation hierarchy.
// The hierarchy is built ONCE on the CPU (north_star) and then shipped to HBM;
// these routines exist so that the build travels to the GPU box without the
// reference: they restate the setup kernels the BASELINE configurations need.
// All reference paths relative to /root/reference.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

// Greedy three-pass aggregation (Vanek/Mandel/Brezina), with the tie-breaking of
// pyamg/amg_core/smoothed_aggregation.h:122-222 so that the aggregates come out identical:
//   pass 1  a node none of whose neighbours is taken seeds an aggregate made of itself and all its
//           neighbours (nodes are visited in index order; a node without neighbours stays isolated);
//   pass 2  every node still free joins the aggregate of the first SEED-PASS member in its row;
//   pass 3  whatever is left seeds new aggregates from its still-free neighbours.
// agg[i] = aggregate of node i (-1: isolated), roots[k] = seed node of aggregate k.
int amgsetup_standard_aggregation(int n, const int *Ap, const int *Aj, int *agg, int *roots)
{
    enum : int { FREE = 0 };
    const int ISOLATED = -n;                 // cannot collide with -(aggregate id) since ids <= n-... < n
    // state encoding during the passes: 0 free, +k member since pass 1 or 3 (k = id+1),
    // -k attached in pass 2, ISOLATED no neighbours
    for (int i = 0; i < n; ++i) agg[i] = FREE;
    int count = 0;                           // aggregates so far

    for (int i = 0; i < n; ++i) {            // ---- pass 1
        if (agg[i] != FREE) continue;
        bool any_neighbour = false, neighbour_taken = false;
        for (int k = Ap[i]; k < Ap[i + 1] && !neighbour_taken; ++k) {
            const int j = Aj[k];
            if (j == i) continue;
            any_neighbour = true;
            neighbour_taken = (agg[j] != FREE);
        }
        if (!any_neighbour) { agg[i] = ISOLATED; continue; }
        if (neighbour_taken) continue;
        roots[count] = i;
        ++count;
        agg[i] = count;
        for (int k = Ap[i]; k < Ap[i + 1]; ++k) agg[Aj[k]] = count;
    }

    for (int i = 0; i < n; ++i) {            // ---- pass 2
        if (agg[i] != FREE) continue;
        for (int k = Ap[i]; k < Ap[i + 1]; ++k) {
            const int tag = agg[Aj[k]];
            if (tag > 0) { agg[i] = -tag; break; }      // only pass-1 members attract
        }
    }

    for (int i = 0; i < n; ++i) {            // ---- pass 3 + decoding to 0-based ids
        const int tag = agg[i];
        if (tag == ISOLATED) { agg[i] = -1; continue; }
        if (tag > 0) { agg[i] = tag - 1; continue; }
        if (tag < 0) { agg[i] = -tag - 1; continue; }
        roots[count] = i;
        agg[i] = count;
        for (int k = Ap[i]; k < Ap[i + 1]; ++k)
            if (agg[Aj[k]] == FREE) agg[Aj[k]] = count;   // later nodes only: earlier ones are decoded already
        ++count;
    }
    return count;
}

// pyamg/amg_core/relaxation.h:34-62 on the host: used only at SETUP time for the
// candidate improvement sweeps when no device is wanted for it.
void amgsetup_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                           int row_start, int row_stop, int row_step)
{
    for (int i = row_start; i != row_stop; i += row_step) {
        double rsum = 0, diag = 0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j) diag = Ax[jj];
            else rsum += Ax[jj] * x[j];
        }
        if (diag != 0.0) x[i] = (b[i] - rsum) / diag;
    }
}

// pyamg/amg_core/relaxation.h:756-810 on the host (candidate improvement on BSR operators at setup)
void amgsetup_block_gauss_seidel(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                                 const double *Dinv, int row_start, int row_stop, int row_step, int bs)
{
    const int B2 = bs * bs;
    std::vector<double> rsum((size_t)bs), v((size_t)bs);
    for (int i = row_start; i != row_stop; i += row_step) {
        std::fill(rsum.begin(), rsum.end(), 0.0);
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (i == j) continue;
            const double *blk = Ax + (int64_t)jj * B2;
            const double *xj = x + (int64_t)j * bs;
            for (int r = 0; r < bs; r++) {
                double sacc = 0.0;
                for (int c = 0; c < bs; c++) sacc += blk[r * bs + c] * xj[c];
                v[r] = sacc;
            }
            for (int k = 0; k < bs; k++) rsum[k] += v[k];
        }
        const int64_t ib = (int64_t)i * bs;
        for (int k = 0; k < bs; k++) rsum[k] = b[ib + k] - rsum[k];
        const double *D = Dinv + (int64_t)i * B2;
        for (int r = 0; r < bs; r++) {
            double sacc = 0.0;
            for (int c = 0; c < bs; c++) sacc += D[r * bs + c] * rsum[c];
            x[ib + r] = sacc;
        }
    }
}

// C = A * B for CSR operands, row-parallel restatement of scipy's SMMP
// (scipy.sparse._sparsetools csr_matmat): per output row the products are
// accumulated in the order of A's row entries, the output columns come out in
// reverse first-touch order and exact zeros are dropped -- so a chain of
// products rounds exactly like scipy's (and hence the reference's R*A*P).
// Pass 1: returns per-row upper bounds in Cp64 (as counts, length n_row+1 after
// prefix sum).  Pass 2 fills Cj/Cx and rewrites Cp64 with the compacted offsets.
// Per-row accumulator of the SMMP product: a small open-addressing table keyed by output column
// (sized from the row's upper bound) instead of scipy's dense arrays of length n_col -- at 10^8
// columns those cost 2 GB per thread to allocate and clear.  What fixes the bits is kept: every
// output entry accumulates its products in traversal order, and the entries come out in reverse
// first-touch order (scipy's linked list, walked from its head).
struct RowTable {
    std::vector<int> key;        // column, -1 = empty
    std::vector<double> sum;
    std::vector<int> order;      // slots in first-touch order
    unsigned mask = 0;
    void reserve_for(int64_t upper)
    {
        size_t want = 16;
        while ((int64_t)want < 2 * upper) want <<= 1;
        if (want > key.size()) { key.assign(want, -1); sum.assign(want, 0.0); }
        mask = (unsigned)(key.size() - 1);
        order.clear();
    }
    inline int slot_of(int k, bool &fresh)
    {
        unsigned h = ((unsigned)k * 2654435761u) & mask;
        for (;;) {
            const int c = key[h];
            if (c == k) { fresh = false; return (int)h; }
            if (c == -1) { key[h] = k; fresh = true; order.push_back((int)h); return (int)h; }
            h = (h + 1) & mask;
        }
    }
    void clear_touched()
    {
        for (int s : order) { key[(size_t)s] = -1; sum[(size_t)s] = 0.0; }
        order.clear();
    }
};

int64_t amgsetup_csr_matmat_count(int n_row, int n_col, const int64_t *Ap, const int *Aj,
                                  const int64_t *Bp, const int *Bj, int64_t *Cp)
{
    (void)n_col;
    Cp[0] = 0;
#pragma omp parallel
    {
        RowTable T;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n_row; i++) {
            int64_t upper = 0;
            for (int64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) upper += Bp[Aj[jj] + 1] - Bp[Aj[jj]];
            T.reserve_for(upper);
            for (int64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                int j = Aj[jj];
                for (int64_t kk = Bp[j]; kk < Bp[j + 1]; kk++) { bool fresh; T.slot_of(Bj[kk], fresh); }
            }
            Cp[i + 1] = (int64_t)T.order.size();
            T.clear_touched();
        }
    }
    for (int i = 0; i < n_row; i++) Cp[i + 1] += Cp[i];
    return Cp[n_row];
}

int64_t amgsetup_csr_matmat_fill(int n_row, int n_col, const int64_t *Ap, const int *Aj, const double *Ax,
                                 const int64_t *Bp, const int *Bj, const double *Bx, int64_t *Cp, int *Cj,
                                 double *Cx)
{
    (void)n_col;
    std::vector<int64_t> actual((size_t)n_row, 0);
#pragma omp parallel
    {
        RowTable T;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n_row; i++) {
            T.reserve_for(Cp[i + 1] - Cp[i]);          // the count pass sized the row exactly
            for (int64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                int j = Aj[jj];
                double v = Ax[jj];
                for (int64_t kk = Bp[j]; kk < Bp[j + 1]; kk++) {
                    bool fresh;
                    const int s = T.slot_of(Bj[kk], fresh);
                    T.sum[(size_t)s] += v * Bx[kk];
                }
            }
            int64_t nnz = Cp[i];
            for (size_t q = T.order.size(); q-- > 0;) {          // reverse first-touch order
                const int s = T.order[q];
                if (T.sum[(size_t)s] != 0) { Cj[nnz] = T.key[(size_t)s]; Cx[nnz] = T.sum[(size_t)s]; nnz++; }
            }
            actual[i] = nnz - Cp[i];
            T.clear_touched();
        }
    }
    // compact rows whose exact-zero results were dropped
    int64_t pos = 0;
    for (int i = 0; i < n_row; i++) {
        int64_t start = Cp[i], cnt = actual[i];
        if (start != pos) {
            std::memmove(Cj + pos, Cj + start, sizeof(int) * (size_t)cnt);
            std::memmove(Cx + pos, Cx + start, sizeof(double) * (size_t)cnt);
        }
        Cp[i] = pos;
        pos += cnt;
    }
    Cp[n_row] = pos;
    return pos;
}

// in-place stable sort of the column indices of every row (scipy sort_indices)
void amgsetup_csr_sort_indices(int n_row, const int64_t *Ap, int *Aj, double *Ax)
{
#pragma omp parallel
    {
        std::vector<std::pair<int, double>> tmp;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n_row; i++) {
            int64_t s = Ap[i], e = Ap[i + 1];
            bool sorted = true;
            for (int64_t k = s + 1; k < e; k++)
                if (Aj[k] < Aj[k - 1]) { sorted = false; break; }
            if (sorted) continue;
            tmp.resize((size_t)(e - s));
            for (int64_t k = s; k < e; k++) tmp[(size_t)(k - s)] = {Aj[k], Ax[k]};
            std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
            for (int64_t k = s; k < e; k++) { Aj[k] = tmp[(size_t)(k - s)].first; Ax[k] = tmp[(size_t)(k - s)].second; }
        }
    }
}

// B = A^T for CSR (scipy csr_tocsc order: entries of each output row in ascending source row)
}   // extern "C" (templates below)

// Row ranges of (nearly) equal entry counts for `parts` threads; part p owns rows [cut[p], cut[p+1])
static std::vector<int> split_by_entries(int n_row, const int64_t *Ap, int parts)
{
    std::vector<int> cut((size_t)parts + 1, n_row);
    cut[0] = 0;
    const int64_t nnz = Ap[n_row];
    for (int p = 1; p < parts; p++) {
        const int64_t want = nnz * p / parts;
        cut[(size_t)p] = (int)(std::lower_bound(Ap, Ap + n_row + 1, want) - Ap);
        if (cut[(size_t)p] > n_row) cut[(size_t)p] = n_row;
        if (cut[(size_t)p] < cut[(size_t)p - 1]) cut[(size_t)p] = cut[(size_t)p - 1];
    }
    return cut;
}

// Counting-sort transpose in parallel: thread p counts the columns of its row range, the per-column offsets are the
// prefix over (column, thread) -- so entries of one output row still come in ascending source row (threads own
// ascending row ranges), exactly csr_tocsc's order -- then every thread scatters its own rows.
// Each thread keeps counters only for the column WINDOW its rows touch (a prolongator's rows reach a narrow range of
// aggregates), allocated and zeroed by the thread itself: about n_col counters in total instead of threads * n_col.
template <class Scatter>
static void transpose_pattern(int n_row, int n_col, const int64_t *Ap, const int *Aj, int64_t *Bp, int *Bi, Scatter put)
{
    int parts = omp_get_max_threads();
    if (Ap[n_row] < 1000000) parts = 1;
    const std::vector<int> cut = split_by_entries(n_row, Ap, parts);
    std::vector<std::vector<int64_t>> win((size_t)parts);
    std::vector<int> wlo((size_t)parts, 0), whi((size_t)parts, 0);           // window [wlo, whi) of thread p
#pragma omp parallel for schedule(static, 1) num_threads(parts)
    for (int p = 0; p < parts; p++) {
        const int64_t k0 = Ap[cut[(size_t)p]], k1 = Ap[cut[(size_t)p + 1]];
        int lo = n_col, hi = -1;
        for (int64_t k = k0; k < k1; k++) { const int c = Aj[k]; lo = std::min(lo, c); hi = std::max(hi, c); }
        if (hi < lo) { lo = 0; hi = -1; }
        wlo[(size_t)p] = lo; whi[(size_t)p] = hi + 1;
        win[(size_t)p].assign((size_t)(hi + 1 - lo), 0);
        int64_t *c = win[(size_t)p].data() - lo;
        for (int64_t k = k0; k < k1; k++) c[Aj[k]]++;
    }
    // entries per output row, then the row pointer
    Bp[0] = 0;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < n_col; c++) {
        int64_t tot = 0;
        for (int p = 0; p < parts; p++)
            if (c >= wlo[(size_t)p] && c < whi[(size_t)p]) tot += win[(size_t)p][(size_t)(c - wlo[(size_t)p])];
        Bp[c + 1] = tot;
    }
    for (int c = 0; c < n_col; c++) Bp[c + 1] += Bp[c];
    // where thread p starts writing in output row c: after the threads with earlier rows
#pragma omp parallel for schedule(static)
    for (int c = 0; c < n_col; c++) {
        int64_t run = Bp[c];
        for (int p = 0; p < parts; p++)
            if (c >= wlo[(size_t)p] && c < whi[(size_t)p]) {
                int64_t &w = win[(size_t)p][(size_t)(c - wlo[(size_t)p])];
                const int64_t v = w; w = run; run += v;
            }
    }
#pragma omp parallel for schedule(static, 1) num_threads(parts)
    for (int p = 0; p < parts; p++) {
        int64_t *cur = win[(size_t)p].data() - wlo[(size_t)p];
        for (int i = cut[(size_t)p]; i < cut[(size_t)p + 1]; i++)
            for (int64_t k = Ap[i]; k < Ap[i + 1]; k++) {
                const int64_t d = cur[Aj[k]]++;
                Bi[d] = i;
                put(d, k);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Sequential sweeps (Gauss-Seidel family) run by several host threads with the SEQUENTIAL result.
// The rows are cut into chunks of `chunk` consecutive rows which the threads claim in sweep order; inside a
// chunk a thread runs the rows in sweep order, and before a row it waits until every row that comes EARLIER in
// the sweep and shares an entry with it (a_ij != 0 with j in an earlier chunk) has been finished by its thread
// (per-chunk progress counters, published every few rows).  On a structurally symmetric matrix those are all the
// conflicts there are: an earlier row that READS x_i is one that row i reads, so it is waited for before x_i is
// overwritten, and a later row that reads x_i waits for row i.  Every row therefore sees exactly the values the
// one-thread loop would give it.  No deadlock: chunks are claimed in sweep order, a thread only ever waits for
// chunks claimed before its own.
// With chunk = the matrix bandwidth, the chunks of a lexicographically numbered grid operator are its planes (rows
// of a 2-D grid): the thread on plane z+1 trails the thread on plane z by one publication interval.  On other
// numberings the sweep is still exact, only less concurrent.
// ---------------------------------------------------------------------------------------------------------
template <class RowOp>
static void pipelined_sweep(const int *Ap, const int *Aj, int n, bool reverse, int chunk, int nthreads, RowOp row_op)
{
    const int nchunks = (n + chunk - 1) / chunk;
    std::vector<std::atomic<int>> done((size_t)nchunks);
    for (auto &d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<int> next(0);
    const int publish = std::max(8, std::min(256, chunk / 64));
#pragma omp parallel num_threads(nthreads)
    {
        int seen_chunk[2] = {-1, -1}, seen_done[2] = {0, 0};          // what this thread last read of other chunks
        for (;;) {
            const int k = next.fetch_add(1, std::memory_order_relaxed);   // k-th chunk of the sweep
            if (k >= nchunks) break;
            const int c = reverse ? nchunks - 1 - k : k;
            const int lo = c * chunk, hi = std::min(n, lo + chunk);
            const int cnt = hi - lo;
            for (int p = 0; p < cnt; ++p) {
                const int i = reverse ? hi - 1 - p : lo + p;
                for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
                    const int j = Aj[jj];
                    if (reverse ? (j < hi) : (j >= lo)) continue;           // own chunk, or later in the sweep
                    const int cj = j / chunk;
                    const int cj_hi = std::min(n, (cj + 1) * chunk);
                    const int need = reverse ? cj_hi - j : j - cj * chunk + 1;   // rows of chunk cj that must be done
                    int slot = (seen_chunk[0] == cj) ? 0 : ((seen_chunk[1] == cj) ? 1 : -1);
                    if (slot < 0) { slot = (seen_chunk[0] == -1 || seen_done[0] >= chunk) ? 0 : 1; seen_chunk[slot] = cj; seen_done[slot] = 0; }
                    while (seen_done[slot] < need) {
                        seen_done[slot] = done[(size_t)cj].load(std::memory_order_acquire);
                        if (seen_done[slot] < need) {
#if defined(__x86_64__)
                            __builtin_ia32_pause();
#endif
                        }
                    }
                }
                row_op(i);
                if (((p + 1) % publish) == 0) done[(size_t)c].store(p + 1, std::memory_order_release);
            }
            done[(size_t)c].store(cnt, std::memory_order_release);
        }
    }
}

// every entry (i, j) has its mirror (j, i): what the pipelined sweeps rely on
static bool pattern_symmetric(const int *Ap, const int *Aj, int n)
{
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < n; ++i) {
        if (bad) continue;
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
            const int j = Aj[jj];
            if (j < 0 || j >= n) { bad = 1; break; }
            if (j == i) continue;
            bool found = false;
            for (int kk = Ap[j]; kk < Ap[j + 1]; ++kk)
                if (Aj[kk] == i) { found = true; break; }
            if (!found) { bad = 1; break; }
        }
    }
    return bad == 0;
}

static int bandwidth_of(const int *Ap, const int *Aj, int n)
{
    int bw = 0;
#pragma omp parallel for schedule(static) reduction(max : bw)
    for (int i = 0; i < n; ++i)
        for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) bw = std::max(bw, std::abs(Aj[jj] - i));
    return bw;
}

// Chunk length for pipelined_sweep: rows whose nearest earlier neighbour is far away (more than half the bandwidth) or
// absent start a "plane" of a lexicographically numbered mesh; if those rows are 0, S, 2S, ... the chunks are cut
// there -- the first rows of a chunk then do not depend on the last rows of the chunk before it (with chunks of one
// bandwidth that holds for the 7-point grid operator, whose bandwidth IS its plane, but not e.g. for a tetrahedral
// mesh, whose bandwidth is a plane plus a grid line: every chunk would wait for the whole chunk before it).
static int plane_stride(const int *Ap, const int *Aj, int n, int bw)
{
    std::vector<int> starts;
    const int T = std::max(1, omp_get_max_threads());
    std::vector<std::vector<int>> part((size_t)T);
#pragma omp parallel num_threads(T)
    {
        std::vector<int> &mine = part[(size_t)omp_get_thread_num()];
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) {
            int nearest = -1;
            for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) { const int j = Aj[jj]; if (j < i && j > nearest) nearest = j; }
            if (nearest < 0 || i - nearest > bw / 2) mine.push_back(i);
        }
    }
    for (auto &v : part) starts.insert(starts.end(), v.begin(), v.end());      // static schedule: already in row order
    if (starts.size() < 3 || starts[0] != 0) return bw;
    const int S = starts[1] - starts[0];
    if (S < bw / 2 || S > bw) return bw;
    for (size_t k = 1; k < starts.size(); ++k) if (starts[k] - starts[k - 1] != S) return bw;
    return S;
}

extern "C" {

// get_diagonal(A, inv=True) (util/utils.py:526-588) on flat CSR arrays, row-parallel: duplicates of the diagonal
// entry are summed in stored order as scipy's csr_diagonal does, a zero diagonal inverts to 0
void amgsetup_csr_diagonal_inv(int n, const int64_t *Ap, const int *Aj, const double *Ax, double *dinv)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double d = 0.0;
        for (int64_t k = Ap[i]; k < Ap[i + 1]; ++k)
            if (Aj[k] == i) d += Ax[k];
        dinv[i] = (d != 0.0) ? 1.0 / d : 0.0;
    }
}

// `iterations` sweeps ("forward" 0, "backward" 1, "symmetric" 2 = forward then backward) of the block Gauss-Seidel
// of relaxation.h:756-810 (bs = 1: Dinv holds the inverted diagonal) by several threads, bit-identical to the same
// calls of amgsetup_block_gauss_seidel.  Returns 1 when it ran, 0 when the operator does not qualify (small,
// narrow band or structurally unsymmetric): the caller then runs the one-thread loop.
int amgsetup_block_gauss_seidel_pipelined(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                                          const double *Dinv, int nb, int bs, int sweep, int iterations)
{
    const int T = std::min(omp_get_max_threads(), 32);
    if (T < 2 || nb < 200000) return 0;
    const int bw = bandwidth_of(Ap, Aj, nb);
    if (bw < 256 || (long)bw * 2 > nb) return 0;              // at least two chunks in flight, chunks worth a hand-off
    if (!pattern_symmetric(Ap, Aj, nb)) return 0;
    const int B2 = bs * bs;
    auto row_op = [&](int i) {
        double rsum[16], v[16];
        for (int k = 0; k < bs; ++k) rsum[k] = 0.0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            const int j = Aj[jj];
            if (i == j) continue;
            const double *blk = Ax + (int64_t)jj * B2;
            const double *xj = x + (int64_t)j * bs;
            for (int r = 0; r < bs; r++) {
                double sacc = 0.0;
                for (int c = 0; c < bs; c++) sacc += blk[r * bs + c] * xj[c];
                v[r] = sacc;
            }
            for (int k = 0; k < bs; k++) rsum[k] += v[k];
        }
        const int64_t ib = (int64_t)i * bs;
        for (int k = 0; k < bs; k++) rsum[k] = b[ib + k] - rsum[k];
        const double *D = Dinv + (int64_t)i * B2;
        for (int r = 0; r < bs; r++) {
            double sacc = 0.0;
            for (int c = 0; c < bs; c++) sacc += D[r * bs + c] * rsum[c];
            x[ib + r] = sacc;
        }
    };
    if (bs > 16) return 0;
    const int chunk = plane_stride(Ap, Aj, nb, bw);
    for (int it = 0; it < iterations; ++it) {
        if (sweep == 0 || sweep == 2) pipelined_sweep(Ap, Aj, nb, false, chunk, T, row_op);
        if (sweep == 1 || sweep == 2) pipelined_sweep(Ap, Aj, nb, true, chunk, T, row_op);
    }
    return 1;
}

// the same for the point sweep of relaxation.h:34-62 (amgsetup_gauss_seidel)
int amgsetup_gauss_seidel_pipelined(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                                    int n, int sweep, int iterations)
{
    const int T = std::min(omp_get_max_threads(), 32);
    if (T < 2 || n < 200000) return 0;
    const int bw = bandwidth_of(Ap, Aj, n);
    if (bw < 256 || (long)bw * 2 > n) return 0;
    if (!pattern_symmetric(Ap, Aj, n)) return 0;
    auto row_op = [&](int i) {
        double rsum = 0, diag = 0;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            const int j = Aj[jj];
            if (i == j) diag = Ax[jj];
            else rsum += Ax[jj] * x[j];
        }
        if (diag != 0.0) x[i] = (b[i] - rsum) / diag;
    };
    const int chunk = plane_stride(Ap, Aj, n, bw);
    for (int it = 0; it < iterations; ++it) {
        if (sweep == 0 || sweep == 2) pipelined_sweep(Ap, Aj, n, false, chunk, T, row_op);
        if (sweep == 1 || sweep == 2) pipelined_sweep(Ap, Aj, n, true, chunk, T, row_op);
    }
    return 1;
}

}   // extern "C"

extern "C" {

void amgsetup_csr_transpose(int n_row, int n_col, const int64_t *Ap, const int *Aj, const double *Ax,
                            int64_t *Bp, int *Bi, double *Bx)
{
    transpose_pattern(n_row, n_col, Ap, Aj, Bp, Bi, [&](int64_t d, int64_t k) { Bx[d] = Ax[k]; });
}

// pyamg/amg_core/smoothed_aggregation.h:323-500 for one candidate and scalar
// unknowns (K1 = K2 = 1): normalise the candidate over every aggregate.
// Ap/Ai: CSC of AggOp (members of aggregate j in ascending order).
void amgsetup_fit_candidates_scalar(int n_col, const int *Ap, const int *Ai, const double *B,
                                    double *Qx, double *R, double tol)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n_col; j++) {
        double norm_j = 0.0;
        for (int ii = Ap[j]; ii < Ap[j + 1]; ii++) {
            double v = B[Ai[ii]];
            Qx[ii] = v;
            norm_j += v * v;
        }
        norm_j = std::sqrt(norm_j);
        const double threshold_j = tol * norm_j;
        // (second norm evaluation of the reference gives the same value: nothing to orthogonalise)
        double scale;
        if (norm_j > threshold_j) { scale = 1.0 / norm_j; R[j] = norm_j; }
        else { scale = 0.0; R[j] = 0.0; }
        for (int ii = Ap[j]; ii < Ap[j + 1]; ii++) Qx[ii] *= scale;
    }
}

// d-D Poisson (d = 1..3) on an nx x ny x nz grid, 2d on the diagonal and -1 off it,
// lexicographic ordering with the LAST axis fastest and sorted column indices: the CSR the
// reference's gallery.poisson produces (pyamg/gallery/laplacian.py:14-69, stencil.py:12-138).
// Ap has n+1 entries (int64), Aj/Ax have nnz = sum over axes of 2*(n - n/len_axis) + n entries.
int64_t amgsetup_poisson_nnz(int nx, int ny, int nz)
{
    int64_t n = (int64_t)nx * ny * nz, nnz = n;
    if (nx > 1) nnz += 2 * (n - n / nx);
    if (ny > 1) nnz += 2 * (n - n / ny);
    if (nz > 1) nnz += 2 * (n - n / nz);
    return nnz;
}

void amgsetup_poisson(int nx, int ny, int nz, double diag, int64_t *Ap, int *Aj, double *Ax)
{
    const int64_t n = (int64_t)nx * ny * nz;
    const int64_t sy = nz, sx = (int64_t)ny * nz;
    // row lengths
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++) {
        int k = (int)(r % nz), j = (int)((r / nz) % ny), i = (int)(r / sx);
        int c = 1;
        if (nx > 1) c += (i > 0) + (i < nx - 1);
        if (ny > 1) c += (j > 0) + (j < ny - 1);
        if (nz > 1) c += (k > 0) + (k < nz - 1);
        Ap[r + 1] = c;
    }
    Ap[0] = 0;
    for (int64_t r = 0; r < n; r++) Ap[r + 1] += Ap[r];
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++) {
        int k = (int)(r % nz), j = (int)((r / nz) % ny), i = (int)(r / sx);
        int64_t p = Ap[r];
        if (nx > 1 && i > 0)      { Aj[p] = (int)(r - sx); Ax[p++] = -1.0; }
        if (ny > 1 && j > 0)      { Aj[p] = (int)(r - sy); Ax[p++] = -1.0; }
        if (nz > 1 && k > 0)      { Aj[p] = (int)(r - 1);  Ax[p++] = -1.0; }
        Aj[p] = (int)r; Ax[p++] = diag;
        if (nz > 1 && k < nz - 1) { Aj[p] = (int)(r + 1);  Ax[p++] = -1.0; }
        if (ny > 1 && j < ny - 1) { Aj[p] = (int)(r + sy); Ax[p++] = -1.0; }
        if (nx > 1 && i < nx - 1) { Aj[p] = (int)(r + sx); Ax[p++] = -1.0; }
    }
}

// Tentative prolongator for one candidate and scalar unknowns, built directly in the layout
// fit_candidates returns (Q.T.tobsr(): one stored entry per aggregated fine node, row i ->
// column agg[i]); agg[i] = -1 leaves row i empty.  The per-aggregate sums run over the members
// in ascending fine index, exactly the CSC order the reference walks
// (pyamg/aggregation/tentative.py:146-160, amg_core/smoothed_aggregation.h:323-500).
// Tp has n+1 entries.  Returns the number of stored entries.
int64_t amgsetup_tentative_scalar(int n, int n_agg, const int *agg, const double *B, double tol,
                                  int64_t *Tp, int *Tj, double *Tx, double *Bc)
{
    std::vector<double> norm2((size_t)n_agg, 0.0);
    for (int i = 0; i < n; i++)
        if (agg[i] >= 0) norm2[agg[i]] += B[i] * B[i];
    std::vector<double> scale((size_t)n_agg);
    for (int j = 0; j < n_agg; j++) {
        double nj = std::sqrt(norm2[j]);
        if (nj > tol * nj) { scale[j] = 1.0 / nj; Bc[j] = nj; }
        else { scale[j] = 0.0; Bc[j] = 0.0; }
    }
    int64_t nnz = 0;
    Tp[0] = 0;
    for (int i = 0; i < n; i++) {
        if (agg[i] >= 0) { Tj[nnz] = agg[i]; Tx[nnz] = B[i] * scale[agg[i]]; nnz++; }
        Tp[i + 1] = nnz;
    }
    return nnz;
}

// Jacobi-smoothed prolongator P = T - (w * D^-1 S) * T for a tentative T with at most one
// entry per row (pyamg/aggregation/smooth.py:163-205 with degree 1):
//   X = (D_inv_S * T) by SMMP in the order of S's row entries, with
//   D_inv_S_ik = (S_ik * dinv_i) * w   (scale_rows, then the scalar multiply), and
//   P = T - X entry by entry on the sorted union of the two patterns, zeros dropped
// (scipy's csr_matmat + csr_minus_csr).  Two calls: count (Pj == nullptr) then fill.
int64_t amgsetup_smooth_prolongator(int n, int n_agg, const int64_t *Sp, const int *Sj, const double *Sx,
                                    const double *dinv, double w, const int64_t *Tp, const int *Tj,
                                    const double *Tx, int64_t *Pp, int *Pj, double *Px)
{
    const bool fill = (Pj != nullptr);
    if (!fill) Pp[0] = 0;
#pragma omp parallel
    {
        std::vector<std::pair<int, double>> acc;
#pragma omp for schedule(dynamic, 8192)
        for (int i = 0; i < n; i++) {
            acc.clear();
            const double di = dinv[i];
            for (int64_t jj = Sp[i]; jj < Sp[i + 1]; jj++) {
                int k = Sj[jj];
                if (Tp[k + 1] == Tp[k]) continue;
                double v = (Sx[jj] * di) * w;
                int c = Tj[Tp[k]];
                double prod = v * Tx[Tp[k]];
                size_t q = 0;
                for (; q < acc.size(); q++)
                    if (acc[q].first == c) { acc[q].second += prod; break; }
                if (q == acc.size()) acc.push_back({c, prod});   // sums start from 0: 0 + prod == prod
            }
            // X drops exact zeros (csr_matmat) and is stored in reverse first-touch order; T - X then
            // runs through scipy's csr_binop_csr_general (X is not in canonical order), whose linked
            // list is touched by T's entry first, then X's entries as stored, and is emitted in
            // reverse touch order: X's columns in first-touch order (T's column excluded), then T's
            // column.  The stored order matters: it is the summation order of every later SpMV.
            const bool hasT = Tp[i + 1] > Tp[i];
            const int tc = hasT ? Tj[Tp[i]] : -1;
            const double tv = hasT ? Tx[Tp[i]] : 0.0;
            int64_t cnt = 0, base = fill ? Pp[i] : 0;
            auto emit = [&](int c, double v) {
                if (v != 0.0) {
                    if (fill) { Pj[base + cnt] = c; Px[base + cnt] = v; }
                    cnt++;
                }
            };
            double xt = 0.0;
            for (auto &e : acc) {
                if (e.second == 0.0) continue;
                if (hasT && e.first == tc) { xt = e.second; continue; }
                emit(e.first, 0.0 - e.second);
            }
            if (hasT) emit(tc, tv - xt);
            if (!fill) Pp[i + 1] = cnt;
        }
    }
    if (!fill)
        for (int i = 0; i < n; i++) Pp[i + 1] += Pp[i];
    return Pp[n];
}

// ---- block (BSR) setup: one candidate per node, square bs x bs blocks on the fine level ------------
// C = A * B for BSR operands with blocks R x N and N x C: row-parallel restatement of scipy's
// bsr_matmat (scipy.sparse._sparsetools bsr.h): per block row the output blocks are created in
// FORWARD first-touch order (unlike csr_matmat's reverse order), each accumulating A_ij * B_jk
// in traversal order with the dense product summed over the inner index last
// (C[r][c] += sum_n A[r][n] B[n][c], the running sum starting from C[r][c]); zero blocks stay.
// R == N == C == 1 is routed to csr_matmat by scipy -- callers use the CSR kernels above for it.
int64_t amgsetup_bsr_matmat_count(int n_brow, const int64_t *Ap, const int *Aj, const int64_t *Bp, const int *Bj, int64_t *Cp)
{
    Cp[0] = 0;
#pragma omp parallel
    {
        RowTable T;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n_brow; i++) {
            int64_t upper = 0;
            for (int64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) upper += Bp[Aj[jj] + 1] - Bp[Aj[jj]];
            T.reserve_for(upper);
            for (int64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                int j = Aj[jj];
                for (int64_t kk = Bp[j]; kk < Bp[j + 1]; kk++) { bool fresh; T.slot_of(Bj[kk], fresh); }
            }
            Cp[i + 1] = (int64_t)T.order.size();
            T.clear_touched();
        }
    }
    for (int i = 0; i < n_brow; i++) Cp[i + 1] += Cp[i];
    return Cp[n_brow];
}

void amgsetup_bsr_matmat_fill(int n_brow, int R, int N, int C, const int64_t *Ap, const int *Aj, const double *Ax,
                              const int64_t *Bp, const int *Bj, const double *Bx, const int64_t *Cp, int *Cj, double *Cx)
{
    const int64_t RC = (int64_t)R * C, RN = (int64_t)R * N, NC = (int64_t)N * C;
#pragma omp parallel
    {
        RowTable T;                                    // sum[] unused: the slot's rank is its block position
        std::vector<int> rank;
#pragma omp for schedule(dynamic, 4096)
        for (int i = 0; i < n_brow; i++) {
            const int64_t base = Cp[i], cnt = Cp[i + 1] - Cp[i];
            T.reserve_for(cnt);
            if (rank.size() < T.key.size()) rank.resize(T.key.size());
            std::fill(Cx + base * RC, Cx + (base + cnt) * RC, 0.0);
            for (int64_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                const int j = Aj[jj];
                const double *a = Ax + jj * RN;
                for (int64_t kk = Bp[j]; kk < Bp[j + 1]; kk++) {
                    bool fresh;
                    const int s = T.slot_of(Bj[kk], fresh);
                    if (fresh) { rank[(size_t)s] = (int)T.order.size() - 1; Cj[base + rank[(size_t)s]] = Bj[kk]; }
                    double *c = Cx + (base + rank[(size_t)s]) * RC;
                    const double *b = Bx + kk * NC;
                    for (int r = 0; r < R; r++)
                        for (int cc = 0; cc < C; cc++) {
                            double dot = c[(int64_t)r * C + cc];
                            for (int n = 0; n < N; n++) dot += a[(int64_t)r * N + n] * b[(int64_t)n * C + cc];
                            c[(int64_t)r * C + cc] = dot;
                        }
                }
            }
            T.clear_touched();
        }
    }
}

// B = A^T for BSR (scipy bsr_transpose: csr_tocsc on the block pattern -- blocks of an output row in
// ascending source row -- and every R x C block transposed to C x R)
void amgsetup_bsr_transpose(int n_brow, int n_bcol, int R, int C, const int64_t *Ap, const int *Aj, const double *Ax,
                            int64_t *Bp, int *Bi, double *Bx)
{
    const int64_t RC = (int64_t)R * C;
    transpose_pattern(n_brow, n_bcol, Ap, Aj, Bp, Bi, [&](int64_t d, int64_t k) {
        const double *a = Ax + k * RC;
        double *b = Bx + d * RC;
        for (int r = 0; r < R; r++)
            for (int c = 0; c < C; c++) b[(int64_t)c * R + r] = a[(int64_t)r * C + c];
    });
}

// Jacobi-smoothed prolongator P = T - (w D^-1 S) T for BSR S (bs x bs blocks) and a tentative T with ONE
// bs x K block per aggregated node (block row i -> block column agg[i], values Tx[i], row-major)
// (pyamg/aggregation/smooth.py:163-205, weighting 'diagonal', degree 1), with scipy's arithmetic and stored order:
//   D_inv_S = w * scale_rows(S, dinv): entry (r, c) of block (i, k) is (S[r][c] * dinv[i*bs + r]) * w;
//   X = D_inv_S * T (bsr_matmat): block (i, J) accumulates over S's row i in stored order,
//       X[r][c] += sum_n D_inv_S[r][n] T_k[n][c] with the running sum starting from X[r][c];
//   P = T - X through bsr_binop_bsr_general (X's indices are not sorted): the row's linked list is touched by
//   T's block first, then X's blocks as stored; blocks are emitted in REVERSE touch order and all-zero blocks
//   are dropped -- so X's columns in reverse first-touch order (T's column excluded), then T's column.
// *x_sorted (count pass: written; fill pass: read): X came out with sorted block columns in EVERY row, in which
// case scipy takes bsr_binop_bsr_canonical instead and P's blocks are the sorted merge of the two rows.
// Two calls: count (Pj == nullptr; fills Pp) then fill.  Returns the number of blocks.
int64_t amgsetup_smooth_prolongator_block(int n_nodes, int bs, int K, const int64_t *Sp, const int *Sj, const double *Sx, const double *dinv,
                                          double w, const int *agg, const double *Tx, int64_t *Pp, int *Pj, double *Px, int *x_sorted)
{
    const bool fill = (Pj != nullptr);
    const int64_t B2 = (int64_t)bs * bs;
    const size_t BK = (size_t)bs * (size_t)K;
    if (!fill) Pp[0] = 0;
    const bool canonical = fill && *x_sorted != 0;
    int all_sorted = 1;
#pragma omp parallel
    {
        std::vector<int> cols;
        std::vector<double> acc, out(BK);               // acc: bs*K per touched column, first-touch order
#pragma omp for schedule(dynamic, 2048)
        for (int i = 0; i < n_nodes; i++) {
            cols.clear(); acc.clear();
            for (int64_t jj = Sp[i]; jj < Sp[i + 1]; jj++) {
                const int k = Sj[jj];
                if (agg[k] < 0) continue;
                const int c = agg[k];
                size_t q = 0;
                for (; q < cols.size(); q++) if (cols[q] == c) break;
                if (q == cols.size()) { cols.push_back(c); acc.resize(acc.size() + BK, 0.0); }
                const double *blk = Sx + jj * B2;
                const double *t = Tx + (int64_t)k * (int64_t)BK;
                double *x = &acc[q * BK];
                for (int r = 0; r < bs; r++) {
                    const double di = dinv[(int64_t)i * bs + r];
                    for (int cc = 0; cc < K; cc++) {
                        double dot = x[(size_t)r * K + cc];
                        for (int nn = 0; nn < bs; nn++) dot += ((blk[(int64_t)r * bs + nn] * di) * w) * t[(size_t)nn * K + cc];
                        x[(size_t)r * K + cc] = dot;
                    }
                }
            }
            const int tc = agg[i];
            int64_t cnt = 0;
            const int64_t base = fill ? Pp[i] : 0;
            auto emit = [&](int c, const double *tv, const double *xv) {
                bool nz = false;
                for (size_t e = 0; e < BK; e++) { out[e] = (tv ? tv[e] : 0.0) - (xv ? xv[e] : 0.0); nz = nz || (out[e] != 0.0); }
                if (!nz) return;
                if (fill) { Pj[base + cnt] = c; for (size_t e = 0; e < BK; e++) Px[(size_t)(base + cnt) * BK + e] = out[e]; }
                cnt++;
            };
            const double *tmine = Tx + (int64_t)i * (int64_t)BK;
            if (!fill)
                for (size_t q = 1; q < cols.size(); q++)
                    if (cols[q] <= cols[q - 1]) {
#pragma omp atomic write
                        all_sorted = 0;
                    }
            if (canonical) {
                bool tdone = (tc < 0);                   // sorted merge of T's single block with X's (ascending) blocks
                for (size_t q = 0; q < cols.size(); q++) {
                    if (!tdone && tc < cols[q]) { emit(tc, tmine, nullptr); tdone = true; }
                    if (cols[q] == tc) { emit(tc, tmine, &acc[q * BK]); tdone = true; }
                    else emit(cols[q], nullptr, &acc[q * BK]);
                }
                if (!tdone) emit(tc, tmine, nullptr);
            } else {
                const double *xt = nullptr;
                for (size_t q = cols.size(); q-- > 0;) {
                    if (cols[q] == tc) { xt = &acc[q * BK]; continue; }
                    emit(cols[q], nullptr, &acc[q * BK]);
                }
                if (tc >= 0) emit(tc, tmine, xt);
            }
            if (!fill) Pp[i + 1] = cnt;
        }
    }
    if (!fill) {
        for (int i = 0; i < n_nodes; i++) Pp[i + 1] += Pp[i];
        *x_sorted = all_sorted;
    }
    return Pp[n_nodes];
}

// Stiffness matrix of -div(K grad u), P1 elements on the Kuhn triangulation of an m x m x m vertex grid
// (every cube cut into the six tetrahedra 000 -> e_a -> e_a + e_b -> 111, one per ordering (a, b, c) of the
// axes), assembled row by row for the (m-2)^3 interior vertices (lexicographic, last axis fastest; boundary
// vertices eliminated): the same operator pyamg_amd.gallery.p1_diffusion assembles from the element list
// (entries agree to rounding; the order of summation over the elements differs).  xyz: m^3 x 3 vertex
// coordinates, K: 3 x 3 row-major.  Two calls: Aj == nullptr counts (fills Ap, returns nnz), then fill.
int64_t amgsetup_kuhn_p1_diffusion(int m, const double *xyz, const double *K, int64_t *Ap, int *Aj, double *Ax)
{
    const int n1 = m - 2;
    const int64_t n = (int64_t)n1 * n1 * n1;
    const bool fill = (Aj != nullptr);
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    // offsets a row can couple to: differences of two corners on a common Kuhn path = all components >= 0 or all <= 0
    int noff = 0, off[15][3];
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                const bool nonneg = dx >= 0 && dy >= 0 && dz >= 0, nonpos = dx <= 0 && dy <= 0 && dz <= 0;
                if (nonneg || nonpos) { off[noff][0] = dx; off[noff][1] = dy; off[noff][2] = dz; noff++; }
            }                                            // lexicographic in (dx, dy, dz) = ascending column
    auto interior = [&](int i, int j, int k) { return i >= 1 && i <= n1 && j >= 1 && j <= n1 && k >= 1 && k <= n1; };
    if (!fill) {
        Ap[0] = 0;
#pragma omp parallel for schedule(static)
        for (int64_t row = 0; row < n; row++) {
            const int k = (int)(row % n1) + 1, j = (int)((row / n1) % n1) + 1, i = (int)(row / ((int64_t)n1 * n1)) + 1;
            int c = 0;
            for (int o = 0; o < noff; o++) c += interior(i + off[o][0], j + off[o][1], k + off[o][2]) ? 1 : 0;
            Ap[row + 1] = c;
        }
        for (int64_t row = 0; row < n; row++) Ap[row + 1] += Ap[row];
        return Ap[n];
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t row = 0; row < n; row++) {
        const int k = (int)(row % n1) + 1, j = (int)((row / n1) % n1) + 1, i = (int)(row / ((int64_t)n1 * n1)) + 1;
        double acc[3][3][3] = {{{0.0}}};                 // by (dx+1, dy+1, dz+1)
        const int v[3] = {i, j, k};
        // the eight cubes around the vertex: the vertex is corner c (components 0/1) of the cube at v - c
        for (int cx = 0; cx <= 1; cx++)
            for (int cy = 0; cy <= 1; cy++)
                for (int cz = 0; cz <= 1; cz++) {
                    const int c[3] = {cx, cy, cz};
                    const int o[3] = {v[0] - cx, v[1] - cy, v[2] - cz};
                    if (o[0] < 0 || o[1] < 0 || o[2] < 0 || o[0] > m - 2 || o[1] > m - 2 || o[2] > m - 2) continue;
                    for (int p = 0; p < 6; p++) {
                        int path[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}};
                        path[1][perms[p][0]] = 1;
                        path[2][perms[p][0]] = 1; path[2][perms[p][1]] = 1;
                        int me = -1;
                        for (int q = 0; q < 4; q++)
                            if (path[q][0] == c[0] && path[q][1] == c[1] && path[q][2] == c[2]) me = q;
                        if (me < 0) continue;
                        double X[4][3];
                        for (int q = 0; q < 4; q++) {
                            const int64_t id = ((int64_t)(o[0] + path[q][0]) * m + (o[1] + path[q][1])) * m + (o[2] + path[q][2]);
                            for (int d = 0; d < 3; d++) X[q][d] = xyz[id * 3 + d];
                        }
                        double E[3][3];
                        for (int q = 0; q < 3; q++)
                            for (int d = 0; d < 3; d++) E[q][d] = X[q + 1][d] - X[0][d];
                        const double det = E[0][0] * (E[1][1] * E[2][2] - E[1][2] * E[2][1]) - E[0][1] * (E[1][0] * E[2][2] - E[1][2] * E[2][0]) +
                                           E[0][2] * (E[1][0] * E[2][1] - E[1][1] * E[2][0]);
                        // inverse of E by cofactors; gradient of lambda_q (q = 1..3) = column q-1 of E^-1
                        double Ei[3][3];
                        Ei[0][0] = (E[1][1] * E[2][2] - E[1][2] * E[2][1]) / det; Ei[0][1] = (E[0][2] * E[2][1] - E[0][1] * E[2][2]) / det; Ei[0][2] = (E[0][1] * E[1][2] - E[0][2] * E[1][1]) / det;
                        Ei[1][0] = (E[1][2] * E[2][0] - E[1][0] * E[2][2]) / det; Ei[1][1] = (E[0][0] * E[2][2] - E[0][2] * E[2][0]) / det; Ei[1][2] = (E[0][2] * E[1][0] - E[0][0] * E[1][2]) / det;
                        Ei[2][0] = (E[1][0] * E[2][1] - E[1][1] * E[2][0]) / det; Ei[2][1] = (E[0][1] * E[2][0] - E[0][0] * E[2][1]) / det; Ei[2][2] = (E[0][0] * E[1][1] - E[0][1] * E[1][0]) / det;
                        double G[4][3];
                        for (int q = 1; q < 4; q++)
                            for (int d = 0; d < 3; d++) G[q][d] = Ei[d][q - 1];
                        for (int d = 0; d < 3; d++) G[0][d] = -(G[1][d] + G[2][d] + G[3][d]);
                        const double vol = std::fabs(det) / 6.0;
                        double KG[3];
                        for (int d = 0; d < 3; d++) KG[d] = K[d * 3 + 0] * G[me][0] + K[d * 3 + 1] * G[me][1] + K[d * 3 + 2] * G[me][2];
                        for (int q = 0; q < 4; q++) {
                            const double e = (KG[0] * G[q][0] + KG[1] * G[q][1] + KG[2] * G[q][2]) * vol;
                            acc[path[q][0] - c[0] + 1][path[q][1] - c[1] + 1][path[q][2] - c[2] + 1] += e;
                        }
                    }
                }
        int64_t pos = Ap[row];
        for (int o = 0; o < noff; o++) {
            const int ii = i + off[o][0], jj = j + off[o][1], kk = k + off[o][2];
            if (!interior(ii, jj, kk)) continue;
            Aj[pos] = (int)((((int64_t)(ii - 1)) * n1 + (jj - 1)) * n1 + (kk - 1));
            Ax[pos] = acc[off[o][0] + 1][off[o][1] + 1][off[o][2] + 1];
            pos++;
        }
    }
    return Ap[n];
}

// ---- classical (Ruge-Stuben) setup, pyamg/amg_core/ruge_stuben.h --------------------------------
// :46-99 classical strength: keep off-diagonals with |a_ij| >= theta * max_k!=i |a_ik|, and the diagonal
int amgsetup_classical_strength(int n_row, double theta, const int *Ap, const int *Aj, const double *Ax,
                                int *Sp, int *Sj, double *Sx)
{
    int nnz = 0;
    Sp[0] = 0;
    for (int i = 0; i < n_row; i++) {
        double max_offdiagonal = 2.2250738585072014e-308;   // std::numeric_limits<double>::min()
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++)
            if (Aj[jj] != i) max_offdiagonal = std::max(max_offdiagonal, std::fabs(Ax[jj]));
        double threshold = theta * max_offdiagonal;
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            double norm_jj = std::fabs(Ax[jj]);
            if (norm_jj >= threshold && Aj[jj] != i) { Sj[nnz] = Aj[jj]; Sx[nnz] = Ax[jj]; nnz++; }
            if (Aj[jj] == i) { Sj[nnz] = Aj[jj]; Sx[nnz] = Ax[jj]; nnz++; }
        }
        Sp[i + 1] = nnz;
    }
    return nnz;
}

// Ruge-Stuben first-pass C/F splitting (ruge_stuben.h:158-310).  Nodes are kept in buckets by
// their weight lambda (= number of points they would interpolate to, plus one per neighbour already
// made F); the heaviest node becomes C, the points that depend on it become F, and weights are
// adjusted.  The bucket bookkeeping below (one array sorted by weight, a position per node, and the
// [first, size) of every weight class) reproduces the reference's choice among equal weights: a
// node whose weight rises moves to the END of its class before the class boundary moves over it, a
// node whose weight drops moves to the FRONT of its class.  F = 0, C = 1.
namespace {
struct WeightBuckets {
    std::vector<int> weight, first, size, order, where;
    explicit WeightBuckets(int n) : weight((size_t)n, 0), first((size_t)n + 1, 0), size((size_t)n + 1, 0),
                                    order((size_t)n), where((size_t)n) {}
    void build()
    {
        const int n = (int)weight.size();
        for (int i = 0; i < n; ++i) size[weight[i]]++;
        for (int w = 0, run = 0; w < n; ++w) { first[w] = run; run += size[w]; size[w] = 0; }
        for (int i = 0; i < n; ++i) {
            const int w = weight[i], slot = first[w] + size[w]++;
            order[slot] = i;
            where[i] = slot;
        }
    }
    void swap_slots(int a, int b)
    {
        where[order[a]] = b;
        where[order[b]] = a;
        std::swap(order[a], order[b]);
    }
    void raise(int k)       // weight[k] += 1
    {
        const int w = weight[k], last = first[w] + size[w] - 1;
        swap_slots(where[k], last);
        size[w] -= 1;
        size[w + 1] += 1;
        first[w + 1] = last;
        weight[k] = w + 1;
    }
    void lower(int k)       // weight[k] -= 1
    {
        const int w = weight[k], front = first[w];
        swap_slots(where[k], front);
        size[w] -= 1;
        size[w - 1] += 1;
        first[w] += 1;
        first[w - 1] = first[w] - size[w - 1];
        weight[k] = w - 1;
    }
};
}  // namespace

void amgsetup_rs_cf_splitting(int n, const int *Sp, const int *Sj, const int *Tp, const int *Tj, int *splitting)
{
    const int F = 0, Cpt = 1, UNDECIDED = 2;
    WeightBuckets wb(n);
    for (int i = 0; i < n; ++i) wb.weight[i] = Tp[i + 1] - Tp[i];
    wb.build();
    for (int i = 0; i < n; ++i) {
        const int w = Tp[i + 1] - Tp[i];
        // nothing depends on i (or only i itself): it can only be a fine point
        splitting[i] = (w == 0 || (w == 1 && Tj[Tp[i]] == i)) ? F : UNDECIDED;
    }
    for (int slot = n - 1; slot >= 0; --slot) {
        const int i = wb.order[slot];
        wb.size[wb.weight[i]] -= 1;                       // i leaves the structure
        if (splitting[i] == F) continue;
        splitting[i] = Cpt;
        for (int jj = Tp[i]; jj < Tp[i + 1]; ++jj) {      // points that depend on i become fine ...
            const int j = Tj[jj];
            if (splitting[j] != UNDECIDED) continue;
            splitting[j] = F;
            for (int kk = Sp[j]; kk < Sp[j + 1]; ++kk) {  // ... which makes their other influences more attractive
                const int k = Sj[kk];
                if (splitting[k] == UNDECIDED && wb.weight[k] < n - 1) wb.raise(k);
            }
        }
        for (int jj = Sp[i]; jj < Sp[i + 1]; ++jj) {      // i no longer needs the points it depends on
            const int j = Sj[jj];
            if (splitting[j] == UNDECIDED && wb.weight[j] > 0) wb.lower(j);
        }
    }
}

// :497-600 direct interpolation.  pass 1: row pointer of P; pass 2: entries
int amgsetup_rs_direct_interpolation_pass1(int n_nodes, const int *Sp, const int *Sj, const int *splitting, int *Bp)
{
    int nnz = 0;
    Bp[0] = 0;
    for (int i = 0; i < n_nodes; i++) {
        if (splitting[i] == 1) {
            nnz++;
        } else {
            for (int jj = Sp[i]; jj < Sp[i + 1]; jj++)
                if (splitting[Sj[jj]] == 1 && Sj[jj] != i) nnz++;
        }
        Bp[i + 1] = nnz;
    }
    return nnz;
}

void amgsetup_rs_direct_interpolation_pass2(int n_nodes, const int *Ap, const int *Aj, const double *Ax,
                                            const int *Sp, const int *Sj, const double *Sx,
                                            const int *splitting, const int *Bp, int *Bj, double *Bx)
{
    for (int i = 0; i < n_nodes; i++) {
        if (splitting[i] == 1) {
            Bj[Bp[i]] = i;
            Bx[Bp[i]] = 1;
        } else {
            double sum_strong_pos = 0, sum_strong_neg = 0;
            for (int jj = Sp[i]; jj < Sp[i + 1]; jj++)
                if (splitting[Sj[jj]] == 1 && Sj[jj] != i) {
                    if (Sx[jj] < 0) sum_strong_neg += Sx[jj];
                    else sum_strong_pos += Sx[jj];
                }
            double sum_all_pos = 0, sum_all_neg = 0, diag = 0;
            for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                if (Aj[jj] == i) diag += Ax[jj];
                else if (Ax[jj] < 0) sum_all_neg += Ax[jj];
                else sum_all_pos += Ax[jj];
            }
            double alpha = sum_all_neg / sum_strong_neg;
            double beta = sum_all_pos / sum_strong_pos;
            if (sum_strong_pos == 0) { diag += sum_all_pos; beta = 0; }
            double neg_coeff = -alpha / diag;
            double pos_coeff = -beta / diag;
            int nnz = Bp[i];
            for (int jj = Sp[i]; jj < Sp[i + 1]; jj++)
                if (splitting[Sj[jj]] == 1 && Sj[jj] != i) {
                    Bj[nnz] = Sj[jj];
                    Bx[nnz] = (Sx[jj] < 0) ? neg_coeff * Sx[jj] : pos_coeff * Sx[jj];
                    nnz++;
                }
        }
    }
    std::vector<int> map((size_t)n_nodes);
    for (int i = 0, sum = 0; i < n_nodes; i++) { map[i] = sum; sum += splitting[i]; }
    for (int i = 0; i < Bp[n_nodes]; i++) Bj[i] = map[Bj[i]];
}

// Greedy (first-fit) vertex colouring in natural order: colour[i] = smallest colour not used by
// an already coloured neighbour.  Rows of one colour are mutually independent, so a Gauss-Seidel
// sweep ordered colour by colour (an index list for gauss_seidel_indexed,
// pyamg/relaxation/relaxation.py:671-741) has as many dependency levels as colours.
// Returns the number of colours.
int amgsetup_greedy_coloring(int n, const int *Ap, const int *Aj, int *colour)
{
    std::vector<int> mark;
    int ncol = 0;
    for (int i = 0; i < n; i++) colour[i] = -1;
    for (int i = 0; i < n; i++) {
        if ((int)mark.size() < ncol + 1) mark.resize((size_t)ncol + 1, -1);
        for (int jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            int j = Aj[jj];
            if (j != i && j >= 0 && j < n && colour[j] >= 0) mark[colour[j]] = i;
        }
        int c = 0;
        while (c < ncol && mark[c] == i) c++;
        colour[i] = c;
        if (c == ncol) { ncol++; mark.resize((size_t)ncol + 1, -1); }
    }
    return ncol;
}

// 1 when every stored entry (i, j) has its mirror (j, i) stored too (checked on all host threads)
int amgsetup_pattern_symmetric(int n, const int *Ap, const int *Aj) { return pattern_symmetric(Ap, Aj, n) ? 1 : 0; }

void amgsetup_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int amgsetup_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}


// Dense diagonal block of A for every subdomain (sorted index lists Sj[Sp[d]..Sp[d+1])), row-major into
// Tx[Tp[d]..]: the input of the Schwarz smoother's block inversion (pyamg/amg_core/relaxation.h:836-899).
// Rows of A and subdomains are both sorted, so a merge per row finds the common columns.
void amgsetup_extract_subblocks(const int *Ap, const int *Aj, const double *Ax, double *Tx, const int *Tp,
                                const int *Sj, const int *Sp, int nsdomains, int nrows)
{
    (void)nrows;
    const long total = nsdomains > 0 ? Tp[nsdomains] : 0;
#pragma omp parallel for schedule(static)
    for (long k = 0; k < total; ++k) Tx[k] = 0.0;
#pragma omp parallel for schedule(dynamic, 256)
    for (int d = 0; d < nsdomains; ++d) {
        const int m = Sp[d + 1] - Sp[d];
        const int *S = Sj + Sp[d];
        for (int li = 0; li < m; ++li) {
            const int row = S[li];
            double *Trow = Tx + Tp[d] + (long)li * m;
            int lc = 0;
            for (int k = Ap[row]; k < Ap[row + 1] && lc < m; ++k) {
                const int col = Aj[k];
                while (lc < m && S[lc] < col) ++lc;
                if (lc < m && S[lc] == col) { Trow[lc] = Ax[k]; ++lc; }
            }
        }
    }
}

}  // extern "C"
