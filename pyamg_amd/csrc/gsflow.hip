// Dataflow Gauss-Seidel: the exact sequential sweep of pyamg/amg_core/relaxation.h:34-62 (gauss_seidel) and :90-173
// (bsr_gauss_seidel with 1x1 blocks) as ONE persistent launch per sequence of directional sweeps.
//
// Dependency-level scheduling (hier.hip build_levels) reproduces the sequential sweep bit for bit, but every wide
// level used to be a launch of its own: ~4.5-5.4 us start to start for a few microseconds' worth of bytes, so the
// reference's default 3-D setup (smoothed aggregation + symmetric Gauss-Seidel) was launch-bound (DESIGN.md
// section 4).  Here nothing waits for a LEVEL any more:
//   * unknowns are renumbered in level order (rows of a level sorted by length), every level is cut into chunks of
//     64 rows -- one wave, one lane per row, the row's off-diagonal entries slot-major as in the sliced form
//     (sell.hip), padded slots = value 0 times a permanent 0.0;
//   * chunks are dealt to the resident waves round-robin IN SWEEP ORDER (static: task Q runs on wave Q mod NW); a wave
//     loads its chunk's entries, diagonal and right-hand side (none of which depend on the sweep), then gathers
//     its operands;
//   * a directional sweep reads the values of the previous sweep from buffer X[s] and publishes its own into
//     X[s+1], which the gather kernel has filled with a signalling-NaN SENTINEL: an operand is ready when it no longer
//     reads as the sentinel.  Every value is written exactly once per launch by ONE relaxed agent-scope 8-byte
//     atomic store and read by relaxed agent-scope atomic loads: the datum is its own flag, so no ordering between
//     different addresses is needed, and there are no write-after-read hazards because no buffer is overwritten;
//     results of IEEE divisions are never signalling NaNs, so a finished value cannot be mistaken for the sentinel
//     (the gather quiets a user-supplied x entry that happens to carry the sentinel's bits: every arithmetic use of
//     it would have quieted it the same way);
//   * forward and backward sweeps (and further iterations) follow each other inside the launch without any
//     barrier: the first chunks of the backward sweep simply wait for the forward values they read.
// Each lane adds ITS row's products in stored order with separately rounded multiply and add and divides once:
// bit-identical to the level-scheduled kernels, the chained sweeps and the oracle.
//
// Progress: every wave handles its tasks in increasing order and a task only waits for tasks that precede it in
// that order, so the earliest unfinished task can always run -- provided all waves are resident, which the launch
// guarantees by sizing the grid from the occupancy query with a margin (MI355X_MICROARCH.md, residency).  Every
// spin is bounded by a wall-clock budget measured from the wave's start; a wave that runs out of it raises the
// status flag and leaves, and so does everybody who waits for it (gs_flow_status()).
#include "hier.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace amg {

namespace {

// host-side loops of the form builders over up to 16 threads: fn(first, last) on contiguous ranges of [0, n)
template <class F> void flow_parallel(long n, long grain, F fn)
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

constexpr unsigned long long FLOW_SENT = 0x7FF4A5A55A5A0001ULL;    // a signalling NaN no arithmetic result can equal
constexpr unsigned long long FLOW_QUIET = 0x0008000000000000ULL;
#ifndef AMG_FLOW_SEG
#define AMG_FLOW_SEG 8
#endif
#ifndef AMG_FLOW_SLEEP
#define AMG_FLOW_SLEEP 1
#endif
constexpr int FLOW_SEG = AMG_FLOW_SEG;          // slots a lane owns at most: a row of up to FLOW_SEG * LPR off-diagonal entries is shared by LPR lanes

#define FCHK(call)                 \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != 0) return rc__; \
    } while (0)

template <class T> int falloc(T **p, long count, long *acct)
{
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, sizeof(T) * (size_t)count);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (dataflow Gauss-Seidel form)", __FILE__, __LINE__);
    if (acct) *acct += (long)(sizeof(T) * (size_t)count);
    return 0;
}

struct FlowArgs {
    unsigned long long *X;      // (nseq + 1) buffers of xstride entries
    double *x_out;              // the caller's x (original numbering): written by the last sweep
    int *status;
    long xstride;
    long long budget;
    int nchunks, nseq, n;
    unsigned dirmask;           // bit s: sweep s runs backward
    int xcd;                    // consecutive tasks on one XCD (flow_slot)
};

// Which task slot a workgroup takes.  Workgroups are dealt to the 8 XCDs round-robin (b and b + 8 share one: observed, not
// promised -- speed only), so with xcd != 0 the slots are handed out such that CONSECUTIVE tasks -- neighbouring chunks of a
// level, whose gathered operands share cache lines -- run on one XCD and find them in its L2.
__device__ __forceinline__ int flow_slot(int b, int nw, int xcd)
{
    if (!xcd || nw < 16) return b;
    const int k = b & 7, j = b >> 3;                       // workgroup b = 8 j + k is the j-th of XCD k
    if (xcd >= 2) {
        // groups of xcd consecutive tasks, dealt to the XCDs round-robin: every dependency level is spread over all eight
        // (each XCD has an eighth of the path to memory), neighbours within a group share their operands' lines
        if (nw % (8 * xcd)) return b;
        return ((j / xcd) * 8 + k) * xcd + j % xcd;
    }
    const int per = nw >> 3, rem = nw & 7;                 // xcd == 1: XCD k holds per (+1 if k < rem) consecutive slots
    return k * per + min(k, rem) + j;
}

__device__ __forceinline__ unsigned long long ald(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// First read of an operand through the caches (L2 is per XCD and not coherent with the others'): an X slot goes from the
// sentinel to its value exactly once per application, so a stale line can only show the sentinel, and a sentinel is polled
// again with ald().  Operands that were final before this XCD first touched their line are L2 hits instead of one
// fabric transaction per gather.
#ifndef AMG_FLOW_PLAIN
#define AMG_FLOW_PLAIN 1
#endif
__device__ __forceinline__ unsigned long long pld(const unsigned long long *p)
{
    if (AMG_FLOW_PLAIN) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the workgroup IS one wave: LDS operations of a wave execute in order, so all that is needed between the lanes'
// product writes and the leaders' reads is that the compiler keeps them in order
__device__ __forceinline__ void flow_wave_sync()
{
#ifdef AMG_FLOW_SYNCTHREADS
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// x (original numbering) -> X[0] (level order), b -> bp; X[1..nseq] = sentinel; entry n of every buffer = 0.0
// (columns n .. ncols-1 are a partitioned level's HALO: operands no local row writes, frozen during the sweep -- they sit
//  behind the owned unknowns in every buffer; entry ncols of every buffer is the permanent 0.0)
__global__ __launch_bounds__(256) void flow_gather_kernel(const int *rowmap, const double *x, const double *b, unsigned long long *X,
                                                           double *bp, long xstride, int n, int ncols, int nseq)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > ncols) return;
    if (k >= n) {
        unsigned long long hb = 0ULL;
        if (k < ncols) {
            hb = (unsigned long long)__double_as_longlong(x[k]);
            if (hb == FLOW_SENT) hb |= FLOW_QUIET;
        }
        for (int s = 0; s <= nseq; ++s) X[(long)s * xstride + k] = hb;
        return;
    }
    const int i = rowmap[k];
    unsigned long long xb = (unsigned long long)__double_as_longlong(x[i]);
    if (xb == FLOW_SENT) xb |= FLOW_QUIET;
    X[k] = xb;
    bp[k] = b[i];
    for (int s = 1; s <= nseq; ++s) X[(long)s * xstride + k] = FLOW_SENT;
}

// The static arrays travel as __restrict__ parameters of their own: the chunk descriptor is then read through the
// scalar cache (one s_load per task) and nothing static is re-read after the stores.
// LPR lanes share a row (64 / LPR rows per wave): lane g of a row's group owns the row's slots [g * seg, (g + 1) * seg)
// -- its entries, its operand gathers, its products -- and the group's first lane adds the products of all groups in
// stored order (through LDS when LPR > 1).  A long row then costs SEG wave-level gathers instead of its length, and
// a level of few long rows spreads over more compute units.
template <int SEG, int LPR, bool BSR1>
__global__ __launch_bounds__(64) void gs_flow_kernel(const FlowChunk *__restrict__ meta, const int *__restrict__ col,
                                                      const double *__restrict__ val, const double *__restrict__ diag,
                                                      const double *__restrict__ bp, const int *__restrict__ rowmap,
                                                      const int *__restrict__ gate_f, const int *__restrict__ gate_b, FlowArgs a)
{
    __shared__ double prod[LPR > 1 ? SEG * 64 : 1];
    const int lane = threadIdx.x;
    const int t = lane / LPR;
    const int NW = (int)gridDim.x;
    const long long t0 = wall_clock64();
    int s = 0, q = flow_slot((int)blockIdx.x, NW, a.xcd);
    while (q >= a.nchunks) { q -= a.nchunks; ++s; }
    auto chunk_of = [&](int ss, int qq) { return (((a.dirmask >> ss) & 1u) != 0) ? a.nchunks - 1 - qq : qq; };
    FlowChunk mnext = meta[s < a.nseq ? chunk_of(s, q) : 0];
    while (s < a.nseq) {
        const bool rev = ((a.dirmask >> s) & 1u) != 0;
        const FlowChunk m = mnext;
        int s2 = s, q2 = q + NW;
        while (q2 >= a.nchunks) { q2 -= a.nchunks; ++s2; }
        mnext = meta[s2 < a.nseq ? chunk_of(s2, q2) : 0];      // the next task's descriptor: a scalar load, not needed before then
        const int seg = m.nslots;                       // slots per lane in this chunk (<= SEG)
        const bool leader = (lane % LPR) == 0 && t < m.nrows;
        const int k = m.row0 + (t < m.nrows ? t : 0);
        // what does not depend on the sweep: entries, right-hand side, diagonal.  Straight-line code: always SEG slots per
        // lane; a slot past the chunk's own `seg` re-reads the last real slot (same cache lines) and is turned into the
        // padding pair (0.0, the permanent zero) -- uniform branches per slot cost more than the redundant requests
        int idx[SEG];
        double v[SEG];
        const long e0 = (long)m.off * 64 + lane;
        const int smax = seg > 0 ? seg - 1 : 0;
#pragma unroll
        for (int u = 0; u < SEG; ++u) {
            const long e = e0 + (long)min(u, smax) * 64;
            idx[u] = col[e];
            v[u] = val[e];
        }
        double bb = 0.0, dd = 1.0;
        int orow = 0, gate = a.n;
        const bool last = s == a.nseq - 1;
        if (leader) {
            bb = bp[k]; dd = diag[k];
            gate = rev ? gate_b[k] : gate_f[k];
            if (last) orow = rowmap[k];
        }
        const unsigned long long *Xo = a.X + (long)s * a.xstride;
        unsigned long long *Xn = a.X + (long)(s + 1) * a.xstride;
        // Gate: the row's latest operand of this sweep that is produced at least TWO levels earlier.  Until it exists the
        // wave is far ahead of the sweep and polls this one word per row instead of all its operands; then the operands:
        // produced by THIS sweep (earlier levels in sweep order) from the new buffer, everything else from the old one.
        // (Requesting gate and operands together -- one round trip less when everything is ready -- measured SLOWER:
        //  1.19 -> 1.46 us per level on the 7-point level, and level on the wide block levels.)
        unsigned long long gv = ald(Xn + gate);
        for (unsigned spin = 0; __any(gv == FLOW_SENT); ++spin) {
            __builtin_amdgcn_s_sleep(4);
            if ((spin & 31u) == 31u && wall_clock64() - t0 > a.budget) {
                if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (gv == FLOW_SENT) gv = ald(Xn + gate);
        }
        const int xs = (int)a.xstride;
        unsigned long long xb[SEG];
#pragma unroll
        for (int u = 0; u < SEG; ++u) {
            const bool real = u < seg && (seg > 0);
            v[u] = real ? v[u] : 0.0;
            idx[u] = real ? idx[u] : a.n;
            const bool fresh = rev ? (idx[u] >= m.lvl_hi) : (idx[u] < m.lvl_lo);
            idx[u] += fresh ? xs : 0;
            xb[u] = pld(Xo + idx[u]);
        }
        for (unsigned spin = 0;; ++spin) {
            bool bad = false;
#pragma unroll
            for (int u = 0; u < SEG; ++u) bad |= (xb[u] == FLOW_SENT);
            if (!__any(bad)) break;
            if (AMG_FLOW_SLEEP > 0) __builtin_amdgcn_s_sleep(AMG_FLOW_SLEEP);
            if ((spin & 31u) == 31u && wall_clock64() - t0 > a.budget) {
                if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
#pragma unroll
            for (int u = 0; u < SEG; ++u)
                if (xb[u] == FLOW_SENT) xb[u] = ald(Xo + idx[u]);
        }
        double acc = BSR1 ? bb : 0.0;
        if constexpr (LPR == 1) {
#pragma unroll
            for (int u = 0; u < SEG; ++u) {
                const double pr = v[u] * __longlong_as_double((long long)xb[u]);
                acc = BSR1 ? (acc - pr) : (acc + pr);
            }
        } else {
#pragma unroll
            for (int u = 0; u < SEG; ++u) prod[u * 64 + lane] = v[u] * __longlong_as_double((long long)xb[u]);
            flow_wave_sync();
            if (leader) {
                // the products of the row's LPR lanes in stored order: lane g's slots 0 .. SEG-1 (padding = +0.0, which
                // leaves the running sum's bits alone), GB lanes' worth of LDS reads in flight at a time
                constexpr int GB = LPR < 4 ? LPR : 4;
#pragma unroll 1
                for (int g0 = 0; g0 < LPR; g0 += GB) {
                    double p[GB * SEG];
#pragma unroll
                    for (int g = 0; g < GB; ++g)
#pragma unroll
                        for (int u = 0; u < SEG; ++u) p[g * SEG + u] = prod[u * 64 + lane + g0 + g];
#pragma unroll
                    for (int w = 0; w < GB * SEG; ++w) acc = BSR1 ? (acc - p[w]) : (acc + p[w]);
                }
            }
            flow_wave_sync();
        }
        if (leader) {
            const double xn = BSR1 ? (acc / dd) : ((bb - acc) / dd);
            __hip_atomic_store(Xn + k, (unsigned long long)__double_as_longlong(xn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last) a.x_out[orow] = xn;
        }
        s = s2; q = q2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Block Gauss-Seidel (relaxation.h:756-810): x_i = Dinv_i (b_i - sum_{j != i} A_ij x_j), block rows in sequence.  A lane
// per SCALAR row r of a block row, LPR lanes per scalar row: lane (block row, g, r) owns blocks [g * seg, (g + 1) * seg) of
// its block row -- row r of each -- and forms v = sum_c a[r][c] x_j[c] from 0 for them (linalg.h gemm); the g = 0 lane adds
// the v's of all groups in stored order, the BS leaders of a block row exchange b - rsum by shuffles and apply Dinv.
// Same arithmetic as bsr_stream_kernel / bsell_kernel<BM_BLOCK_GS>.
struct BlockFlowArgs {
    unsigned long long *X;
    double *x_out;
    const double *Dinv;
    int *status;
    long xstride;
    long long budget;
    int nchunks, nseq, nb;
    unsigned dirmask;
    int xcd;
};

template <int BS>
__global__ __launch_bounds__(256) void bflow_gather_kernel(const int *rows, const double *x, const double *b, unsigned long long *X,
                                                            double *bp, long xstride, int nb, int nseq)
{
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)nb * BS;
    if (p >= n + BS) return;
    if (p >= n) {
        for (int s = 0; s <= nseq; ++s) X[(long)s * xstride + p] = 0ULL;
        return;
    }
    const long kb = p / BS;
    const int r = (int)(p - kb * BS);
    const long i = (long)rows[kb] * BS + r;
    unsigned long long xb = (unsigned long long)__double_as_longlong(x[i]);
    if (xb == FLOW_SENT) xb |= FLOW_QUIET;
    X[p] = xb;
    bp[p] = b[i];
    for (int s = 1; s <= nseq; ++s) X[(long)s * xstride + p] = FLOW_SENT;
}

// what a block task needs that does not depend on the sweep's values: chunk descriptor, block columns, values, and the
// block row's own right-hand side / gate / row number.  (Requesting the NEXT task's share before waiting for the current
// task's operands was tried -- two of these in registers -- and gained nothing: vector-memory results return in issue
// order, so the operand polls queue behind the prefetch; C5 at 252^3: 5.17 vs 5.10 ms per level-0 application.)
// SEG blocks per lane, LPR lanes per scalar row: (8, 1) up to 8 off-diagonal blocks per block row, (5, 3) up to 15 -- the
// tet-mesh operator of configuration C5, 7 block rows per wave without a padded slot and few enough registers for four
// waves per SIMD --, (8, 2), (8, 4), ... beyond.
template <int BS, int SEG>
struct BlockTask {
    int bc[SEG];
    double v[SEG][BS];
    double bb;
    double dinv[BS];            // leader lanes: row r of the block row's inverted diagonal block
    int gate, orow;
};

// original block-row number and gate of a task's leader lanes (0 / the zero position elsewhere): requested one task before
// the rest of the task's share, so that neither the Dinv request nor the first gate poll has to wait for an index
struct BlockAhead { int orow, gate; };
template <int BS, int LPR>
__device__ __forceinline__ BlockAhead bflow_ahead(bool valid, int s, const FlowChunk &m, const int *__restrict__ rows,
                                                  const int *__restrict__ gate_f, const int *__restrict__ gate_b, const BlockFlowArgs &a)
{
    constexpr int LW = BS * LPR, NBR = 64 / LW;
    const int lane = threadIdx.x;
    const int br = lane / LW;
    const bool leader = valid && br < m.nrows && br < NBR && (lane - br * LW) < BS;
    const bool rev = ((a.dirmask >> s) & 1u) != 0;
    BlockAhead h;
    h.orow = 0; h.gate = a.nb;
    if (leader) {
        h.orow = rows[m.row0 + br];
        h.gate = rev ? gate_b[m.row0 + br] : gate_f[m.row0 + br];
    }
    return h;
}

// block columns and values of task (s, m): the HBM stream
template <int BS, int SEG, int LPR>
__device__ __forceinline__ void bflow_load_matrix(BlockTask<BS, SEG> &T, bool valid, const FlowChunk &m, const int *__restrict__ col,
                                                   const double *__restrict__ val)
{
    constexpr int NG = 64 / BS;
    if (!valid) return;
    const int lane = threadIdx.x;
    const int grp = lane / BS;
    const int seg = m.nslots;
    const int smax = seg > 0 ? seg - 1 : 0;
    const int gcl = min(grp, NG - 1);
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
        const long su = (long)m.off + min(u, smax);
        T.bc[u] = col[su * NG + gcl];
#pragma unroll
        for (int cc = 0; cc < BS; ++cc) T.v[u][cc] = val[(su * BS + cc) * 64 + lane];
    }
}

// what the leader lanes of task (s, m) need besides: right-hand side, gate, their row of Dinv
template <int BS, int SEG, int LPR>
__device__ __forceinline__ void bflow_load_leader(BlockTask<BS, SEG> &T, bool valid, const FlowChunk &m, const BlockAhead &h,
                                                   const double *__restrict__ bp, const BlockFlowArgs &a)
{
    constexpr int LW = BS * LPR, NBR = 64 / LW;
    if (!valid) return;
    const int lane = threadIdx.x;
    const int br = lane / LW, grp = lane / BS, r = lane - grp * BS;
    const bool rowok = br < m.nrows && br < NBR;
    const bool leader = rowok && (lane - br * LW) < BS;
    const int kb = m.row0 + (rowok ? br : 0);
    T.bb = 0.0; T.orow = h.orow; T.gate = h.gate;
#pragma unroll
    for (int cc = 0; cc < BS; ++cc) T.dinv[cc] = 0.0;
    if (leader) {
        T.bb = bp[(long)kb * BS + r];
#pragma unroll
        for (int cc = 0; cc < BS; ++cc) T.dinv[cc] = a.Dinv[(long)h.orow * (BS * BS) + r * BS + cc];
    }
}

// first half of a task: waits for the operands and forms the per-block products vb[] (in stored order, from 0).
// returns false when the wave ran out of its time budget
template <int BS, int SEG, int LPR>
__device__ __forceinline__ bool bflow_operands(BlockTask<BS, SEG> &T, int s, const FlowChunk &m, double (&vb)[SEG], long long t0,
                                               const BlockFlowArgs &a)
{
    constexpr int NG = 64 / BS;
    const int lane = threadIdx.x;
    const int grp = lane / BS, r = lane - grp * BS;
    const int seg = m.nslots;
    const bool rev = ((a.dirmask >> s) & 1u) != 0;
    const unsigned long long *Xo = a.X + (long)s * a.xstride;
    const unsigned long long *gp_ = a.X + (long)(s + 1) * a.xstride + (long)T.gate * BS + r;
    unsigned long long gv = ald(gp_);
    for (unsigned spin = 0; __any(gv == FLOW_SENT); ++spin) {
        __builtin_amdgcn_s_sleep(4);
        if ((spin & 31u) == 31u && wall_clock64() - t0 > a.budget) {
            if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (gv == FLOW_SENT) gv = ald(gp_);
    }
    const int xs = (int)a.xstride;
    int idx[SEG];
    unsigned long long xb[SEG][BS];
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
        const bool real = u < seg && seg > 0 && lane < NG * BS;
        const int cb = real ? T.bc[u] : a.nb;
#pragma unroll
        for (int cc = 0; cc < BS; ++cc) T.v[u][cc] = real ? T.v[u][cc] : 0.0;
        const bool fresh = rev ? (cb >= m.lvl_hi) : (cb < m.lvl_lo);
        idx[u] = cb * BS + (fresh ? xs : 0);
#pragma unroll
        for (int cc = 0; cc < BS; ++cc) xb[u][cc] = pld(Xo + idx[u] + cc);
    }
    for (unsigned spin = 0;; ++spin) {
        bool bad = false;
#pragma unroll
        for (int u = 0; u < SEG; ++u)
#pragma unroll
            for (int cc = 0; cc < BS; ++cc) bad |= (xb[u][cc] == FLOW_SENT);
        if (!__any(bad)) break;
        if (AMG_FLOW_SLEEP > 0) __builtin_amdgcn_s_sleep(AMG_FLOW_SLEEP);
        if ((spin & 31u) == 31u && wall_clock64() - t0 > a.budget) {
            if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
#pragma unroll
        for (int u = 0; u < SEG; ++u)
#pragma unroll
            for (int cc = 0; cc < BS; ++cc)
                if (xb[u][cc] == FLOW_SENT) xb[u][cc] = ald(Xo + idx[u] + cc);
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
        double w = 0.0;
#pragma unroll
        for (int cc = 0; cc < BS; ++cc) w = w + T.v[u][cc] * __longlong_as_double((long long)xb[u][cc]);
        vb[u] = w;
    }
    return true;
}

// second half: row sums in stored order, x_i = Dinv_i (b_i - rsum), store
template <int BS, int SEG, int LPR>
__device__ __forceinline__ void bflow_finish(const BlockTask<BS, SEG> &Q, int s, const FlowChunk &m, const double (&vb)[SEG], double *prod,
                                             const BlockFlowArgs &a)
{
    constexpr int LW = BS * LPR, NBR = 64 / LW;
    const int lane = threadIdx.x;
    const int br = lane / LW, grp = lane / BS, r = lane - grp * BS;
    const bool last = s == a.nseq - 1;
    const bool rowok = br < m.nrows && br < NBR;
    const bool leader = rowok && (lane - br * LW) < BS;
    const int kb = m.row0 + (rowok ? br : 0);
    unsigned long long *Xn = a.X + (long)(s + 1) * a.xstride;
    double rsum = 0.0;
    if constexpr (LPR == 1) {
#pragma unroll
        for (int u = 0; u < SEG; ++u) rsum = rsum + vb[u];
    } else {
#pragma unroll
        for (int u = 0; u < SEG; ++u) prod[u * 64 + lane] = vb[u];
        flow_wave_sync();
        if (leader) {
            constexpr int GB = LPR < 4 ? LPR : 4;
#pragma unroll 1
            for (int g0 = 0; g0 < LPR; g0 += GB) {
                double p[GB * SEG];
#pragma unroll
                for (int g = 0; g < GB; ++g)
#pragma unroll
                    for (int u = 0; u < SEG; ++u) p[g * SEG + u] = prod[u * 64 + lane + (g0 + g) * BS];
#pragma unroll
                for (int w = 0; w < GB * SEG; ++w) rsum = rsum + p[w];
            }
        }
        flow_wave_sync();
    }
    // x_i = Dinv_i (b_i - rsum): the BS leaders of the block row exchange their entries of b - rsum
    const double t = Q.bb - rsum;
    double vD = 0.0;
    const int base = br * LW;
#pragma unroll
    for (int cc = 0; cc < BS; ++cc) {
        const double tc = __shfl(t, base + cc, 64);
        vD = vD + Q.dinv[cc] * tc;
    }
    if (leader) {
        __hip_atomic_store(Xn + (long)kb * BS + r, (unsigned long long)__double_as_longlong(vD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last) a.x_out[(long)Q.orow * BS + r] = vD;
    }
}

template <int BS, int SEG, int LPR>
__global__ __launch_bounds__(64) void bgs_flow_kernel(const FlowChunk *__restrict__ meta, const int *__restrict__ col,
                                                       const double *__restrict__ val, const double *__restrict__ bp,
                                                       const int *__restrict__ rows, const int *__restrict__ gate_f,
                                                       const int *__restrict__ gate_b, BlockFlowArgs a)
{
    __shared__ double prod[LPR > 1 ? SEG * 64 : 1];
    const int NW = (int)gridDim.x;
    const long long t0 = wall_clock64();
    int s = 0, q = flow_slot((int)blockIdx.x, NW, a.xcd);
    while (q >= a.nchunks) { q -= a.nchunks; ++s; }
    BlockTask<BS, SEG> T;
    auto chunk_of = [&](int ss, int qq) { return (((a.dirmask >> ss) & 1u) != 0) ? a.nchunks - 1 - qq : qq; };
    auto advance = [&](int &ss, int &qq) { qq += NW; while (qq >= a.nchunks) { qq -= a.nchunks; ++ss; } };
    // task i = (s, q), descriptor m0, its share in T; m1 / h1: descriptor and leader indices of task i + 1; m2: descriptor of i + 2
    int s1 = s, q1 = q;
    advance(s1, q1);
    FlowChunk m0 = meta[s < a.nseq ? chunk_of(s, q) : 0];
    FlowChunk m1 = meta[s1 < a.nseq ? chunk_of(s1, q1) : 0];
    bflow_load_matrix<BS, SEG, LPR>(T, s < a.nseq, m0, col, val);
    bflow_load_leader<BS, SEG, LPR>(T, s < a.nseq, m0, bflow_ahead<BS, LPR>(s < a.nseq, s, m0, rows, gate_f, gate_b, a), bp, a);
    BlockAhead h1 = bflow_ahead<BS, LPR>(s1 < a.nseq, s1, m1, rows, gate_f, gate_b, a);
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): with nothing of the prologue pending, the loop's first poll is not made to wait for the share in flight
    while (s < a.nseq) {
        int s2 = s1, q2 = q1;
        advance(s2, q2);
        const FlowChunk m2 = meta[s2 < a.nseq ? chunk_of(s2, q2) : 0];      // scalar load, not needed before the products are formed
        double vb[SEG];
        if (!bflow_operands<BS, SEG, LPR>(T, s, m0, vb, t0, a)) return;
        // the products are formed: T's matrix registers are free, and the next task's columns and values (HBM) travel while
        // this one sums, solves and stores.  Requested any earlier they would sit in front of the operand polls (in-order
        // returns); the compiler must neither lift them above the products nor sink the products below them
#pragma unroll
        for (int u = 0; u < SEG; ++u) asm volatile("" : "+v"(vb[u]));
        __builtin_amdgcn_sched_barrier(0);
        bflow_load_matrix<BS, SEG, LPR>(T, s1 < a.nseq, m1, col, val);
        bflow_finish<BS, SEG, LPR>(T, s, m0, vb, prod, a);
        __builtin_amdgcn_sched_barrier(0);
        bflow_load_leader<BS, SEG, LPR>(T, s1 < a.nseq, m1, h1, bp, a);        // into the registers the store just freed
        h1 = bflow_ahead<BS, LPR>(s2 < a.nseq, s2, m2, rows, gate_f, gate_b, a);
        s = s1; q = q1; s1 = s2; q1 = q2; m0 = m1; m1 = m2;
    }
}

// ---------------------------------------------------------------------------
// Gauss-Seidel in the operator's OWN row order, straight from its CSR arrays: no level-ordered copy, no host-built
// schedule -- for the setup's candidate improvement (aggregation.py:313-320: a few sweeps of A x = 0 over an operator that
// is in HBM anyway for the spectral-radius estimate), where building a dataflow form would cost more than the sweeps.
// A task is 64 consecutive rows of the sweep order, one per lane; tasks go to the resident one-wave workgroups in a given
// order in which every task comes after all tasks it depends on (ascending if none is given), so the first unfinished
// task of that order never waits for a wave that is not running.  (In ASCENDING order a lexicographic 3-D grid offers
// only a handful of independent tasks among the few thousand resident ones -- a grid line waits for the line before it
// -- and 500^3 did not finish a sweep in 20 s; amg_hier_gs_natural therefore cuts the tasks of a stencil-form operator
// where the chain of neighbouring rows breaks (grid-line starts) and orders them by their dependency level, computed from
// the stencil's offsets and row masks at TASK granularity.)  A lane (i) gathers its row (at most NAT_SEG
// entries, checked on the host): operands the sweep has not reached are read from x_old (final: the previous sweep was
// another launch), operands of earlier TASKS are polled in x_new (sentinel-prefilled), (ii) once those are there, waits in
// LDS for the operands produced by earlier lanes of its own wave, then forms the row sum over the stored entries in
// stored order with separate multiply and add -- relaxation.h:34-62 bit for bit -- and publishes x_i to LDS and memory.
// On a 7-point grid operator the in-wave chain (x_{i-1}) makes a task ~64 LDS hand-offs long; tasks of different grid lines
// overlap, a sweep over 125 M rows is a few milliseconds of dependency depth plus the stream of the CSR arrays.
// ---------------------------------------------------------------------------
constexpr int NAT_SEG = 8;
struct NatArgs {
    const int *Ap, *Aj;
    const double *Ax, *b, *xold;
    unsigned long long *xnew;
    const int *order;              // tasks in the order they are taken (null: ascending); every task after all it depends on
    const int *tstart;             // task q = positions [tstart[q], tstart[q + 1]) of the sweep, at most 64 (null: 64 q ...)
    int *status;
    long long budget;
    int n, ntasks, reverse;
};

__global__ __launch_bounds__(64) void gs_natural_kernel(NatArgs a)
{
    __shared__ unsigned long long sx[64];
    const int lane = threadIdx.x, NW = (int)gridDim.x;
    const long long t0 = wall_clock64();
    for (int slot = blockIdx.x; slot < a.ntasks; slot += NW) {
        const int q = a.order ? a.order[slot] : slot;
        const int p0 = a.tstart ? a.tstart[q] : q * 64, p = p0 + lane;          // positions in sweep order
        const bool have = a.tstart ? (p < a.tstart[q + 1]) : (p < a.n);
        const int i = have ? (a.reverse ? a.n - 1 - p : p) : 0;
        sx[lane] = FLOW_SENT;
        int start = 0, len = 0;
        if (have) { start = a.Ap[i]; len = a.Ap[i + 1] - start; }
        double v[NAT_SEG];
        unsigned long long xb[NAT_SEG];
        int cj[NAT_SEG], kind[NAT_SEG];                              // 0 none, 1 diagonal, 2 value in xb, 3 earlier task (poll), 4 earlier lane (LDS slot in cj)
#pragma unroll
        for (int e = 0; e < NAT_SEG; ++e) {
            v[e] = 0.0; xb[e] = 0ULL; cj[e] = 0; kind[e] = 0;
            if (e < len) {
                const int j = a.Aj[start + e];
                v[e] = a.Ax[start + e];
                cj[e] = j;
                if (j == i) kind[e] = 1;
                else {
                    const bool fresh = a.reverse ? (j > i) : (j < i);
                    if (!fresh) { kind[e] = 2; xb[e] = (unsigned long long)__double_as_longlong(a.xold[j]); }
                    else {
                        const int pj = a.reverse ? a.n - 1 - j : j;
                        if (pj >= p0) { kind[e] = 4; cj[e] = pj - p0; }
                        else { kind[e] = 3; xb[e] = ald(a.xnew + j); }
                    }
                }
            }
        }
        for (unsigned spin = 0;; ++spin) {                           // operands of earlier tasks
            bool bad = false;
#pragma unroll
            for (int e = 0; e < NAT_SEG; ++e) bad |= (kind[e] == 3 && xb[e] == FLOW_SENT);
            if (!__any(bad)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spin & 31u) == 31u && wall_clock64() - t0 > a.budget) {
                if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
#pragma unroll
            for (int e = 0; e < NAT_SEG; ++e)
                if (kind[e] == 3 && xb[e] == FLOW_SENT) xb[e] = ald(a.xnew + cj[e]);
        }
        const double bi = (have && a.b) ? a.b[i] : 0.0;
        const double xo = have ? a.xold[i] : 0.0;
        flow_wave_sync();                                            // every lane's slot holds the sentinel
        bool done = !have;
        unsigned long long res = 0ULL;
        for (unsigned round = 0; round < 66u; ++round) {             // lane l's operands come from lanes < l: at most 64 rounds
            if (!done) {
                bool ready = true;
#pragma unroll
                for (int e = 0; e < NAT_SEG; ++e)
                    if (kind[e] == 4) { xb[e] = sx[cj[e]]; ready = ready && (xb[e] != FLOW_SENT); }
                if (ready) {
                    double rsum = 0.0, diag = 0.0;
#pragma unroll
                    for (int e = 0; e < NAT_SEG; ++e) {
                        if (kind[e] == 1) diag = v[e];
                        else if (kind[e] != 0) rsum = rsum + v[e] * __longlong_as_double((long long)xb[e]);
                    }
                    double xi = xo;
                    if (diag != 0.0) xi = (bi - rsum) / diag;
                    unsigned long long u = (unsigned long long)__double_as_longlong(xi);
                    if (u == FLOW_SENT) u |= FLOW_QUIET;
                    res = u;
                    sx[lane] = u;                                    // (to memory after the chain: a store per round would put
                    done = true;                                     //  a memory round trip into every hand-off -- 0.64 s per sweep at 500^3)
                }
            }
            flow_wave_sync();
            if (!__any(!done)) break;
        }
        if (have) __hip_atomic_store(a.xnew + i, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flow_wave_sync();                                            // all reads of sx done before the next task resets it
    }
}

__global__ __launch_bounds__(256) void nat_fill_kernel(unsigned long long *p, long n)
{
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < n) p[k] = FLOW_SENT;
}

int g_flow_mode = std::getenv("AMG_GS_FLOW") ? std::atoi(std::getenv("AMG_GS_FLOW")) : 1;   // A/B runs of tools without a knob of their own
int g_flow_la = 0;              // look-ahead in levels; 0: default
int *g_status = nullptr;        // device word, one per process and device (first use)
int g_status_dev = -1;

int flow_status_word(int **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice", __FILE__, __LINE__);
    if (!g_status || g_status_dev != dev) {
        e = hipMalloc((void **)&g_status, 64);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (dataflow status word)", __FILE__, __LINE__);
        e = hipMemset(g_status, 0, 64);
        if (e != hipSuccess) return hip_fail(e, "hipMemset", __FILE__, __LINE__);
        g_status_dev = dev;
    }
    *out = g_status;
    return 0;
}

// consecutive tasks on one XCD: measured -4 .. -7 % per sweep on the point operators of a 3-D SA hierarchy (128^3: 1.17 -> 1.12,
// 1.40 -> 1.30, 1.94 -> 1.87 us per level; 200^3 cycle 16.9 -> 16.1 ms), level on the wide block levels of C5 (off there)
// Task-to-XCD mapping (flow_slot): point sweeps 1 (consecutive slots per XCD; SA + symmetric GS at 200^3 16.8 -> 16.0 ms per
// cycle), block sweeps groups of 64 tasks dealt round-robin (C5 at 360^3 with 8 waves per CU: 11.9 -> 11.5 ms per level-0
// application; the contiguous mapping keeps only the XCDs of the current levels busy there: 4.99 vs 4.70 ms at 252^3)
int flow_xcd(bool block)
{
    static const int x = std::getenv("AMG_FLOW_XCD") ? std::atoi(std::getenv("AMG_FLOW_XCD")) : 1;
    static const int xb = std::getenv("AMG_FLOW_XCD_BLOCK") ? std::atoi(std::getenv("AMG_FLOW_XCD_BLOCK")) : 64;
    return block ? xb : x;
}

// one-wave workgroups a compute unit is asked to hold at most (the occupancy query is the other bound).  Block sweeps: 8 --
// their time is bytes / 6.2 TB/s + levels x 1.7 us (profiles/r03_c5_traffic.txt), and waves beyond what covers one hop
// only put their 8 KB shares in front of the operand polls of the level in progress
int flow_wpc(bool block)
{
    static const int w = std::getenv("AMG_FLOW_WPC") ? std::atoi(std::getenv("AMG_FLOW_WPC")) : 16;
    static const int wb = std::getenv("AMG_FLOW_WPC_BLOCK") ? std::atoi(std::getenv("AMG_FLOW_WPC_BLOCK"))
                                                            : (std::getenv("AMG_FLOW_WPC") ? w : 8);
    return std::max(1, block ? wb : w);
}

template <int LPR>
int flow_waves_cap(bool bsr1)
{
    // resident waves the launch may count on: occupancy query, at most 8 per compute unit, 3/4 of that as a margin
    static int cap[2] = {0, 0};
    int &c = cap[bsr1 ? 1 : 0];
    if (c == 0) {
        int nb = 0, dev = 0, ncu = 0;
        hipError_t e = bsr1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gs_flow_kernel<FLOW_SEG, LPR, true>, 64, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gs_flow_kernel<FLOW_SEG, LPR, false>, 64, 0);
        if (e != hipSuccess || nb < 1) nb = 1;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu < 1) ncu = 64;
        c = std::max(16, ncu * std::min(nb, flow_wpc(false)) * 3 / 4);
    }
    return c;
}

template <int LPR>
int launch_flow(const FlowForm &F, bool bsr1, const FlowArgs &a, hipStream_t st)
{
    const int la = g_flow_la > 0 ? g_flow_la : 4;
    long nw = (long)la * ((F.nchunks + std::max(1, F.nlevels) - 1) / std::max(1, F.nlevels));
    static const int minw = std::getenv("AMG_FLOW_MINW") ? std::atoi(std::getenv("AMG_FLOW_MINW")) : 32;
    nw = std::max<long>(nw, minw);
    nw = std::min<long>(nw, flow_waves_cap<LPR>(bsr1));
    nw = std::min<long>(nw, (long)a.nseq * F.nchunks);
    if (a.xcd >= 2 && nw >= 8L * a.xcd) nw -= nw % (8L * a.xcd);
    if (bsr1) hipLaunchKernelGGL((gs_flow_kernel<FLOW_SEG, LPR, true>), dim3((unsigned)nw), dim3(64), 0, st, F.meta, F.col, F.val, F.diag, F.bp, F.rowmap, F.gate_f, F.gate_b, a);
    else hipLaunchKernelGGL((gs_flow_kernel<FLOW_SEG, LPR, false>), dim3((unsigned)nw), dim3(64), 0, st, F.meta, F.col, F.val, F.diag, F.bp, F.rowmap, F.gate_f, F.gate_b, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "dataflow Gauss-Seidel launch", __FILE__, __LINE__);
    return 0;
}

template <int BS, int SEG, int LPR>
int launch_bflow(const BlockFlowForm &F, const BlockFlowArgs &a, hipStream_t st)
{
    static int cap = 0;
    if (cap == 0) {
        int nb = 0, dev = 0, ncu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bgs_flow_kernel<BS, SEG, LPR>, 64, 0) != hipSuccess || nb < 1) nb = 1;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu < 1) ncu = 64;
        cap = std::max(16, ncu * std::min(nb, flow_wpc(true)) * 3 / 4);
    }
    const int la = g_flow_la > 0 ? g_flow_la : 4;
    long nw = (long)la * ((F.nchunks + std::max(1, F.nlevels) - 1) / std::max(1, F.nlevels));
    static const int minw = std::getenv("AMG_FLOW_MINW") ? std::atoi(std::getenv("AMG_FLOW_MINW")) : 32;
    nw = std::max<long>(nw, minw);
    nw = std::min<long>(nw, cap);
    nw = std::min<long>(nw, (long)a.nseq * F.nchunks);
    if (a.xcd >= 2 && nw >= 8L * a.xcd) nw -= nw % (8L * a.xcd);
    hipLaunchKernelGGL((bgs_flow_kernel<BS, SEG, LPR>), dim3((unsigned)nw), dim3(64), 0, st, F.meta, F.col, F.val, F.bp, F.rows, F.gate_f, F.gate_b, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "dataflow block Gauss-Seidel launch", __FILE__, __LINE__);
    return 0;
}

template <int BS>
int launch_bflow_bs(const BlockFlowForm &F, const BlockFlowArgs &a, hipStream_t st)
{
    if (F.seg == 5 && F.lpr == 3) return launch_bflow<BS, 5, 3>(F, a, st);
    switch (F.lpr) {
    case 1: return launch_bflow<BS, 8, 1>(F, a, st);
    case 2: return launch_bflow<BS, 8, 2>(F, a, st);
    case 4: return launch_bflow<BS, 8, 4>(F, a, st);
    case 8: return launch_bflow<BS, 8, 8>(F, a, st);
    default: return launch_bflow<BS, 8, 16>(F, a, st);
    }
}

}  // namespace

void BlockFlowForm::release()
{
    for (void *p : {(void *)rows, (void *)meta, (void *)col, (void *)val, (void *)bp, (void *)X, (void *)gate_f, (void *)gate_b})
        if (p) hipFree(p);
    rows = nullptr; meta = nullptr; col = nullptr; val = nullptr; bp = nullptr; X = nullptr; gate_f = gate_b = nullptr;
    ready = false;
    bytes = 0;
}

// rows / gp / gj / gx: the block schedule's level-ordered copy (block row k of it = original block row rows[k], ORIGINAL
// block columns, blocks row-major)
int build_block_flow_form(BlockFlowForm &F, int nb, int bs, int ntasks, const std::vector<int> &level_ptr, const std::vector<int> &rows,
                          const std::vector<int> &gp, const std::vector<int> &gj, const std::vector<double> &gx)
{
    F.release();
    const int nl = (int)level_ptr.size() - 1;
    if (nb <= 0 || ntasks != nb || nl <= 0 || (bs != 2 && bs != 3)) return 0;
    if ((double)nb * bs * (FLOW_MAXSEQ + 1) >= 2.0e9) return 0;
    const long B2 = (long)bs * bs;
    std::vector<int> cnt((size_t)nb), seen((size_t)nb, 0);
    int longest = 0;
    for (int k = 0; k < nb; ++k) {
        const int i = rows[(size_t)k];
        if (i < 0 || i >= nb || seen[(size_t)i]) return 0;
        seen[(size_t)i] = 1;
        int c = 0;
        for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) {
            const int j = gj[(size_t)q];
            if (j < 0 || j >= nb) return 0;
            if (j != i) ++c;                                                    // diagonal blocks are not part of the sum (relaxation.h:785)
        }
        cnt[(size_t)k] = c;
        longest = std::max(longest, c);
    }
    if (longest > FLOW_SEG * 16) return 0;
    int lpr = 1, seg = FLOW_SEG;
    while (lpr * seg < longest) lpr *= 2;
    static const int allow53 = std::getenv("AMG_FLOW_53") ? std::atoi(std::getenv("AMG_FLOW_53")) : 1;
    if (allow53 && longest > 8 && longest <= 15) { seg = 5; lpr = 3; }          // 15 = 3 lanes x 5 blocks: the C5 operator's block rows
    F.lpr = lpr; F.seg = seg; F.bs = bs;
    const int NG = 64 / bs, R = 64 / (bs * lpr);                                // groups per wave, block rows per chunk
    if (R < 1) return 0;
    std::vector<int> ord((size_t)nb), pos_of((size_t)nb);
    flow_parallel(nl, 64, [&](long llo, long lhi) {
        for (long l = llo; l < lhi; ++l) {
            const int lo = level_ptr[(size_t)l], hi = level_ptr[(size_t)l + 1];
            for (int k = lo; k < hi; ++k) ord[(size_t)k] = k;
            // (by SLOTS PER LANE, not by length: rows of one class keep the schedule's order -- neighbours in the operator's
            //  numbering stay neighbours in a chunk and their gathers share cache lines)
            std::stable_sort(ord.begin() + lo, ord.begin() + hi, [&](int p, int q) { return (cnt[(size_t)p] + lpr - 1) / lpr > (cnt[(size_t)q] + lpr - 1) / lpr; });
        }
    });
    flow_parallel(nb, 1 << 18, [&](long klo, long khi) { for (long k = klo; k < khi; ++k) pos_of[(size_t)rows[(size_t)ord[(size_t)k]]] = (int)k; });
    std::vector<FlowChunk> meta;
    long slot_rows = 0;
    for (int l = 0; l < nl; ++l) {
        const int lo = level_ptr[(size_t)l], hi = level_ptr[(size_t)l + 1];
        for (int r0 = lo; r0 < hi; r0 += R) {
            FlowChunk m;
            std::memset(&m, 0, sizeof(m));
            m.row0 = r0; m.nrows = std::min(R, hi - r0);
            m.nslots = (cnt[(size_t)ord[(size_t)r0]] + lpr - 1) / lpr;
            m.lvl_lo = lo; m.lvl_hi = hi;
            if (slot_rows > 2000000000L) return 0;                              // FlowChunk::off is an int (entries are addressed in 64 bits)
            m.off = (int)slot_rows;
            slot_rows += m.nslots;
            meta.push_back(m);
        }
    }
    // (every slot is written exactly once, padding included, by the thread that owns the chunk: no zero-fill pass)
    const size_t ncol = (size_t)std::max(slot_rows, 1L) * NG, nval = (size_t)std::max(slot_rows, 1L) * bs * 64;
    std::unique_ptr<int[]> col(new int[ncol]);
    std::unique_ptr<double[]> val(new double[nval]);
    std::vector<int> rmap((size_t)nb), lev((size_t)nb), gf((size_t)nb, nb), gb((size_t)nb, nb);
    flow_parallel(nl, 64, [&](long llo, long lhi) {
        for (long l = llo; l < lhi; ++l)
            for (int k = level_ptr[(size_t)l]; k < level_ptr[(size_t)l + 1]; ++k) lev[(size_t)k] = (int)l;
    });
    flow_parallel((long)meta.size(), 256, [&](long clo, long chi) {
        for (long cc = clo; cc < chi; ++cc) {
            const FlowChunk &m = meta[(size_t)cc];
            for (size_t w = (size_t)m.off * NG; w < ((size_t)m.off + (size_t)m.nslots) * NG; ++w) col[w] = nb;
            for (size_t w = (size_t)m.off * bs * 64; w < ((size_t)m.off + (size_t)m.nslots) * bs * 64; ++w) val[w] = 0.0;
            for (int t = 0; t < m.nrows; ++t) {
                const int kk = m.row0 + t;
                const int k = ord[(size_t)kk], i = rows[(size_t)k];
                rmap[(size_t)kk] = i;
                const int lk = lev[(size_t)kk];
                int j = 0, gfl = -1, gbl = nl;
                for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) {
                    const int cj = gj[(size_t)q];
                    if (cj == i) continue;
                    const int g = j / m.nslots, u = j % m.nslots;
                    const int pc = pos_of[(size_t)cj], lc = lev[(size_t)pc];
                    const int grp = t * lpr + g;                                // lanes grp * bs .. grp * bs + bs - 1
                    col[((size_t)m.off + (size_t)u) * NG + (size_t)grp] = pc;
                    for (int r = 0; r < bs; ++r)
                        for (int c = 0; c < bs; ++c)
                            val[(((size_t)m.off + (size_t)u) * bs + (size_t)c) * 64 + (size_t)grp * bs + (size_t)r] = gx[(size_t)q * B2 + (size_t)r * bs + (size_t)c];
                    if (lc <= lk - 2 && lc > gfl) { gfl = lc; gf[(size_t)kk] = pc; }
                    if (lc >= lk + 2 && lc < gbl) { gbl = lc; gb[(size_t)kk] = pc; }
                    ++j;
                }
            }
        }
    });
    F.nb = nb; F.nchunks = (int)meta.size(); F.nlevels = nl; F.slot_rows = slot_rows;
    F.xstride = ((long)(nb + 1) * bs + 15) / 16 * 16;
    long acct = 0;
    FCHK(falloc(&F.rows, nb, &acct));
    FCHK(falloc(&F.meta, (long)meta.size(), &acct));
    FCHK(falloc(&F.col, (long)ncol, &acct));
    FCHK(falloc(&F.val, (long)nval, &acct));
    FCHK(falloc(&F.bp, (long)nb * bs, &acct));
    FCHK(falloc(&F.gate_f, nb, &acct));
    FCHK(falloc(&F.gate_b, nb, &acct));
    FCHK(falloc(&F.X, (FLOW_MAXSEQ + 1) * F.xstride, &acct));
    AMG_HIP(hipMemcpy(F.rows, rmap.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.meta, meta.data(), sizeof(FlowChunk) * meta.size(), hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.col, col.get(), sizeof(int) * ncol, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.val, val.get(), sizeof(double) * nval, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.gate_f, gf.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.gate_b, gb.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice));
    F.bytes = acct;
    F.ready = true;
    return 0;
}

int block_flow_sweep(const BlockFlowForm &F, const double *Dinv, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st)
{
    if (!F.ready) { set_error("dataflow block Gauss-Seidel form was not built"); return -3; }
    int *status = nullptr;
    FCHK(flow_status_word(&status));
    for (int s0 = 0; s0 < nseq; s0 += FLOW_MAXSEQ) {
        const int ns = std::min(FLOW_MAXSEQ, nseq - s0);
        const long np = (long)(F.nb + 1) * F.bs;
        if (F.bs == 3) hipLaunchKernelGGL((bflow_gather_kernel<3>), dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, F.rows, x, b, (unsigned long long *)F.X, F.bp, F.xstride, F.nb, ns);
        else hipLaunchKernelGGL((bflow_gather_kernel<2>), dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, F.rows, x, b, (unsigned long long *)F.X, F.bp, F.xstride, F.nb, ns);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "dataflow block gather launch", __FILE__, __LINE__);
        BlockFlowArgs a;
        std::memset(&a, 0, sizeof(a));
        a.X = (unsigned long long *)F.X; a.x_out = x; a.Dinv = Dinv; a.status = status;
        a.xstride = F.xstride; a.budget = 100000000LL * 4;
        a.nchunks = F.nchunks; a.nseq = ns; a.nb = F.nb;
        a.xcd = flow_xcd(true);
        for (int k = 0; k < ns; ++k) a.dirmask |= (seq[s0 + k] != 0 ? 1u : 0u) << k;
        if (F.bs == 3) FCHK(launch_bflow_bs<3>(F, a, st));
        else FCHK(launch_bflow_bs<2>(F, a, st));
    }
    return 0;
}

namespace {
}  // namespace

// nsweeps directional sweeps (dirs[k] != 0: descending rows) of A x = b (b may be null: zero) on device arrays; x is
// updated in place.  Returns -40 when a row holds more than NAT_SEG entries (the caller falls back).
int gs_natural_sweeps(const int *Ap, const int *Aj, const double *Ax, int n, int longest_row, double *x, const double *b,
                      const unsigned char *dirs, int nsweeps, hipStream_t st, const int *order_fwd, const int *order_bwd,
                      const int *tstart_fwd, const int *tstart_bwd, int ntasks_fwd, int ntasks_bwd)
{
    if (longest_row > NAT_SEG) return -40;
    if (n <= 0 || nsweeps <= 0) return 0;
    int *status = nullptr;
    FCHK(flow_status_word(&status));
    static int cap = 0;
    if (cap == 0) {
        int nb = 0, dev = 0, ncu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gs_natural_kernel, 64, 0) != hipSuccess || nb < 1) nb = 1;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu < 1) ncu = 64;
        cap = std::max(16, ncu * std::min(nb, 24) * 3 / 4);
        if (std::getenv("AMG_NAT_WAVES") && std::atoi(std::getenv("AMG_NAT_WAVES")) > 0) cap = std::atoi(std::getenv("AMG_NAT_WAVES"));
        if (std::getenv("AMG_SETUP_VERBOSE") && std::atoi(std::getenv("AMG_SETUP_VERBOSE")))
            std::fprintf(stderr, "[setup]     natural-order Gauss-Seidel: %d one-wave workgroups (occupancy query %d per CU, %d CUs)\n", cap, nb, ncu);
    }
    double *tmp = nullptr;
    hipError_t e = hipMalloc((void **)&tmp, sizeof(double) * (size_t)n);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (natural-order Gauss-Seidel buffer)", __FILE__, __LINE__);
    double *cur = x, *nxt = tmp;
    const int ntasks = (n + 63) / 64;
    for (int k = 0; k < nsweeps; ++k) {
        hipLaunchKernelGGL(nat_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned long long *)nxt, (long)n);
        NatArgs a;
        a.Ap = Ap; a.Aj = Aj; a.Ax = Ax; a.b = b; a.xold = cur; a.xnew = (unsigned long long *)nxt;
        a.status = status; a.budget = 100000000LL * 20; a.n = n; a.reverse = dirs[k] != 0 ? 1 : 0;
        a.order = a.reverse ? order_bwd : order_fwd;
        a.tstart = a.reverse ? tstart_bwd : tstart_fwd;
        a.ntasks = a.tstart ? (a.reverse ? ntasks_bwd : ntasks_fwd) : ntasks;
        hipLaunchKernelGGL(gs_natural_kernel, dim3((unsigned)std::min(cap, a.ntasks)), dim3(64), 0, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) { hipFree(tmp); return hip_fail(e, "natural-order Gauss-Seidel launch", __FILE__, __LINE__); }
        std::swap(cur, nxt);
    }
    if (cur != x) e = hipMemcpyAsync(x, cur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st);
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
    if (e2 != hipSuccess) return hip_fail(e2, "hipStreamSynchronize", __FILE__, __LINE__);
    return 0;
}

void FlowForm::release()
{
    for (void *p : {(void *)rowmap, (void *)meta, (void *)col, (void *)val, (void *)diag, (void *)bp, (void *)X, (void *)gate_f, (void *)gate_b})
        if (p) hipFree(p);
    gate_f = gate_b = nullptr;
    rowmap = nullptr; meta = nullptr; col = nullptr; val = nullptr; diag = nullptr; bp = nullptr; X = nullptr;
    ready = false;
    bytes = 0;
}

int gs_flow_mode() { return g_flow_mode; }
void set_gs_flow(int mode) { g_flow_mode = mode; bump_config_epoch(); }
void set_gs_flow_lookahead(int levels) { g_flow_la = levels; bump_config_epoch(); }

int gs_flow_status()
{
    if (!g_status) return 0;
    int v = 0;
    if (hipMemcpy(&v, g_status, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v != 0) hipMemset(g_status, 0, sizeof(int));
    return v;
}

// rowmap / gp / gj / gx: the schedule's level-ordered copy (row k of it = original row rowmap[k], ORIGINAL columns)
int build_flow_form(FlowForm &F, int n, int ntasks, const std::vector<int> &level_ptr, const std::vector<int> &rowmap,
                    const std::vector<int> &gp, const std::vector<int> &gj, const std::vector<double> &gx, int ncols)
{
    F.release();
    const int nl = (int)level_ptr.size() - 1;
    if (ncols < n) ncols = n;
    if (n <= 0 || ntasks != n || nl <= 0) return 0;
    if ((double)ncols * (FLOW_MAXSEQ + 1) >= 2.0e9) return 0;                  // operand positions are 32-bit offsets across the buffers
    // every unknown listed exactly once, all columns inside, a nonzero diagonal in every row (relaxation.h:58-60
    // leaves a row with a zero diagonal untouched: such operators keep the level-scheduled kernels)
    std::vector<int> cnt((size_t)n), seen((size_t)n, 0);
    std::vector<double> dg((size_t)n, 0.0);
    for (int k = 0; k < n; ++k) {
        const int i = rowmap[(size_t)k];
        if (i < 0 || i >= n || seen[(size_t)i]) return 0;
        seen[(size_t)i] = 1;
    }
    std::vector<int>().swap(seen);
    std::atomic<int> longest_a(0), invalid(0);
    flow_parallel(n, 1 << 16, [&](long klo, long khi) {
        int lmax = 0;
        for (long k = klo; k < khi; ++k) {
            const int i = rowmap[(size_t)k];
            int c = 0;
            bool has_d = false;
            double d = 0.0;
            for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) {
                const int j = gj[(size_t)q];
                if (j < 0 || j >= ncols) { invalid.store(1); return; }
                if (j == i) { d = gx[(size_t)q]; has_d = true; }               // the last diagonal entry wins (relaxation.h:51-52)
                else ++c;
            }
            if (!has_d || d == 0.0) { invalid.store(1); return; }
            cnt[(size_t)k] = c;
            dg[(size_t)k] = d;
            lmax = std::max(lmax, c);
        }
        int cur = longest_a.load();
        while (lmax > cur && !longest_a.compare_exchange_weak(cur, lmax)) {}
    });
    if (invalid.load()) return 0;
    const int longest = longest_a.load();
    if (longest > FLOW_SEG * 64) return 0;
    int lpr = 1;
    while (lpr * FLOW_SEG < longest) lpr *= 2;
    F.lpr = lpr;
    const int R = 64 / lpr;                                                    // rows per chunk
    // rows of a level sorted by length class (longest first, ties in schedule order): chunks are nearly rectangular
    std::vector<int> ord((size_t)n), pos_of((size_t)n);
    flow_parallel(nl, 64, [&](long llo, long lhi) {
        for (long l = llo; l < lhi; ++l) {
            const int lo = level_ptr[(size_t)l], hi = level_ptr[(size_t)l + 1];
            for (int k = lo; k < hi; ++k) ord[(size_t)k] = k;
            // (by SLOTS PER LANE, not by length: rows of one class keep the schedule's order -- neighbours in the operator's
            //  numbering stay neighbours in a chunk and their gathers share cache lines)
            std::stable_sort(ord.begin() + lo, ord.begin() + hi, [&](int p, int q) { return (cnt[(size_t)p] + lpr - 1) / lpr > (cnt[(size_t)q] + lpr - 1) / lpr; });
        }
    });
    flow_parallel(n, 1 << 18, [&](long klo, long khi) { for (long k = klo; k < khi; ++k) pos_of[(size_t)rowmap[(size_t)ord[(size_t)k]]] = (int)k; });
    std::vector<FlowChunk> meta;
    long slot_rows = 0;
    for (int l = 0; l < nl; ++l) {
        const int lo = level_ptr[(size_t)l], hi = level_ptr[(size_t)l + 1];
        for (int r0 = lo; r0 < hi; r0 += R) {
            FlowChunk m;
            std::memset(&m, 0, sizeof(m));
            m.row0 = r0; m.nrows = std::min(R, hi - r0);
            m.nslots = (cnt[(size_t)ord[(size_t)r0]] + lpr - 1) / lpr;          // per lane; the chunk's first row is its longest
            m.lvl_lo = lo; m.lvl_hi = hi;
            if (slot_rows > 2000000000L) return 0;                              // FlowChunk::off is an int (entries are addressed in 64 bits)
            m.off = (int)slot_rows;
            slot_rows += m.nslots;
            meta.push_back(m);
        }
    }
    // (the 12 B per slot are written exactly once, padding included, by the thread that owns the chunk: no zero-fill pass)
    const size_t nslot = (size_t)std::max(slot_rows, 1L) * 64;
    std::unique_ptr<int[]> col(new int[nslot]);
    std::unique_ptr<double[]> val(new double[nslot]);
    std::vector<int> rmap((size_t)n), lev((size_t)n), gf((size_t)n, ncols), gb((size_t)n, ncols);
    std::vector<double> dgs((size_t)n);
    flow_parallel(nl, 64, [&](long llo, long lhi) {
        for (long l = llo; l < lhi; ++l)
            for (int k = level_ptr[(size_t)l]; k < level_ptr[(size_t)l + 1]; ++k) lev[(size_t)k] = (int)l;
    });
    flow_parallel((long)meta.size(), 256, [&](long clo, long chi) {
        for (long c = clo; c < chi; ++c) {
            const FlowChunk &m = meta[(size_t)c];
            for (size_t w = (size_t)m.off * 64; w < ((size_t)m.off + (size_t)m.nslots) * 64; ++w) { col[w] = ncols; val[w] = 0.0; }
            for (int t = 0; t < m.nrows; ++t) {
                const int kk = m.row0 + t;                                      // position in the form's numbering
                const int k = ord[(size_t)kk], i = rowmap[(size_t)k];
                rmap[(size_t)kk] = i;
                dgs[(size_t)kk] = dg[(size_t)k];
                const int lk = lev[(size_t)kk];
                int j = 0, gfl = -1, gbl = nl;                                  // off-diagonal entries in stored order
                for (int q = gp[(size_t)k]; q < gp[(size_t)k + 1]; ++q) {
                    const int cj = gj[(size_t)q];
                    if (cj == i) continue;
                    const int g = j / m.nslots, u = j % m.nslots;               // lane g of the row's group, its slot u
                    const size_t at = ((size_t)m.off + (size_t)u) * 64 + (size_t)t * lpr + (size_t)g;
                    if (cj >= n) {                                              // halo column: a frozen operand at its own position
                        col[at] = cj;
                        val[at] = gx[(size_t)q];
                        ++j;
                        continue;
                    }
                    const int pc = pos_of[(size_t)cj], lc = lev[(size_t)pc];
                    col[at] = pc;
                    val[at] = gx[(size_t)q];
                    if (lc <= lk - 2 && lc > gfl) { gfl = lc; gf[(size_t)kk] = pc; }  // forward gate: the latest level at least two back
                    if (lc >= lk + 2 && lc < gbl) { gbl = lc; gb[(size_t)kk] = pc; }  // backward gate
                    ++j;
                }
            }
        }
    });
    F.n = n; F.ncols = ncols; F.nchunks = (int)meta.size(); F.nlevels = nl; F.slot_rows = slot_rows;
    F.xstride = ((long)ncols + 1 + 15) / 16 * 16;
    long acct = 0;
    FCHK(falloc(&F.rowmap, n, &acct));
    FCHK(falloc(&F.meta, (long)meta.size(), &acct));
    FCHK(falloc(&F.col, slot_rows * 64, &acct));
    FCHK(falloc(&F.val, slot_rows * 64, &acct));
    FCHK(falloc(&F.diag, n, &acct));
    FCHK(falloc(&F.bp, n, &acct));
    FCHK(falloc(&F.gate_f, n, &acct));
    FCHK(falloc(&F.gate_b, n, &acct));
    FCHK(falloc(&F.X, (FLOW_MAXSEQ + 1) * F.xstride, &acct));
    AMG_HIP(hipMemcpy(F.rowmap, rmap.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.meta, meta.data(), sizeof(FlowChunk) * meta.size(), hipMemcpyHostToDevice));
    if (slot_rows > 0) {
        AMG_HIP(hipMemcpy(F.col, col.get(), sizeof(int) * (size_t)slot_rows * 64, hipMemcpyHostToDevice));
        AMG_HIP(hipMemcpy(F.val, val.get(), sizeof(double) * (size_t)slot_rows * 64, hipMemcpyHostToDevice));
    }
    AMG_HIP(hipMemcpy(F.diag, dgs.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.gate_f, gf.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    AMG_HIP(hipMemcpy(F.gate_b, gb.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    F.bytes = acct;
    F.ready = true;
    return 0;
}

// all directional sweeps of one smoother application (seq[k] != 0: backward), FLOW_MAXSEQ per launch
int gs_flow_sweep(const FlowForm &F, bool bsr1, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st)
{
    if (!F.ready) { set_error("dataflow Gauss-Seidel form was not built"); return -3; }
    int *status = nullptr;
    FCHK(flow_status_word(&status));
    for (int s0 = 0; s0 < nseq; s0 += FLOW_MAXSEQ) {
        const int ns = std::min(FLOW_MAXSEQ, nseq - s0);
        hipLaunchKernelGGL(flow_gather_kernel, dim3((unsigned)((F.ncols + 1 + 255) / 256)), dim3(256), 0, st, F.rowmap, x, b,
                           (unsigned long long *)F.X, F.bp, F.xstride, F.n, F.ncols, ns);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "dataflow gather launch", __FILE__, __LINE__);
        FlowArgs a;
        std::memset(&a, 0, sizeof(a));
        a.X = (unsigned long long *)F.X; a.x_out = x; a.status = status;
        a.xstride = F.xstride; a.budget = 100000000LL * 4;                   // 4 s of the 100 MHz wall clock
        a.nchunks = F.nchunks; a.nseq = ns; a.n = F.ncols;          // position of the permanent 0.0
        a.xcd = flow_xcd(false);
        a.dirmask = 0;
        for (int k = 0; k < ns; ++k) a.dirmask |= (seq[s0 + k] != 0 ? 1u : 0u) << k;
        switch (F.lpr) {
        case 1: FCHK(launch_flow<1>(F, bsr1, a, st)); break;
        case 2: FCHK(launch_flow<2>(F, bsr1, a, st)); break;
        case 4: FCHK(launch_flow<4>(F, bsr1, a, st)); break;
        case 8: FCHK(launch_flow<8>(F, bsr1, a, st)); break;
        case 16: FCHK(launch_flow<16>(F, bsr1, a, st)); break;
        case 32: FCHK(launch_flow<32>(F, bsr1, a, st)); break;
        default: FCHK(launch_flow<64>(F, bsr1, a, st)); break;
        }
    }
    return 0;
}

}  // namespace amg
