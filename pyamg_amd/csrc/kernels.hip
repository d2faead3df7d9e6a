// HIP kernels for gfx950 (MI355X): the solve-phase hot path of the AMG cycle.
//
// Everything here is HBM-bandwidth bound (~0.17 flop/byte): no MFMA.  The
// design goals are (1) every matrix byte is read once, fully coalesced, (2) the
// gathered vector is served from L2 / Infinity Cache, (3) per-row sums are
// accumulated strictly left to right with no FMA contraction (this file is
// compiled with -ffp-contract=off), so results are bit-identical to the
// reference's sequential C++ / scipy loops.
//
// csr_stream: a workgroup of 256 threads owns 256 consecutive rows.  The
// products a_ij * x_j of those rows form one contiguous slice of the CSR arrays;
// the workgroup streams that slice through an LDS tile with coalesced loads
// (lane k -> entry k), then thread t walks the products of row t in LDS in
// storage order.  Short rows (7-pt stencil) and long rows (30-60 nnz on coarse
// SA levels) go through the same path; a slice longer than the tile is
// processed in several passes with the running sums kept in registers.
#include "amg_dev.hpp"

namespace amg {

constexpr int WG = 256;      // threads = rows per workgroup
#ifndef AMG_TILE
#define AMG_TILE 2048
#endif
constexpr int TILE = AMG_TILE;   // products staged in LDS per pass (16 KiB at 2048)

// 16-byte loads of the matrix stream; AMG_NT_LOADS marks them non-temporal (streamed once: keep
// the gathered vector, not the matrix, in L2 / Infinity Cache)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4i load_v4i(const int *p)
{
#ifdef AMG_NT_LOADS
    return __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p));
#else
    return *reinterpret_cast<const v4i *>(p);
#endif
}
__device__ __forceinline__ v2d load_v2d(const double *p)
{
#ifdef AMG_NT_LOADS
    return __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
#else
    return *reinterpret_cast<const v2d *>(p);
#endif
}

static int g_stream_variant = 1;
static int g_xcd_chunk = 32;   // consecutive row blocks per XCD (PMC: -15 % L2-miss traffic, time -1..2 %)
static int g_tile_target = 2048;   // products per workgroup aimed at when choosing rows per workgroup
// every change of a launch knob bumps this; hierarchies drop their captured graphs when they see a new value
static int g_config_epoch = 0;
int config_epoch() { return g_config_epoch; }
void bump_config_epoch() { ++g_config_epoch; }
void set_tile_target(int t) { g_tile_target = t > 0 ? t : 2048; ++g_config_epoch; }

// Rows per workgroup for a matrix with `nnz` entries in `rows` rows: enough rows to fill about
// one LDS tile, so that long-row operators (restriction, coarse levels) still spread over many
// workgroups and need one staging pass per workgroup.
int rows_per_wg_for(long nnz, long rows)
{
    if (rows <= 0 || nnz <= 0) return WG;
    double avg = (double)nnz / (double)rows;
    int r = (int)(g_tile_target / (avg > 1.0 ? avg : 1.0));
    int p = 1;
    while (p * 2 <= r && p < WG) p *= 2;
    return p;
}
void set_stream_variant(int v) { g_stream_variant = v; ++g_config_epoch; }
int stream_variant() { return g_stream_variant; }
void set_xcd_chunk(int c) { g_xcd_chunk = c; ++g_config_epoch; }
static int g_xcd_period = 1;
void set_xcd_period(int on) { g_xcd_period = on; ++g_config_epoch; }

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).
// With chunk > 0, each group of 8*chunk consecutive logical blocks is laid out
// so that one XCD sweeps `chunk` consecutive row blocks: neighbouring rows (and
// the x entries they gather) then live in one L2.  Speed only, never correctness.
__device__ __forceinline__ int remap_block(int b, int nb, int chunk)
{
    if (chunk <= 0) return b;
    int gsz = 8 * chunk;
    int g = b / gsz;
    int base = g * gsz;
    int n_g = min(gsz, nb - base);
    int l = b - base;
    int xcd = l & 7, j = l >> 3;
    int q = n_g >> 3, r = n_g & 7;
    int start = xcd * q + min(xcd, r);
    return base + start + j;
}

__device__ __forceinline__ double block_reduce_sum(double v, double *smem)
{
    // wave64 butterfly, then 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) smem[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        int nw = blockDim.x >> 6;
        for (int k = 0; k < nw; ++k) r += smem[k];
    }
    return r;
}

// epilogue stores of the result vectors (streamed once per launch)
__device__ __forceinline__ void store_out(double *p, double v)
{
#ifndef AMG_PLAIN_STORES
    __builtin_nontemporal_store(v, p);      // measured -2..3 % launch time on the 500^3 level vs a plain store
#else
    *p = v;
#endif
}

// x entries another wave of the SAME workgroup may have rewritten since this CU last cached them: an agent-scope
// relaxed load (global_load ... sc1) is served from L2, where the producer's store has landed before the barrier
// that separates the two (s_waitcnt vmcnt(0) precedes s_barrier).  NOT `volatile`: the compiler follows every
// volatile access with s_waitcnt vmcnt(0), which turns a batch of independent gathers into a chain of round trips
// (measured: ~3 us per dependency level in the chained sweeps, whatever else they did).
__device__ __forceinline__ double load_fresh(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE> struct ModeTraits {
    static constexpr bool jac = (MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1);
    static constexpr bool gs = (MODE == SM_GS || MODE == SM_GS_BSR1);
    static constexpr bool sub = (MODE == SM_JACOBI_BSR1 || MODE == SM_GS_BSR1);
};

// (A/B builds: -DAMG_ROWSUM_BATCH=4, -DAMG_STREAM_MIN_WAVES=8; tools/lib_ab.py)
#ifndef AMG_ROWSUM_BATCH
#define AMG_ROWSUM_BATCH 8
#endif
#ifdef AMG_STREAM_MIN_WAVES
#define AMG_STREAM_BOUNDS __launch_bounds__(WG, AMG_STREAM_MIN_WAVES)
#else
#define AMG_STREAM_BOUNDS __launch_bounds__(WG)
#endif
template <int MODE, int VEC>
__global__ AMG_STREAM_BOUNDS void csr_stream_kernel(StreamArgs a, int xcd_chunk, int rpb)
{
    const long nnz_total = a.nnz_total;
    const double gscale = a.gscale;      // 1.0 unless the operand is scaled on the fly (1.0*x is exact)
    using MT = ModeTraits<MODE>;
    __shared__ double sp[TILE];
    __shared__ int sAp[WG + 1];
    __shared__ double sdiag[MT::jac ? WG : 1];

    const int t = threadIdx.x;
    const int blk = remap_block(blockIdx.x, gridDim.x, xcd_chunk);
    const int r0 = a.row_lo + blk * rpb;          // rpb rows per workgroup (<= WG)
    const int nr = min(rpb, a.row_hi - r0);

    for (int i = t; i <= nr; i += WG) sAp[i] = a.Ap[r0 + i];
    if (MT::jac) sdiag[t] = 0.0;
    // 16-bit column codes: the launcher passes Aj16 only when this launch's row blocks are the
    // ones the coding was built for (r0 is then a multiple of rpb)
    __shared__ int sbase[16];
    bool use16 = false;
    if (VEC && a.Aj16) {
        const int wg = r0 / rpb;
        use16 = a.wg_flag[wg] != 0;
        if (use16 && t < 16) sbase[t] = a.wg_base[wg * 16 + t];
    }
    __syncthreads();

    const int kbeg = sAp[0], kend = sAp[nr];
    const int my_s = (t < nr) ? sAp[t] : kend;
    const int my_e = (t < nr) ? sAp[t + 1] : kend;

    int row = r0 + t;            // index into b / out (original numbering)
    int dpos = -1;
    if (MT::gs && t < nr) {
        if (a.rowmap) row = a.rowmap[r0 + t];
        dpos = a.diagpos[r0 + t];
    }
    double acc = 0.0;
    if (MT::sub && t < nr) acc = a.b[row];
    // Gauss-Seidel levels are latency-bound (a few thousand rows per launch): fetch the diagonal
    // and the right-hand side now, so that their latency overlaps the matrix stream instead of
    // following it
    double gs_d = 0.0, gs_b = 0.0;
    if (MT::gs && t < nr) {
        gs_d = (dpos >= 0) ? a.Ax[dpos] : 0.0;
        if (!MT::sub) gs_b = a.b[row];
    }

    // the epilogue's streamed operands do not depend on the row sums: request them now, so that their latency
    // overlaps the matrix stream instead of following it (one dependent round trip less per workgroup)
    double ep_b = 0.0, ep_v2 = 0.0;
    if (!MT::gs && t < nr) {
        if (MODE == SM_RESIDUAL || MODE == SM_RESIDUAL_SUMSQ || MODE == SM_POLY_FIRST || MODE == SM_POLY_STEP ||
            MODE == SM_POLY_LAST || MODE == SM_JACOBI)
            ep_b = a.b[r0 + t];
        if (MODE == SM_POLY_LAST || MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1) ep_v2 = a.v2[r0 + t];
        if (MODE == SM_MATVEC_ACC) ep_v2 = a.out[r0 + t];
    }

    // tile origin aligned to 4 entries so that 16-byte loads are aligned
    const int abeg = VEC ? (kbeg & ~3) : kbeg;

    for (int tile_lo = abeg; tile_lo < kend; tile_lo += TILE) {
        const int tile_hi = min(tile_lo + TILE, kend);
        if (VEC) {
            // each thread: NQ quads of 4 consecutive entries
            constexpr int NQ = TILE / (4 * WG);
            int e[NQ];
            v4i cj[NQ];
            v2d av[NQ][2];
            bool full[NQ], any[NQ];
#pragma unroll
            for (int p = 0; p < NQ; ++p) {
                e[p] = tile_lo + p * (4 * WG) + 4 * t;
                any[p] = e[p] < tile_hi;
                full[p] = any[p] && ((long)e[p] + 4 <= nnz_total);
                if (full[p]) {
                    if (use16) {
                        // 4 codes in 8 bytes, decoded after all loads of the tile have been issued
                        const uint2 cc = *reinterpret_cast<const uint2 *>(a.Aj16 + e[p]);
                        cj[p] = v4i{(int)cc.x, (int)cc.y, 0, 0};
                    } else {
                        cj[p] = load_v4i(a.Aj + e[p]);
                    }
                    av[p][0] = load_v2d(a.Ax + e[p]);
                    av[p][1] = load_v2d(a.Ax + e[p] + 2);
                } else if (any[p]) {
                    int c[4]; double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bool ok = (long)e[p] + u < nnz_total;
                        c[u] = ok ? a.Aj[e[p] + u] : 0;            // the last quad of the array: plain indices
                        v[u] = ok ? a.Ax[e[p] + u] : 0.0;
                    }
                    cj[p] = v4i{c[0], c[1], c[2], c[3]};
                    av[p][0] = v2d{v[0], v[1]};
                    av[p][1] = v2d{v[2], v[3]};
                }
            }
#pragma unroll
            for (int p = 0; p < NQ; ++p) {
                if (!any[p]) continue;
                int c[4] = {cj[p].x, cj[p].y, cj[p].z, cj[p].w};
                if (use16 && full[p]) {
                    // entries of the neighbouring row blocks that share the first / last quad carry
                    // THEIR blocks' codes: give them column 0 (their products are never summed)
                    const unsigned cx = (unsigned)cj[p].x, cy = (unsigned)cj[p].y;
                    const unsigned code[4] = {cx & 0xFFFFu, cx >> 16, cy & 0xFFFFu, cy >> 16};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = e[p] + u;
                        c[u] = (k >= kbeg && k < kend) ? sbase[code[u] >> 12] + (int)(code[u] & 4095u) : 0;
                    }
                }
                double v[4] = {av[p][0].x, av[p][0].y, av[p][1].x, av[p][1].y};
                double xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = gscale * a.xg[c[u]];
                double pr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pr[u] = v[u] * xv[u];
                    if (MT::jac) {
                        int k = e[p] + u;
                        int lc = c[u] - r0;
                        if ((unsigned)lc < (unsigned)nr && k >= sAp[lc] && k < sAp[lc + 1]) {
                            sdiag[lc] = v[u];
                            pr[u] = 0.0;
                        }
                    }
                }
                int q = e[p] - tile_lo;
                *reinterpret_cast<double2 *>(&sp[q]) = make_double2(pr[0], pr[1]);
                *reinterpret_cast<double2 *>(&sp[q + 2]) = make_double2(pr[2], pr[3]);
            }
        } else {
            constexpr int U = TILE / WG;   // 8 entries per thread, lane-contiguous
            int c[U];
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int k = tile_lo + u * WG + t;
                bool ok = k < tile_hi;
                c[u] = ok ? a.Aj[k] : 0;
                v[u] = ok ? a.Ax[k] : 0.0;
            }
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int k = tile_lo + u * WG + t;
                xv[u] = (k < tile_hi) ? gscale * a.xg[c[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int k = tile_lo + u * WG + t;
                if (k < tile_hi) {
                    double pr = v[u] * xv[u];
                    if (MT::jac) {
                        int lc = c[u] - r0;
                        if ((unsigned)lc < (unsigned)nr && k >= sAp[lc] && k < sAp[lc + 1]) {
                            sdiag[lc] = v[u];
                            pr = 0.0;
                        }
                    }
                    sp[u * WG + t] = pr;
                }
            }
        }
        __syncthreads();
        {
            // the row's products, eight LDS reads in flight at a time (one read, one wait, one add per entry -- what
            // the plain loop compiles to -- made this phase ~1.5 us of a workgroup's ~8 us on 27-entry rows), added
            // strictly left to right
            const int s = max(my_s, tile_lo), e2 = min(my_e, tile_hi);
            constexpr int RB = AMG_ROWSUM_BATCH;
            for (int k = s; k < e2; k += RB) {
                double p[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) p[u] = sp[min(k + u, e2 - 1) - tile_lo];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const bool take = (k + u < e2) && !(MT::gs && k + u == dpos);
                    const double nxt = MT::sub ? (acc - p[u]) : (acc + p[u]);
                    acc = take ? nxt : acc;
                }
            }
        }
        __syncthreads();
    }

    if (MODE == SM_RESIDUAL_SUMSQ) {
        // per-workgroup partial of ||b - A x||^2 (fixed order: lanes by shuffle tree, waves in order)
        double sq = 0.0;
        if (t < nr) {
            double rr = ep_b - acc;
            sq = rr * rr;
            if (a.out) store_out(&a.out[r0 + t], rr);     // kept for the next pre-smoother (hier.hip)
        }
        __syncthreads();
        double tot = block_reduce_sum(sq, sp);
        if (t == 0) a.out2[blk] = tot;
        return;
    }
    if (t >= nr) return;
    const int i = r0 + t;
    if (MODE == SM_MATVEC) {
        store_out(&a.out[i], acc);
    } else if (MODE == SM_MATVEC_ACC) {
        store_out(&a.out[i], ep_v2 + acc);
    } else if (MODE == SM_RESIDUAL) {
        store_out(&a.out[i], ep_b - acc);
    } else if (MODE == SM_POLY_FIRST) {
        double r = ep_b - acc;
        a.out[i] = r;
        a.out2[i] = a.c0 * r;
    } else if (MODE == SM_POLY_STEP) {
        double cr = a.c0 * ep_b;
        a.out[i] = cr + acc;
    } else if (MODE == SM_POLY_LAST) {
        double cr = a.c0 * ep_b;
        double h = cr + acc;
        store_out(&a.out[i], ep_v2 + h);
    } else if (MODE == SM_JACOBI) {
        double d = sdiag[t];
        double told = ep_v2;
        if (d != 0.0) {
            double q = (ep_b - acc) / d;
            double t1 = (1.0 - a.c0) * told;
            double t2 = a.c0 * q;
            a.out[i] = t1 + t2;
        } else {
            a.out[i] = told;
        }
    } else if (MODE == SM_JACOBI_BSR1) {
        double d = sdiag[t];
        double told = ep_v2;
        if (d != 0.0) {
            double t1 = (1.0 - a.c0) * told;
            double t2 = (a.c0 * acc) / d;
            a.out[i] = t1 + t2;
        } else {
            a.out[i] = told;
        }
    } else if (MODE == SM_GS) {
        if (gs_d != 0.0) a.out[row] = (gs_b - acc) / gs_d;
    } else if (MODE == SM_GS_BSR1) {
        if (gs_d != 0.0) a.out[row] = acc / gs_d;
    }
}

// ---------------------------------------------------------------------------
// csr_stream_pipe: the same row-block arithmetic as csr_stream_kernel<MODE, 1>, software-pipelined across the row
// blocks a PERSISTENT workgroup sweeps.  The plain kernel makes three dependent memory round trips per workgroup
// (row pointers -> entries -> gathered operands) and then a fourth for the epilogue operands; with ~5 workgroups
// resident per CU that leaves too few bytes in flight to fill HBM (measured 0.62-0.71 of the 8 TB/s spec on the
// irregular operators).  Here a workgroup walks blocks w, w + G, w + 2G, ... and keeps three of them in flight:
//   block k+2 : its first / last entry positions (two scalar loads)
//   block k+1 : its row pointers, the entries of its first tile and its epilogue operands -- requested right after
//               block k's gathers were issued (vector loads return in issue order: the gathers come back first)
//   block k   : gathers, products to LDS, row sums, epilogue
// so that what a block waits for was requested one block earlier.  Rows are summed exactly as before (LDS tile,
// strict left-to-right, separate multiply and add): results are bit-identical to csr_stream_kernel.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(WG) void csr_stream_pipe_kernel(StreamArgs a, int xcd_chunk, int rpb, int nb)
{
    using MT = ModeTraits<MODE>;
    static_assert(!MT::gs, "Gauss-Seidel levels keep the plain kernel");
    __shared__ double sp[TILE];
    __shared__ int sAp[WG + 1];
    __shared__ double sdiag[MT::jac ? WG : 1];
    constexpr int NQ = TILE / (4 * WG);
    constexpr bool need_b = MT::sub || MODE == SM_RESIDUAL || MODE == SM_RESIDUAL_SUMSQ || MODE == SM_POLY_FIRST ||
                            MODE == SM_POLY_STEP || MODE == SM_POLY_LAST || MODE == SM_JACOBI;
    constexpr bool need_v2 = MODE == SM_POLY_LAST || MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1;
    const int t = threadIdx.x;
    const int G = gridDim.x;
    const long nnz_total = a.nnz_total;
    const double gscale = a.gscale;

    struct Span { int r0, nr, kb, ke; };
    auto span_of = [&](int i) -> Span {                     // uniform: scalar loads
        Span s{0, 0, 0, 0};
        if (i < nb) {
            const int blk = remap_block(i, nb, xcd_chunk);
            s.r0 = a.row_lo + blk * rpb;
            s.nr = min(rpb, a.row_hi - s.r0);
            s.kb = a.Ap[s.r0];
            s.ke = a.Ap[s.r0 + s.nr];
        }
        return s;
    };
    struct Pre {
        int rp;                       // Ap[r0 + t]
        v4i cj[NQ];
        v2d av[NQ][2];
        double b, v2;
    };
    auto prefetch = [&](const Span &s, Pre &p) {
        p.rp = (t <= s.nr && s.nr > 0) ? a.Ap[s.r0 + t] : s.ke;      // rpb <= 256: thread nr holds the end pointer when nr < 256
        const int tile_lo = s.kb & ~3;
        const int tile_hi = min(tile_lo + TILE, s.ke);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tile_lo + q * (4 * WG) + 4 * t;
            p.cj[q] = v4i{0, 0, 0, 0};
            p.av[q][0] = v2d{0.0, 0.0};
            p.av[q][1] = v2d{0.0, 0.0};
            if (e < tile_hi) {
                if ((long)e + 4 <= nnz_total) {
                    p.cj[q] = load_v4i(a.Aj + e);
                    p.av[q][0] = load_v2d(a.Ax + e);
                    p.av[q][1] = load_v2d(a.Ax + e + 2);
                } else {
                    int c[4]; double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = (long)e + u < nnz_total;
                        c[u] = ok ? a.Aj[e + u] : 0;
                        v[u] = ok ? a.Ax[e + u] : 0.0;
                    }
                    p.cj[q] = v4i{c[0], c[1], c[2], c[3]};
                    p.av[q][0] = v2d{v[0], v[1]};
                    p.av[q][1] = v2d{v[2], v[3]};
                }
            }
        }
        p.b = 0.0; p.v2 = 0.0;
        if (t < s.nr) {
            if (need_b) p.b = a.b[s.r0 + t];
            if (need_v2) p.v2 = a.v2[s.r0 + t];
            if (MODE == SM_MATVEC_ACC) p.v2 = a.out[s.r0 + t];
        }
    };

    Span cur = span_of(blockIdx.x), nxt = span_of(blockIdx.x + G);
    Pre pc, pn;
    prefetch(cur, pc);
    for (int i = blockIdx.x; i < nb; i += G) {
        const Span aft = span_of(i + 2 * G);
        const int r0 = cur.r0, nr = cur.nr, kbeg = cur.kb, kend = cur.ke;
        // row pointers of this block from registers to LDS (entry 256 of a full block is the end pointer)
        if (t <= nr) sAp[t] = pc.rp;
        if (t == 0) sAp[nr] = kend;
        if (MT::jac) sdiag[t] = 0.0;
        __syncthreads();
        const int my_s = (t < nr) ? sAp[t] : kend;
        const int my_e = (t < nr) ? sAp[t + 1] : kend;
        double acc = MT::sub ? pc.b : 0.0;
        if (t >= nr) acc = 0.0;
        const int abeg = kbeg & ~3;
        bool first = true;
        for (int tile_lo = abeg; tile_lo < kend; tile_lo += TILE) {
            const int tile_hi = min(tile_lo + TILE, kend);
            int e[NQ];
            bool any[NQ];
            v4i cj[NQ];
            v2d av[NQ][2];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                e[q] = tile_lo + q * (4 * WG) + 4 * t;
                any[q] = e[q] < tile_hi;
            }
            if (first) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { cj[q] = pc.cj[q]; av[q][0] = pc.av[q][0]; av[q][1] = pc.av[q][1]; }
            } else {
                // rows longer than one tile: the following tiles are loaded in place (not prefetched)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    cj[q] = v4i{0, 0, 0, 0}; av[q][0] = v2d{0.0, 0.0}; av[q][1] = v2d{0.0, 0.0};
                    if (any[q]) {
                        if ((long)e[q] + 4 <= nnz_total) {
                            cj[q] = load_v4i(a.Aj + e[q]);
                            av[q][0] = load_v2d(a.Ax + e[q]);
                            av[q][1] = load_v2d(a.Ax + e[q] + 2);
                        } else {
                            int c[4]; double v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const bool ok = (long)e[q] + u < nnz_total;
                                c[u] = ok ? a.Aj[e[q] + u] : 0;
                                v[u] = ok ? a.Ax[e[q] + u] : 0.0;
                            }
                            cj[q] = v4i{c[0], c[1], c[2], c[3]};
                            av[q][0] = v2d{v[0], v[1]};
                            av[q][1] = v2d{v[2], v[3]};
                        }
                    }
                }
            }
            // 1. this tile's operands
            double xv[NQ][4];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c[4] = {cj[q].x, cj[q].y, cj[q].z, cj[q].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[q][u] = any[q] ? a.xg[c[u]] : 0.0;
            }
            // 2. the NEXT block's row pointers, first tile and epilogue operands (independent of everything above)
            if (first) prefetch(nxt, pn);
            // 3. products into the LDS tile
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!any[q]) continue;
                const int c[4] = {cj[q].x, cj[q].y, cj[q].z, cj[q].w};
                const double v[4] = {av[q][0].x, av[q][0].y, av[q][1].x, av[q][1].y};
                double pr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pr[u] = v[u] * (gscale * xv[q][u]);
                    if (MT::jac) {
                        const int k = e[q] + u;
                        const int lc = c[u] - r0;
                        if ((unsigned)lc < (unsigned)nr && k >= sAp[lc] && k < sAp[lc + 1]) {
                            sdiag[lc] = v[u];
                            pr[u] = 0.0;
                        }
                    }
                }
                const int qq = e[q] - tile_lo;
                *reinterpret_cast<double2 *>(&sp[qq]) = make_double2(pr[0], pr[1]);
                *reinterpret_cast<double2 *>(&sp[qq + 2]) = make_double2(pr[2], pr[3]);
            }
            __syncthreads();
            {
                const int s = max(my_s, tile_lo), e2 = min(my_e, tile_hi);
                for (int k = s; k < e2; ++k) {
                    const double p = sp[k - tile_lo];
                    acc = MT::sub ? (acc - p) : (acc + p);
                }
            }
            __syncthreads();
            first = false;
        }
        if (kend <= abeg) prefetch(nxt, pn);       // an empty block issued no tile: still feed the pipeline

        // epilogue (operands were requested one block ago)
        if (MODE == SM_RESIDUAL_SUMSQ) {
            double sq = 0.0;
            if (t < nr) {
                const double rr = pc.b - acc;
                sq = rr * rr;
                if (a.out) store_out(&a.out[r0 + t], rr);
            }
            const double tot = block_reduce_sum(sq, sp);
            if (t == 0) a.out2[remap_block(i, nb, xcd_chunk)] = tot;
            __syncthreads();
        } else if (t < nr) {
            const int row = r0 + t;
            if (MODE == SM_MATVEC) {
                store_out(&a.out[row], acc);
            } else if (MODE == SM_MATVEC_ACC) {
                store_out(&a.out[row], pc.v2 + acc);
            } else if (MODE == SM_RESIDUAL) {
                store_out(&a.out[row], pc.b - acc);
            } else if (MODE == SM_POLY_FIRST) {
                const double r = pc.b - acc;
                a.out[row] = r;
                a.out2[row] = a.c0 * r;
            } else if (MODE == SM_POLY_STEP) {
                const double cr = a.c0 * pc.b;
                a.out[row] = cr + acc;
            } else if (MODE == SM_POLY_LAST) {
                const double cr = a.c0 * pc.b;
                const double h = cr + acc;
                store_out(&a.out[row], pc.v2 + h);
            } else if (MODE == SM_JACOBI) {
                const double d = sdiag[t];
                if (d != 0.0) {
                    const double q = (pc.b - acc) / d;
                    const double t1 = (1.0 - a.c0) * pc.v2;
                    const double t2 = a.c0 * q;
                    a.out[row] = t1 + t2;
                } else {
                    a.out[row] = pc.v2;
                }
            } else if (MODE == SM_JACOBI_BSR1) {
                const double d = sdiag[t];
                if (d != 0.0) {
                    const double t1 = (1.0 - a.c0) * pc.v2;
                    const double t2 = (a.c0 * acc) / d;
                    a.out[row] = t1 + t2;
                } else {
                    a.out[row] = pc.v2;
                }
            }
        }
        if (MT::jac) __syncthreads();               // sdiag is rewritten by the next block
        cur = nxt; nxt = aft; pc = pn;
    }
}

static int g_stream_pipe = 0;   // measured slower than one row block per workgroup: opt-in
void set_stream_pipe(int on) { g_stream_pipe = on; ++g_config_epoch; }
bool stream_pipe_enabled() { return g_stream_pipe != 0; }

// ---------------------------------------------------------------------------
// Chained Gauss-Seidel sweep over a run of NARROW dependency levels (each at most CHAIN_WG rows): one
// workgroup, one launch.  Thread t owns row t of the current level; a barrier separates the levels.
// What makes a level cheap here is that everything that does not depend on x -- the row's entries, its
// diagonal and right-hand side -- is requested for the NEXT level before the current one is computed,
// so the per-level critical path is barrier -> gather x (L2) -> multiply-adds -> store, not the three
// dependent memory round trips of a fresh launch.  Same per-row arithmetic as SM_GS / SM_GS_BSR1.
// ---------------------------------------------------------------------------
constexpr int CHAIN_WG = 512;         // widest level a chain takes; the launch picks 64 .. 512 threads by the run's widest level
constexpr int CHAIN_PF = 12;         // entries of a row carried in registers through the pipeline
constexpr int CHAIN_LMAX = 4096;     // levels per launch (their offsets sit in LDS)

// Software pipeline over the levels (vector-memory loads return in issue order, so what a level waits
// for must be issued BEFORE anything it does not need yet, and nothing issued late may be waited for in
// the same iteration):
//   level q   : gather x for the row whose entries are already in registers, compute, store, barrier
//   level q+1 : its entries / right-hand side / diagonal are requested after q's gathers were issued,
//               from row pointers that were requested two iterations ago
//   level q+3 : its row pointers (and row / diagonal positions) are requested; the level offsets they
//               need come from LDS (loaded once), not from a dependent global load
template <bool BSR1, int WGS>
__global__ __launch_bounds__(WGS) void gs_chain_kernel(const int *Ap, const int *Aj, const double *Ax, const int *rowmap,
                                                           const int *diagpos, double *x, const double *b,
                                                           const int *lp, int l_first, int nl, int reverse)
{
    const int t = threadIdx.x;
    __shared__ int slp[CHAIN_LMAX + 1];
    for (int k = t; k <= nl; k += WGS) slp[k] = lp[l_first + k];
    __syncthreads();
    auto level_of = [&](int q) { return reverse ? nl - 1 - q : q; };      // index into slp

    // stage A (row pointers) of a level
    struct RowRef { int s, e, row, dp; };
    auto stage_a = [&](int q) -> RowRef {
        RowRef r{0, 0, -1, -1};
        if (q < nl) {
            const int l = level_of(q);
            const int p = slp[l] + t;
            if (p < slp[l + 1]) { r.s = Ap[p]; r.e = Ap[p + 1]; r.row = rowmap ? rowmap[p] : p; r.dp = diagpos[p]; }
        }
        return r;
    };

    RowRef cur = stage_a(0);
    int cc[CHAIN_PF];
    double cv[CHAIN_PF];
    double c_b = 0.0, c_d = 0.0;
    if (cur.row >= 0) {
        c_b = b[cur.row];
        c_d = (cur.dp >= 0) ? Ax[cur.dp] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < CHAIN_PF; ++u) {
        const int k = cur.s + u;
        cc[u] = (k < cur.e) ? Aj[k] : 0;
        cv[u] = (k < cur.e) ? Ax[k] : 0.0;
    }
    RowRef nxt = stage_a(1);
    RowRef nx2 = stage_a(2);

    for (int q = 0; q < nl; ++q) {
        // 1. operands of the current level first
        double xv[CHAIN_PF];
#pragma unroll
        for (int u = 0; u < CHAIN_PF; ++u) xv[u] = (cur.s + u < cur.e) ? load_fresh(&x[cc[u]]) : 0.0;
        // 2. entries of the next level, 3. row pointers of the one after (neither depends on x)
        int pc[CHAIN_PF];
        double pv[CHAIN_PF];
        double n_b = 0.0, n_d = 0.0;
        if (nxt.row >= 0) {
            n_b = b[nxt.row];
            n_d = (nxt.dp >= 0) ? Ax[nxt.dp] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CHAIN_PF; ++u) {
            const int k = nxt.s + u;
            pc[u] = (k < nxt.e) ? Aj[k] : 0;
            pv[u] = (k < nxt.e) ? Ax[k] : 0.0;
        }
        const RowRef aft = stage_a(q + 3);
        // 4. the row sum in stored order, diagonal skipped
        if (cur.row >= 0) {
            double acc = BSR1 ? c_b : 0.0;
#pragma unroll
            for (int u = 0; u < CHAIN_PF; ++u) {
                const int k = cur.s + u;
                if (k < cur.e && k != cur.dp) {
                    const double pr = cv[u] * xv[u];
                    acc = BSR1 ? (acc - pr) : (acc + pr);
                }
            }
            for (int k = cur.s + CHAIN_PF; k < cur.e; ++k) {        // rows longer than the register window
                if (k == cur.dp) continue;
                const double pr = Ax[k] * load_fresh(&x[Aj[k]]);
                acc = BSR1 ? (acc - pr) : (acc + pr);
            }
            if (c_d != 0.0) x[cur.row] = BSR1 ? (acc / c_d) : ((c_b - acc) / c_d);
        }
        // One workgroup = one CU: its waves share the L1, so workgroup-scope ordering (what __syncthreads
        // provides: stores complete, then the barrier) is all the next level needs -- no cache maintenance
        __syncthreads();
        cur = nxt; c_b = n_b; c_d = n_d;
#pragma unroll
        for (int u = 0; u < CHAIN_PF; ++u) { cc[u] = pc[u]; cv[u] = pv[u]; }
        nxt = nx2;
        nx2 = aft;
    }
}

// ---------------------------------------------------------------------------
// gs_chain2: a run of narrow dependency levels swept by ONE workgroup with the new values handed from level to
// level through LDS.  In gs_chain_kernel a level costs ~3 us: its operands are stored to L2 by the previous level
// and gathered back after the barrier.  Here thread t of level q
//   * reads the operands produced by levels q-1 .. q-D from an LDS ring (D + 1 buffers of 512 doubles, written by
//     their rows right after the global store),
//   * finds everything else already in registers: entry values, diagonal, right-hand side and the operands that were
//     final in memory >= D + 1 levels ago were requested D levels ahead, the entry codes they depend on D + 1 levels
//     ahead (dependent loads never sit in one stage: a wave issues in order and would stall on them),
// so the per-level critical path is barrier -> LDS reads -> the row sum -> divide -> store, a few hundred cycles.
// Row sums run over the stored off-diagonal entries in stored order with separate multiply and add: bit-identical
// to relaxation.h:34-62 / :90-173 (bs = 1) and to the level-per-launch path.
// ---------------------------------------------------------------------------
template <bool BSR1, int PF, int WGS>
__global__ __launch_bounds__(WGS) void gs_chain2_kernel(const int *lp, const double *cval, const int *ccode, const int *coff,
                                                       double *x, const double2 *bd, double *dummy, int nzero, int l_first, int nl, int reverse)
{
    constexpr int D = CHAIN2_D, NB = CHAIN2_D + 1;
    // Vector-memory results return in issue order, so whatever an iteration needs must have been requested at least
    // TWO iterations earlier -- then the requests of the iteration in between may still be in flight while it runs:
    //   codes (operand addresses) of level q + D + A are requested in iteration q        (A = 2)
    //   values / diagonal / right-hand side / settled operands of level q + D, which need those codes, in iteration q
    //   level q is computed in iteration q from what was requested in iteration q - D   (D = 2)
    // Every request is UNCONDITIONAL (idle lanes and the steps past the last level read harmless addresses): a load
    // inside a branch makes the number of outstanding requests path-dependent and the compiler then waits for all.
    constexpr int A = 2;
    // x and the (right-hand side, diagonal) pairs are numbered in LEVEL ORDER (the sweep runs on the gathered copies
    // Schedule::xp / bd): the rows of a level are lp[l] .. lp[l+1]-1, so they and the store are contiguous over the
    // lanes, and an operand from a neighbouring level sits at a neighbouring position.  (Addressed by original row
    // these were one cache line per lane and per access: the vector L1's tag rate set the cost of a level.)
    // What a level costs now is the NUMBER of vector-memory instructions its waves issue (rocprofv3: ~16 cycles of the
    // compute unit each, whatever their width), then the rest of the instruction stream.  So: a row's codes come four
    // per 16-byte load and its values two per load (the copy keeps them side by side), right-hand side and diagonal in
    // one load; addresses are 32-bit byte offsets from uniform bases; a slot's code is either the operand's position
    // (>= 0; padded slots point at the permanent 0.0 behind the last unknown and carry the value 0, which leaves the
    // running sum untouched bit for bit) or ~(byte offset into the LDS ring) (< 0).
    static_assert(PF % 4 == 0, "codes are fetched four at a time, values two at a time");
    __shared__ int soff[CHAIN2_LMAX + 1];
    __shared__ int slp[CHAIN2_LMAX + 1];
    __shared__ double ring[NB * CHAIN2_WG + 1];               // + 1: what a slot without a ring operand reads
    constexpr unsigned RING_BYTES = NB * CHAIN2_WG * 8u;
    constexpr bool KEEP_LA = !(PF == 12 && WGS == 512);       // (that variant has no registers to spare for the ring addresses)
    const int t = threadIdx.x;
    for (int k = t; k <= nl; k += WGS) { soff[k] = coff[l_first + k]; slp[k] = lp[l_first + k]; }
    if (t == 0) ring[NB * CHAIN2_WG] = 0.0;
    __syncthreads();

    auto at_bytes = [](const void *base, unsigned off) { return (const void *)((const char *)base + (size_t)off); };
    struct Codes { unsigned at; int arow; int n; bool live; int code[PF]; };          // at = position in the copy (clamped lane)
    struct Stage { int row; unsigned arow8; int lvl; double d, bb; int code[PF]; unsigned la[PF]; double val[PF]; double xv[PF]; };
    auto level_of = [&](int q) { return reverse ? nl - 1 - q : q; };
    auto stage_a = [&](int q) -> Codes {           // row position and operand codes
        Codes c;
        const int l = level_of(min(q, nl - 1));
        const int base = soff[l];
        c.n = soff[l + 1] - base;                  // >= 1: dependency levels are never empty
        const int tt = min(t, c.n - 1);
        c.live = (q < nl) && (t < c.n);
        c.at = (unsigned)(base + tt);
        c.arow = slp[l] + tt;
        const unsigned step = (unsigned)c.n * 16u;
        unsigned off = ((unsigned)PF * (unsigned)base + 4u * (unsigned)tt) * 4u;      // quad k of this row: + k * n * 16
#pragma unroll
        for (int k = 0; k < PF / 4; ++k) {
            const int4 v = *(const int4 *)at_bytes(ccode, off); off += step;
            c.code[4 * k] = v.x; c.code[4 * k + 1] = v.y; c.code[4 * k + 2] = v.z; c.code[4 * k + 3] = v.w;
        }
        return c;
    };
    auto stage_b = [&](int q, const Codes &c) -> Stage {   // values, diagonal, right-hand side, settled operands
        Stage s;
        s.row = c.live ? c.arow : -1;
        s.arow8 = (unsigned)c.arow * 8u;
        s.lvl = l_first + level_of(min(q, nl - 1));
        const double2 rhs_diag = *(const double2 *)at_bytes(bd, (unsigned)c.arow * 16u);
        s.bb = rhs_diag.x; s.d = rhs_diag.y;
        const int tt = min(t, c.n - 1);
        const unsigned step = (unsigned)c.n * 16u;
        unsigned off = ((unsigned)PF * (c.at - (unsigned)tt) + 2u * (unsigned)tt) * 8u;   // pair k of this row: + k * n * 16
#pragma unroll
        for (int k = 0; k < PF / 2; ++k) {
            const double2 v = *(const double2 *)at_bytes(cval, off); off += step;
            s.val[2 * k] = v.x; s.val[2 * k + 1] = v.y;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            s.code[u] = c.code[u];
            if (KEEP_LA) s.la[u] = min(~(unsigned)c.code[u], RING_BYTES);     // where in the ring (the spare word if not there), off the critical path
            // a ring operand's slot requests the permanent 0.0 behind the last unknown (one line for the whole wave):
            // the operand is then `ring value | memory value`, one of the two being all zero bits
            s.xv[u] = load_fresh((const double *)at_bytes(x, min((unsigned)c.code[u], (unsigned)nzero) * 8u));
        }
        return s;
    };

    // Rotating buffers WITHOUT register moves (a move of a register a load is still filling would wait for that
    // load): level L's stage lives in sb[L % 3], its codes in cq[L % 3], and the loop body is written out for the
    // three residues so that every index is a compile-time constant.
    static_assert(D == 2 && A == 2, "the rotation below is written for D = A = 2");
    Stage sb[3];
    Codes cq[3];
    cq[0] = stage_a(0);
    cq[1] = stage_a(1);
    sb[0] = stage_b(0, cq[0]);
    sb[1] = stage_b(1, cq[1]);
    cq[2] = stage_a(2);
    cq[0] = stage_a(3);

    auto compute = [&](Stage &cur) {
        // operands produced by the last D levels come from the LDS ring (every lane reads: a slot without one reads
        // the spare word behind the ring and drops it); no branch and no memory request in here besides the one store
        // (all ring reads are issued back to back and waited for once: one LDS latency per level, not one per slot)
        double lv[PF], xo[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) lv[u] = *(const double *)((const char *)ring + (KEEP_LA ? cur.la[u] : min(~(unsigned)cur.code[u], RING_BYTES)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            // exactly one of the two is the operand, the other is +0.0 (the spare word behind the ring / the permanent
            // zero behind x): a bitwise OR selects it without a test (a select on `code < 0` would also let the compiler
            // sink the LDS read into a branch of its own: one read, one wait, per slot)
            xo[u] = __longlong_as_double(__double_as_longlong(lv[u]) | __double_as_longlong(cur.xv[u]));
        }
        double acc = BSR1 ? cur.bb : 0.0;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const double pr = cur.val[u] * xo[u];                // padded slot: 0 * 0, the sum keeps its bits
            acc = BSR1 ? (acc - pr) : (acc + pr);
        }
        const double xn = BSR1 ? (acc / cur.d) : ((cur.bb - acc) / cur.d);      // no zero diagonals in a chained copy (declined at build)
        double *dst = (cur.row >= 0) ? (double *)((char *)x + (size_t)cur.arow8) : &dummy[t];     // idle lanes store to a scratch line of their own
        *dst = xn;
        ring[(cur.lvl % NB) * CHAIN2_WG + t] = xn;
    };
    // one level: requests for the levels ahead first (nothing in `compute` waits for them), then the row, then the barrier
    auto step = [&](int q, Stage &cur, Stage &fill, const Codes &use, Codes &refill) {
        fill = stage_b(q + D, use);
        refill = stage_a(q + D + A);
        compute(cur);
        __syncthreads();
    };
    // (the compiler drains every outstanding request at the loop's back edge: four rotations per trip make that
    // one drain per 12 levels; steps past the last level run on idle lanes only)
    for (int q = 0; q < nl; q += 12) {
#pragma unroll
        for (int r = 0; r < 12; r += 3) {
            step(q + r, sb[0], sb[2], cq[2], cq[1]);
            step(q + r + 1, sb[1], sb[0], cq[0], cq[2]);
            step(q + r + 2, sb[2], sb[1], cq[1], cq[0]);
        }
    }
}

// threads per workgroup for a run whose widest level has `width` rows: the smallest of 64 .. 512 that covers it (idle
// waves still issue every instruction of the pipeline, so a 40-row level is swept by ONE wave, not by eight)
static int chain_threads(int width) { return width <= 64 ? 64 : (width <= 128 ? 128 : (width <= 256 ? 256 : 512)); }

int launch_gs_chain2(const int *lp, const double *val, const int *code, const int *off, double *dummy, int pf,
                     int l_first, int nlevels, int width, bool reverse, bool bsr1, double *x, const double *bd, int nzero, hipStream_t st)
{
    if (nlevels <= 0) return 0;
    if (nlevels > CHAIN2_LMAX) { set_error("gs_chain2: run longer than one launch holds"); return -4; }
    if (width > CHAIN2_WG) { set_error("gs_chain2: level wider than the workgroup"); return -4; }
    const int wg = chain_threads(width);
#define C2_LAUNCH(B, P, W) hipLaunchKernelGGL((gs_chain2_kernel<B, P, W>), dim3(1), dim3(W), 0, st, lp, val, code, off, x, \
                                              (const double2 *)bd, dummy, nzero, l_first, nlevels, reverse ? 1 : 0)
#define C2_WIDTH(B, P) do { if (wg == 64) C2_LAUNCH(B, P, 64); else if (wg == 128) C2_LAUNCH(B, P, 128); \
                            else if (wg == 256) C2_LAUNCH(B, P, 256); else C2_LAUNCH(B, P, 512); } while (0)
    if (pf == 4) { if (bsr1) C2_WIDTH(true, 4); else C2_WIDTH(false, 4); }
    else if (pf == 8) { if (bsr1) C2_WIDTH(true, 8); else C2_WIDTH(false, 8); }
    else if (pf == 12) { if (bsr1) C2_WIDTH(true, 12); else C2_WIDTH(false, 12); }
    else { set_error("gs_chain2: unsupported slot count"); return -1; }
#undef C2_WIDTH
#undef C2_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gs chain2 launch", __FILE__, __LINE__);
    return 0;
}

// level-order numbering of a scheduled sweep: xp[k] = x[rowmap[k]], bp[k] = b[rowmap[k]] before, x[rowmap[k]] = xp[k] after
__global__ __launch_bounds__(256) void perm_gather_kernel(const int *rowmap, const double *x, const double *b, const double *diag,
                                                           double *xp, double *bp, double2 *bd, int n)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) {
        const int i = rowmap[k];
        const double bi = b[i];
        xp[k] = x[i]; bp[k] = bi;
        bd[k] = make_double2(bi, diag[k]);        // right-hand side and diagonal side by side: one load in the chained sweep
    }
}
__global__ __launch_bounds__(256) void perm_scatter_kernel(const int *rowmap, const double *xp, double *x, int n)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) x[rowmap[k]] = xp[k];
}
int launch_perm_gather(const int *rowmap, const double *x, const double *b, const double *diag, double *xp, double *bp, double *bd,
                       int n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(perm_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rowmap, x, b, diag, xp, bp, (double2 *)bd, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "perm gather launch", __FILE__, __LINE__);
    return 0;
}
int launch_perm_scatter(const int *rowmap, const double *xp, double *x, int n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(perm_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rowmap, xp, x, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "perm scatter launch", __FILE__, __LINE__);
    return 0;
}

static int g_gs_chain = 2;       // 0: a launch per level, 1: chain with operands gathered back from L2, 2 (default): LDS hand-off chain (tools/gs_chain_ab.py)
void set_gs_chain(int on) { g_gs_chain = on; ++g_config_epoch; }
bool gs_chain_enabled() { return g_gs_chain != 0; }
int gs_chain_generation() { return g_gs_chain >= 2 ? 2 : 1; }
int gs_chain_max_rows() { return CHAIN_WG; }

int launch_gs_chain(const DevCsr &G, const int *rowmap, const int *diagpos, const int *level_ptr_dev, int l_first,
                    int nlevels, int width, bool reverse, bool bsr1, double *x, const double *b, hipStream_t st)
{
    if (nlevels <= 0) return 0;
    if (width > CHAIN_WG) { set_error("gs_chain: level wider than the workgroup"); return -4; }
    const int wg = chain_threads(width);
    // at most CHAIN_LMAX levels per launch, pieces in sweep order
    const int npiece = (nlevels + CHAIN_LMAX - 1) / CHAIN_LMAX;
    for (int c = 0; c < npiece; ++c) {
        const int piece = reverse ? npiece - 1 - c : c;
        const int lf = l_first + piece * CHAIN_LMAX;
        const int cnt = std::min(CHAIN_LMAX, nlevels - piece * CHAIN_LMAX);
#define C1_LAUNCH(B, W) hipLaunchKernelGGL((gs_chain_kernel<B, W>), dim3(1), dim3(W), 0, st, G.Ap, G.Aj, G.Ax, rowmap, diagpos, x, b, \
                                           level_ptr_dev, lf, cnt, reverse ? 1 : 0)
#define C1_WIDTH(B) do { if (wg == 64) C1_LAUNCH(B, 64); else if (wg == 128) C1_LAUNCH(B, 128); \
                         else if (wg == 256) C1_LAUNCH(B, 256); else C1_LAUNCH(B, 512); } while (0)
        if (bsr1) C1_WIDTH(true); else C1_WIDTH(false);
#undef C1_WIDTH
#undef C1_LAUNCH
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gs chain launch", __FILE__, __LINE__);
    return 0;
}

// ---------------------------------------------------------------------------
// Chained Gauss-Seidel sweep for runs of dependency levels with FEW BUT LONG rows (the coarse levels of a
// smoothed-aggregation hierarchy: a dozen rows of 30-60 entries per level, hundreds of levels per sweep -- each
// was a launch of ~5 us: three dependent memory round trips for a few hundred entries).  One workgroup, one launch;
// a level is handled the way csr_stream_kernel does it -- entry-parallel products into LDS, then one lane per row
// sums ITS products left to right (same order, same bits) -- but everything that does not depend on x (the level's
// entries, row bounds, diagonal, right-hand side) was requested one or two levels earlier, so the critical path of a
// level is gather x (L2) -> products -> barrier -> row sums from LDS -> store -> barrier.
// A level takes at most WGS rows and WGS * CHAINL_KE entries.
// ---------------------------------------------------------------------------
constexpr int CHAINL_KE = 8;          // entries per lane and level
constexpr int CHAINL_LMAX = 1024;     // levels per launch (row and entry offsets in LDS)
constexpr int CHAINL_WG = 512;

// Written like gs_chain2_kernel: every request unconditional (idle lanes and the steps past the last level re-read a
// valid row / entry and drop it) and the stage buffers rotated by residue, never moved -- so that the compiler's wait
// counts stay exact and a level waits for ITS gathers only, not for the requests of the levels ahead.
template <bool BSR1, int WGS>
__global__ __launch_bounds__(WGS) void gs_chainl_kernel(const int *Ap, const int *Aj, const double *Ax, const int *rowmap,
                                                            const int *diagpos, double *x, const double *b,
                                                            const int *lp, int l_first, int nl, int reverse)
{
    constexpr int KE = CHAINL_KE;
    const int t = threadIdx.x;
    __shared__ int slp[CHAINL_LMAX + 1];          // first row (position in the level-ordered copy) of each level
    __shared__ int sep[CHAINL_LMAX + 1];          // its first entry
    __shared__ double prod[WGS * KE];
    for (int k = t; k <= nl; k += WGS) {
        const int p = lp[l_first + k];
        slp[k] = p;
        sep[k] = Ap[p];
    }
    __syncthreads();
    auto level_of = [&](int q) { q = min(q, nl - 1); return reverse ? nl - 1 - q : q; };

    struct Ent { int c[KE]; double v[KE]; };
    struct Row { int s, e, row, dp, base; bool live; double bb, d; };
    auto entries = [&](int q, Ent &E) {            // the level's entries, lane-strided; lanes past the end re-read the last one
        const int l = level_of(q);
        const int e0 = sep[l], last = sep[l + 1] - e0 - 1;                 // >= 0: a chained level has entries
#pragma unroll
        for (int u = 0; u < KE; ++u) {
            const int k = e0 + min(t + u * WGS, last);
            E.c[u] = Aj[k];
            E.v[u] = Ax[k];
        }
    };
    auto row_a = [&](int q, Row &R) {              // row bounds, row number, diagonal position
        const int l = level_of(q);
        const int cnt = slp[l + 1] - slp[l];                                // >= 1
        const int p = slp[l] + min(t, cnt - 1);
        R.live = (q < nl) && (t < cnt);
        R.base = sep[l];
        R.s = Ap[p]; R.e = Ap[p + 1]; R.row = rowmap ? rowmap[p] : p; R.dp = diagpos[p];
    };
    auto row_b = [&](Row &R) {                     // right-hand side and diagonal (need row_a's answers)
        R.bb = b[R.row];
        const double d = Ax[max(R.dp, 0)];
        R.d = (R.dp >= 0) ? d : 0.0;
    };
    auto step = [&](int q, const Ent &E, Ent &fillE, const Row &R, Row &nextR, Row &fillR) {
        // 1. the operands of this level's entries first; 2. requests for the levels ahead (nothing below waits for them)
        double xv[KE];
#pragma unroll
        for (int u = 0; u < KE; ++u) xv[u] = load_fresh(&x[E.c[u]]);
        entries(q + 2, fillE);
        row_b(nextR);
        row_a(q + 2, fillR);
        // 3. products, entry k of the level at prod[k] (lanes past the end write a slot nobody reads)
#pragma unroll
        for (int u = 0; u < KE; ++u) prod[t + u * WGS] = E.v[u] * xv[u];
        __syncthreads();
        // 4. one lane per row: the sum in stored order, diagonal skipped (reads batched: one wait per 8)
        if (R.live) {
            double acc = BSR1 ? R.bb : 0.0;
            for (int k = R.s; k < R.e; k += 8) {
                double pr[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pr[j] = (k + j < R.e) ? prod[k + j - R.base] : 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k + j < R.e && k + j != R.dp) acc = BSR1 ? (acc - pr[j]) : (acc + pr[j]);
            }
            if (R.d != 0.0) x[R.row] = BSR1 ? (acc / R.d) : ((R.bb - acc) / R.d);
        }
        // one workgroup = one CU: the store is ordered before the next level's gathers (they bypass L1: load_fresh) by the barrier
        __syncthreads();
    };

    Ent eb[3];
    Row rb[3];
    entries(0, eb[0]);
    entries(1, eb[1]);
    row_a(0, rb[0]);
    row_a(1, rb[1]);
    row_b(rb[0]);
    // (the compiler drains every outstanding request at the loop's back edge: four rotations per trip make that one
    // drain per 12 levels; steps past the last level run on idle lanes only)
    for (int q = 0; q < nl; q += 12) {
#pragma unroll
        for (int r = 0; r < 12; r += 3) {
            step(q + r, eb[0], eb[2], rb[0], rb[1], rb[2]);
            step(q + r + 1, eb[1], eb[0], rb[1], rb[2], rb[0]);
            step(q + r + 2, eb[2], eb[1], rb[2], rb[0], rb[1]);
        }
    }
}

// ---------------------------------------------------------------------------
// The same sweep with the hand-off through LDS (what gs_chain2_kernel is to gs_chain_kernel).  Handing a level's new
// values to the next level through memory costs ~2 us per level (store to L2, gather back from L2): 3.6 us per level
// measured for gs_chainl_kernel on 14 rows x 59 entries.  Here a level's new values also go to an LDS ring (3 levels x
// 512), every entry carries a CODE precomputed on the host -- its column, or ~(ring slot) when its operand is produced
// one or two levels earlier in the same launch -- and an operand that is read from memory is final there at least
// two levels before it is needed, so it is requested two levels ahead together with the entry's value (codes four
// levels ahead): nothing on a level's critical path touches memory.
//   step q : ring reads + products -> LDS -> barrier -> row sums (one lane per row, stored order) -> x, ring -> barrier
// Needs every unknown listed once and no zero diagonal in a chained row (declined on the host otherwise).
// ---------------------------------------------------------------------------
// 16-byte requests at 4-byte alignment (a level's first entry sits anywhere in the copy)
struct __attribute__((packed, aligned(4))) QuadI { int v[4]; };
struct __attribute__((packed, aligned(4))) PairD { double v[2]; };

template <bool BSR1, int WGS>
__global__ __launch_bounds__(WGS) void gs_chainl2_kernel(const int *Ap, const int *code, const double *Ax, const int *rowmap,
                                                             const int *diagpos, double *x, const double *b,
                                                             const int *lp, int l_first, int nl, int reverse)
{
    constexpr int KE = CHAINL_KE, NQ = CHAINL_KE / 4;
    constexpr int RING = 3 * CHAINL_WG;
    const int t = threadIdx.x;
    __shared__ int slp[CHAINL_LMAX + 1];
    __shared__ int sep[CHAINL_LMAX + 1];
    __shared__ double prod[WGS * KE];
    __shared__ double ring[RING + 1];              // + 1: what an entry without a ring operand reads
    for (int k = t; k <= nl; k += WGS) {
        const int p = lp[l_first + k];
        slp[k] = p;
        sep[k] = Ap[p];
    }
    if (t == 0) ring[RING] = 0.0;
    __syncthreads();
    auto level_of = [&](int q) { q = min(q, nl - 1); return reverse ? nl - 1 - q : q; };

    // A lane owns NQ quads of 4 consecutive entries (one 16-byte request for the codes, two for the values: a compute
    // unit takes ~16 cycles per wave-level memory instruction whatever its width); quads past the level's end re-read
    // its last four entries (same products to the same slots).
    struct Codes { int c[KE]; int at[NQ]; int s, e, row, dp, base, lvl; bool live; };
    struct Stage { int c[KE]; double v[KE], xv[KE]; int at[NQ]; int s, e, row, dp, base, lvl; bool live; double bb, d; };
    auto stage_a = [&](int q) -> Codes {           // entry codes, row bounds, row number, diagonal position
        Codes C;
        const int l = level_of(q);
        const int e0 = sep[l], lastq = sep[l + 1] - e0 - 4;                 // >= 0: a chained level has at least 4 entries
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            C.at[h] = min(4 * (t + h * WGS), lastq);
            const QuadI cc = *reinterpret_cast<const QuadI *>(code + e0 + C.at[h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) C.c[4 * h + j] = cc.v[j];
        }
        const int cnt = slp[l + 1] - slp[l];                                // >= 1
        const int p = slp[l] + min(t, cnt - 1);
        C.live = (q < nl) && (t < cnt);
        C.base = e0;
        C.lvl = l_first + l;
        C.s = Ap[p]; C.e = Ap[p + 1]; C.row = rowmap ? rowmap[p] : p; C.dp = diagpos[p];
        return C;
    };
    auto stage_b = [&](const Codes &C) -> Stage {  // values, operands that are final in memory, right-hand side, diagonal
        Stage S;
        S.s = C.s; S.e = C.e; S.row = C.row; S.dp = C.dp; S.base = C.base; S.lvl = C.lvl; S.live = C.live;
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            S.at[h] = C.at[h];
            const PairD v0 = *reinterpret_cast<const PairD *>(Ax + C.base + C.at[h]);
            const PairD v1 = *reinterpret_cast<const PairD *>(Ax + C.base + C.at[h] + 2);
            S.v[4 * h] = v0.v[0]; S.v[4 * h + 1] = v0.v[1]; S.v[4 * h + 2] = v1.v[0]; S.v[4 * h + 3] = v1.v[1];
        }
#pragma unroll
        for (int u = 0; u < KE; ++u) {
            S.c[u] = C.c[u];
            S.xv[u] = load_fresh(&x[max(C.c[u], 0)]);
        }
        S.bb = b[C.row];
        S.d = Ax[max(C.dp, 0)];
        return S;
    };
    auto step = [&](Stage &cur, Stage &fill, const Codes &use, Codes &refill, int q) {
        fill = stage_b(use);
        refill = stage_a(q + 4);
        // operands produced by the last two levels come from the ring (every lane reads: an entry without one reads the
        // spare word and drops it)
        double lv[KE];
#pragma unroll
        for (int u = 0; u < KE; ++u) lv[u] = ring[min((unsigned)~cur.c[u], (unsigned)RING)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KE; ++u) prod[cur.at[u / 4] + (u & 3)] = cur.v[u] * ((cur.c[u] < 0) ? lv[u] : cur.xv[u]);
        __syncthreads();
        if (cur.live) {
            // The whole wave runs this for its few live lanes, so the phase is bound by its INSTRUCTION count (SQ counters:
            // ~1000 vector-ALU instructions per level, 17 % of the cycles waiting): the row is summed as two runs, left
            // and right of the diagonal, whole batches of 8 without any per-entry test, then one tested batch.
            double acc = BSR1 ? cur.bb : 0.0;
            auto run = [&](int lo, int hi) {                                // prod[lo .. hi), left to right
                const double *pp = prod + lo;
                int left = hi - lo;
                for (; left >= 8; left -= 8, pp += 8) {
                    double pr[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pr[j] = pp[j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = BSR1 ? (acc - pr[j]) : (acc + pr[j]);
                }
                if (left > 0) {
                    double pr[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) pr[j] = pp[min(j, left - 1)];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const double nxt = BSR1 ? (acc - pr[j]) : (acc + pr[j]);
                        acc = (j < left) ? nxt : acc;
                    }
                }
            };
            run(cur.s - cur.base, cur.dp - cur.base);                        // (a chained row has its diagonal: dp >= s)
            run(cur.dp + 1 - cur.base, cur.e - cur.base);
            const double xn = BSR1 ? (acc / cur.d) : ((cur.bb - acc) / cur.d);      // no zero diagonals in a chained row (declined at build)
            x[cur.row] = xn;
            ring[(cur.lvl % 3) * CHAINL_WG + t] = xn;
        }
        __syncthreads();
    };

    Stage sb[3];
    Codes cq[3];
    cq[0] = stage_a(0);
    cq[1] = stage_a(1);
    sb[0] = stage_b(cq[0]);
    sb[1] = stage_b(cq[1]);
    cq[2] = stage_a(2);
    cq[0] = stage_a(3);
    // (the compiler drains every outstanding request at the loop's back edge: four rotations per trip make that one
    // drain per 12 levels; steps past the last level run on idle lanes only)
    for (int q = 0; q < nl; q += 12) {
#pragma unroll
        for (int r = 0; r < 12; r += 3) {
            step(sb[0], sb[2], cq[2], cq[1], q + r);
            step(sb[1], sb[0], cq[0], cq[2], q + r + 1);
            step(sb[2], sb[1], cq[1], cq[0], q + r + 2);
        }
    }
}

int gs_chainl_max_rows() { return CHAINL_WG; }
int gs_chainl_entries_per_lane() { return CHAINL_KE; }
int gs_chainl_max_levels() { return CHAINL_LMAX; }

int launch_gs_chain_long(const DevCsr &G, const int *rowmap, const int *diagpos, const int *level_ptr_dev, int l_first,
                         int nlevels, int width, bool reverse, bool bsr1, double *x, const double *b, hipStream_t st,
                         const int *ring_code)
{
    if (ring_code) {
        // LDS hand-off: one launch holds the whole piece (the host cut the pieces accordingly)
        if (nlevels <= 0) return 0;
        if (nlevels > CHAINL_LMAX) { set_error("gs_chain_long: run longer than one launch holds"); return -4; }
        if (width != 64 && width != 128 && width != 256 && width != 512) { set_error("gs_chain_long: workgroup size not instantiated"); return -4; }
#define CL2_LAUNCH(B, W) hipLaunchKernelGGL((gs_chainl2_kernel<B, W>), dim3(1), dim3(W), 0, st, G.Ap, ring_code, G.Ax, rowmap, diagpos, x, b, \
                                            level_ptr_dev, l_first, nlevels, reverse ? 1 : 0)
#define CL2_WIDTH(B) do { if (width == 64) CL2_LAUNCH(B, 64); else if (width == 128) CL2_LAUNCH(B, 128); else if (width == 256) CL2_LAUNCH(B, 256); else CL2_LAUNCH(B, 512); } while (0)
        if (bsr1) CL2_WIDTH(true); else CL2_WIDTH(false);
#undef CL2_WIDTH
#undef CL2_LAUNCH
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) return hip_fail(e2, "gs long-row chain (LDS hand-off) launch", __FILE__, __LINE__);
        return 0;
    }
    if (nlevels <= 0) return 0;
    if (width == 64) width = 128;
    if (width != 128 && width != 256 && width != 512) { set_error("gs_chain_long: workgroup size not instantiated"); return -4; }
    const int npiece = (nlevels + CHAINL_LMAX - 1) / CHAINL_LMAX;
    for (int c = 0; c < npiece; ++c) {
        const int piece = reverse ? npiece - 1 - c : c;
        const int lf = l_first + piece * CHAINL_LMAX;
        const int cnt = std::min(CHAINL_LMAX, nlevels - piece * CHAINL_LMAX);
#define CL_LAUNCH(B, W) hipLaunchKernelGGL((gs_chainl_kernel<B, W>), dim3(1), dim3(W), 0, st, G.Ap, G.Aj, G.Ax, rowmap, diagpos, x, b, \
                                           level_ptr_dev, lf, cnt, reverse ? 1 : 0)
#define CL_WIDTH(B) do { if (width == 128) CL_LAUNCH(B, 128); else if (width == 256) CL_LAUNCH(B, 256); else CL_LAUNCH(B, 512); } while (0)
        if (bsr1) CL_WIDTH(true); else CL_WIDTH(false);
#undef CL_WIDTH
#undef CL_LAUNCH
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gs long-row chain launch", __FILE__, __LINE__);
    return 0;
}

// ---------------------------------------------------------------------------
// csr_pattern: the same operator application for matrices whose rows repeat a few column-offset
// patterns (stencil operators: 27 patterns for a 7-point grid operator with boundaries).  Only the
// values are streamed (8 B per entry, 16-byte loads through the same LDS tile); a row's columns
// come from its pattern id (4 B per row) and the dictionary held in LDS.  Thread t gathers
// x[i + off_j] for ITS row -- neighbouring lanes read neighbouring addresses, so the gather is
// coalesced -- and accumulates in storage order: bit-identical to csr_stream, 21 % fewer bytes
// on the 7-point level.
// ---------------------------------------------------------------------------
struct PatternArgs {
    const int *pat;
    const int *dict_ptr;
    const int *dict_off;
    int npat, ndict;
    // plane-periodic block -> XCD mapping (0 = off): the operator's slowest axis has a stride of
    // about period_blocks row blocks; each period is cut into 8 segments of seg_blocks and XCD k
    // (= blockIdx % 8, the hardware's round-robin dispatch) sweeps segment k of every period in
    // turn.  A row's neighbours one plane up and down are then gathered by the SAME XCD, a few
    // hundred blocks earlier or later, i.e. out of its own L2 instead of over the fabric again.
    int period_blocks, seg_blocks, nblocks;
};

template <int MODE>
__global__ __launch_bounds__(WG) void csr_pattern_kernel(StreamArgs a, PatternArgs P, int xcd_chunk, int rpb)
{
    using MT = ModeTraits<MODE>;
    __shared__ double sp[TILE];
    __shared__ int sAp[WG + 1];
    __shared__ int sDptr[PAT_MAX + 2];
    __shared__ int sDict[PAT_DICT_MAX];

    const int t = threadIdx.x;
    int blk;
    if (P.period_blocks > 0) {
        const int k = blockIdx.x & 7, s = blockIdx.x >> 3;
        const int p = s / P.seg_blocks, j = s - p * P.seg_blocks;
        const int q = k * P.seg_blocks + j;
        blk = p * P.period_blocks + q;
        if (q >= P.period_blocks || blk >= P.nblocks) return;       // padding of the last segment / period
    } else {
        blk = remap_block(blockIdx.x, gridDim.x, xcd_chunk);
    }
    const int r0 = a.row_lo + blk * rpb;
    const int nr = min(rpb, a.row_hi - r0);
    const double gscale = a.gscale;
    const long nnz_total = a.nnz_total;

    for (int i = t; i <= nr; i += WG) sAp[i] = a.Ap[r0 + i];
    for (int i = t; i <= P.npat; i += WG) sDptr[i] = P.dict_ptr[i];
    for (int i = t; i < P.ndict; i += WG) sDict[i] = P.dict_off[i];
    __syncthreads();

    const int kbeg = sAp[0], kend = sAp[nr];
    const int my_s = (t < nr) ? sAp[t] : kend;
    const int my_e = (t < nr) ? sAp[t + 1] : kend;
    const int i = r0 + t;
    const int *offs = sDict;
    if (t < nr) offs = &sDict[sDptr[P.pat[i]]];
    double acc = 0.0, diag = 0.0;
    // The operands of a row's first 8 entries do not depend on the matrix values: gather them (and
    // the right-hand side) NOW, so that their latency overlaps the value stream instead of
    // following it -- one dependent memory round trip less per workgroup.
    const int my_len = my_e - my_s;
    double xv0[8];
    int off0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        off0[u] = (u < my_len) ? offs[u] : 0;
        xv0[u] = (u < my_len) ? a.xg[i + off0[u]] : 0.0;
    }
    double bval = 0.0;
    if (t < nr && (MT::sub || MODE == SM_RESIDUAL || MODE == SM_RESIDUAL_SUMSQ || MODE == SM_POLY_STEP ||
                   MODE == SM_POLY_LAST || MODE == SM_JACOBI))
        bval = a.b[i];
    if (MT::sub && t < nr) acc = bval;
    double pre2 = 0.0;                                  // second epilogue operand, same reasoning
    if (t < nr) {
        if (MODE == SM_MATVEC_ACC) pre2 = a.out[i];
        else if (MODE == SM_POLY_LAST || MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1) pre2 = a.v2[i];
    }

    const int abeg = kbeg & ~3;
    for (int tile_lo = abeg; tile_lo < kend; tile_lo += TILE) {
        const int tile_hi = min(tile_lo + TILE, kend);
        constexpr int NQ = TILE / (4 * WG);
#pragma unroll
        for (int p = 0; p < NQ; ++p) {
            const int e = tile_lo + p * (4 * WG) + 4 * t;
            if (e < tile_hi) {
                v2d v0, v1;
                if ((long)e + 4 <= nnz_total) {
                    v0 = load_v2d(a.Ax + e);
                    v1 = load_v2d(a.Ax + e + 2);
                } else {
                    double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = ((long)e + u < nnz_total) ? a.Ax[e + u] : 0.0;
                    v0 = v2d{v[0], v[1]};
                    v1 = v2d{v[2], v[3]};
                }
                const int q = e - tile_lo;
                *reinterpret_cast<v2d *>(&sp[q]) = v0;
                *reinterpret_cast<v2d *>(&sp[q + 2]) = v1;
            }
        }
        __syncthreads();
        {
            const int s = max(my_s, tile_lo), e2 = min(my_e, tile_hi);
            for (int k0 = s; k0 < e2; k0 += 8) {
                double xv[8];
                int off[8];
                const bool first = (k0 == my_s);       // the pre-gathered batch
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u;
                    if (first) { off[u] = off0[u]; xv[u] = xv0[u]; }
                    else {
                        off[u] = (k < e2) ? offs[k - my_s] : 0;
                        xv[u] = (k < e2) ? a.xg[i + off[u]] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u;
                    if (k < e2) {
                        const double av = sp[k - tile_lo];
                        if (MT::jac && off[u] == 0) { diag = av; continue; }
                        const double pr = av * (gscale * xv[u]);
                        acc = MT::sub ? (acc - pr) : (acc + pr);
                    }
                }
            }
        }
        __syncthreads();
    }

    if (MODE == SM_RESIDUAL_SUMSQ) {
        double sq = 0.0;
        if (t < nr) {
            double rr = bval - acc;
            sq = rr * rr;
            if (a.out) store_out(&a.out[i], rr);
        }
        __syncthreads();
        double tot = block_reduce_sum(sq, sp);
        if (t == 0) a.out2[blk] = tot;
        return;
    }
    if (t >= nr) return;
    if (MODE == SM_MATVEC) {
        store_out(&a.out[i], acc);
    } else if (MODE == SM_MATVEC_ACC) {
        store_out(&a.out[i], pre2 + acc);
    } else if (MODE == SM_RESIDUAL) {
        store_out(&a.out[i], bval - acc);
    } else if (MODE == SM_POLY_STEP) {
        double cr = a.c0 * bval;
        a.out[i] = cr + acc;
    } else if (MODE == SM_POLY_LAST) {
        double cr = a.c0 * bval;
        double h = cr + acc;
        store_out(&a.out[i], pre2 + h);
    } else if (MODE == SM_JACOBI) {
        double told = pre2;
        if (diag != 0.0) {
            double q = (bval - acc) / diag;
            double t1 = (1.0 - a.c0) * told;
            double t2 = a.c0 * q;
            a.out[i] = t1 + t2;
        } else {
            a.out[i] = told;
        }
    } else if (MODE == SM_JACOBI_BSR1) {
        double told = pre2;
        if (diag != 0.0) {
            double t1 = (1.0 - a.c0) * told;
            double t2 = (a.c0 * acc) / diag;
            a.out[i] = t1 + t2;
        } else {
            a.out[i] = told;
        }
    }
}

bool pattern_supports(StreamMode mode)
{
    switch (mode) {
    case SM_MATVEC: case SM_MATVEC_ACC: case SM_RESIDUAL: case SM_POLY_STEP: case SM_POLY_LAST:
    case SM_JACOBI: case SM_JACOBI_BSR1: case SM_RESIDUAL_SUMSQ: return true;
    default: return false;
    }
}

template <int MODE>
static int launch_pattern_mode(const StreamArgs &a, const PatternArgs &P, int period_rows, hipStream_t st)
{
    int rows = a.row_hi - a.row_lo;
    if (rows <= 0) return 0;
    int rpb = a.rows_per_wg;
    if (rpb < 1 || rpb > WG) rpb = WG;
    int nb = (rows + rpb - 1) / rpb;
    StreamArgs b = a;
    if (b.gscale == 0.0) b.gscale = 1.0;
    PatternArgs Q = P;
    Q.nblocks = nb;
    Q.period_blocks = Q.seg_blocks = 0;
    int grid = nb;
    if (g_xcd_period && period_rows > 0) {
        const int S = (period_rows + rpb / 2) / rpb;            // blocks per plane (rounded; drift is harmless)
        const int G = (S + 7) / 8;
        // worth it when a plane spans many blocks and three segments of x fit an XCD's 4 MB L2
        if (S >= 64 && nb >= 4 * S && 3.0 * G * rpb * 8.0 <= 2.0e6) {
            Q.period_blocks = S;
            Q.seg_blocks = G;
            grid = 8 * G * ((nb + S - 1) / S);
        }
    }
    hipLaunchKernelGGL((csr_pattern_kernel<MODE>), dim3(grid), dim3(WG), 0, st, b, Q, g_xcd_chunk, rpb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "csr_pattern launch", __FILE__, __LINE__);
    return 0;
}

int launch_pattern(StreamMode mode, const StreamArgs &a, const DevCsr &M, hipStream_t st)
{
    PatternArgs P{M.pat, M.dict_ptr, M.dict_off, M.npat, M.ndict, 0, 0, 0};
    const int period_rows = M.period_rows;
    switch (mode) {
    case SM_MATVEC: return launch_pattern_mode<SM_MATVEC>(a, P, period_rows, st);
    case SM_MATVEC_ACC: return launch_pattern_mode<SM_MATVEC_ACC>(a, P, period_rows, st);
    case SM_RESIDUAL: return launch_pattern_mode<SM_RESIDUAL>(a, P, period_rows, st);
    case SM_POLY_STEP: return launch_pattern_mode<SM_POLY_STEP>(a, P, period_rows, st);
    case SM_POLY_LAST: return launch_pattern_mode<SM_POLY_LAST>(a, P, period_rows, st);
    case SM_JACOBI: return launch_pattern_mode<SM_JACOBI>(a, P, period_rows, st);
    case SM_JACOBI_BSR1: return launch_pattern_mode<SM_JACOBI_BSR1>(a, P, period_rows, st);
    case SM_RESIDUAL_SUMSQ: return launch_pattern_mode<SM_RESIDUAL_SUMSQ>(a, P, period_rows, st);
    default: break;
    }
    set_error("launch_pattern: mode not supported");
    return -1;
}

// ---------------------------------------------------------------------------
// Stencil form (DevCsr::st_*): one thread per row, every operand requested before anything is
// consumed -- NU coalesced value loads (slot-major inside a 256-row block), NU speculative gathers
// x[i + U[u]] (bounds-checked, masked later), the row mask and the epilogue operands.  One memory
// round trip per workgroup, no LDS, no barrier (except the SM_RESIDUAL_SUMSQ reduction).  The row
// sum runs over the slots whose mask bit is set, in increasing slot order = increasing column =
// the CSR's stored order, so the result is bit-identical to csr_stream_kernel's.
// ---------------------------------------------------------------------------
// operands the stencil kernel reads exactly once per launch (values, masks, right-hand side): loaded
// non-temporally so that they do not displace the gathered vector in L2 (measured -1.2 % per launch)
template <class T> __device__ __forceinline__ T stream_load(const T *p)
{
#ifndef AMG_STENCIL_PLAIN_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct StencilArgs {
    const double *vals;
    const unsigned char *codes;          // value index: one byte per slot instead of vals (null = off)
    const double *dict;                  // its dictionary (<= 256 doubles), copied to LDS by every workgroup
    int ndict;
    int ranges;                          // != 0: some rows are left to the pattern kernel (their mask has the top bit set)
    const void *mask;                    // uint8 per row in the NUB == 8 instantiation (|U| <= 7), else uint32;
                                         // top bit set = the row is applied by the pattern kernel instead
    int nu, u0;
    int blk_lo, nblocks;                 // blocks [blk_lo, blk_lo + nblocks) cover the row range
    int period_blocks, seg_blocks;       // plane-periodic block -> XCD mapping, see PatternArgs
    int ncols;
    int off[STENCIL_MAX];
};

// what a stencil application does with a row's sum (shared by the 8-byte-value and the coded-value kernels)
template <int MODE>
__device__ __forceinline__ void stencil_epilogue(const StreamArgs &a, const StencilArgs &E, int i, int blk, int t, bool covered,
                                                  double acc, double diag, double bval, double pre2, double *red)
{
    if (MODE == SM_RESIDUAL_SUMSQ) {
        double sq = 0.0;
        if (covered) {
            double rr = bval - acc;
            sq = rr * rr;
            if (a.out) store_out(&a.out[i], rr);
        }
        double tot = block_reduce_sum(sq, red);
        if (t == 0) a.out2[blk - E.blk_lo] = tot;
        return;
    }
    if (!covered) return;
    if (MODE == SM_MATVEC) {
        store_out(&a.out[i], acc);
    } else if (MODE == SM_MATVEC_ACC) {
        store_out(&a.out[i], pre2 + acc);
    } else if (MODE == SM_RESIDUAL) {
        store_out(&a.out[i], bval - acc);
    } else if (MODE == SM_POLY_STEP) {
        double cr = a.c0 * bval;
        a.out[i] = cr + acc;
    } else if (MODE == SM_POLY_LAST) {
        double cr = a.c0 * bval;
        double h = cr + acc;
        store_out(&a.out[i], pre2 + h);
    } else if (MODE == SM_JACOBI) {
        double told = pre2;
        if (diag != 0.0) {
            double q = (bval - acc) / diag;
            double t1 = (1.0 - a.c0) * told;
            double t2 = a.c0 * q;
            a.out[i] = t1 + t2;
        } else {
            a.out[i] = told;
        }
    } else if (MODE == SM_JACOBI_BSR1) {
        double told = pre2;
        if (diag != 0.0) {
            double t1 = (1.0 - a.c0) * told;
            double t2 = (a.c0 * acc) / diag;
            a.out[i] = t1 + t2;
        } else {
            a.out[i] = told;
        }
    }
}

// ---------------------------------------------------------------------------
// Coded values (value index), stencils of up to 7 offsets (uint8 row masks): the level-0 kernel of every constant-coefficient configuration.
// Same rows, same arithmetic as stencil_kernel<MODE, 8> with E.codes set, written as straight-line code: every request is
// unconditional (a lane outside the row range works on the range's first row and stores nothing; an offset that leaves the
// operand vector is clamped -- its mask bit is clear, the value is not used), the dictionary is requested first and written
// to LDS after all other requests, so no wave waits for it alone, and absent slots are skipped by selects instead of branches.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(WG) void stencil_coded_kernel(StreamArgs a, StencilArgs E, int xcd_chunk)
{
    using MT = ModeTraits<MODE>;
    __shared__ double red[8];
    __shared__ double sdict[256];
    const int t = threadIdx.x;
    int blk;
    if (E.period_blocks > 0) {
        const int k = blockIdx.x & 7, s = blockIdx.x >> 3;
        const int p = s / E.seg_blocks, j = s - p * E.seg_blocks;
        const int q = k * E.seg_blocks + j;
        blk = p * E.period_blocks + q;
        if (q >= E.period_blocks || blk >= E.nblocks) return;
    } else {
        blk = remap_block(blockIdx.x, gridDim.x, xcd_chunk);
    }
    blk += E.blk_lo;
    const int i = blk * WG + t;
    const bool live = (i >= a.row_lo && i < a.row_hi);
    double dv = 0.0;
    if (t < E.ndict) dv = E.dict[t];
    // lanes outside the range read what the nearest live lane of this block reads
    const int ic = min(max(i, a.row_lo), a.row_hi - 1);
    const int tc = ic - blk * WG;
    const unsigned long long cw = stream_load(reinterpret_cast<const unsigned long long *>(E.codes) + (size_t)blk * WG + tc);
    const int last = E.ncols - 1;
    double xv[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) xv[u] = a.xg[min(max(ic + E.off[u], 0), last)];   // (slots >= nu: offset 0, mask bit clear)
    // which slots a row stores is in its codes (255 = absent: value_encode_kernel); the row mask is only read where some
    // rows are left to the pattern kernel (top bit)
    unsigned m = 0;
    if (E.ranges) m = (unsigned)stream_load(static_cast<const unsigned char *>(E.mask) + ic);      // (uniform)
    double bval = 0.0, pre2 = 0.0;
    if (MT::sub || MODE == SM_RESIDUAL || MODE == SM_RESIDUAL_SUMSQ || MODE == SM_POLY_STEP || MODE == SM_POLY_LAST || MODE == SM_JACOBI)
        bval = stream_load(&a.b[ic]);
    if (MODE == SM_MATVEC_ACC) pre2 = a.out[ic];
    else if (MODE == SM_POLY_LAST || MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1) pre2 = a.v2[ic];
    if (t < E.ndict) sdict[t] = dv;
    const bool covered = live && !(m & 0x80u);
    __syncthreads();                                              // dictionary in LDS
    double v[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) v[u] = sdict[(unsigned)(cw >> (8 * u)) & 0xFFu];
    const double gscale = a.gscale;
    double acc = MT::sub ? bval : 0.0, diag = 0.0;
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const bool on = covered && ((unsigned)(cw >> (8 * u)) & 0xFFu) != 255u;      // (slots >= nu and unstored slots: 255)
        if (MT::jac) {
            const bool isd = u == E.u0;
            diag = (on && isd) ? v[u] : diag;
            const double pr = v[u] * (gscale * xv[u]);
            const double nx = MT::sub ? (acc - pr) : (acc + pr);
            acc = (on && !isd) ? nx : acc;
        } else {
            const double pr = v[u] * (gscale * xv[u]);
            const double nx = MT::sub ? (acc - pr) : (acc + pr);
            acc = on ? nx : acc;
        }
    }
    stencil_epilogue<MODE>(a, E, i, blk, t, covered, acc, diag, bval, pre2, red);
}

template <int MODE, int NUB>
__global__ __launch_bounds__(WG) void stencil_kernel(StreamArgs a, StencilArgs E, int xcd_chunk)
{
    using MT = ModeTraits<MODE>;
    __shared__ double red[8];
    __shared__ double sdict[256];
    const int t = threadIdx.x;
    int blk;
    if (E.period_blocks > 0) {
        const int k = blockIdx.x & 7, s = blockIdx.x >> 3;
        const int p = s / E.seg_blocks, j = s - p * E.seg_blocks;
        const int q = k * E.seg_blocks + j;
        blk = p * E.period_blocks + q;
        if (q >= E.period_blocks || blk >= E.nblocks) return;
    } else {
        blk = remap_block(blockIdx.x, gridDim.x, xcd_chunk);
    }
    blk += E.blk_lo;
    const int i = blk * WG + t;
    const bool live = (i >= a.row_lo && i < a.row_hi);
    const bool vi = E.codes != nullptr;                       // uniform
    const double *vp = E.vals + ((size_t)blk * E.nu) * WG + t;
    // value index: the codes of 8 slots of a row packed into one 64-bit word, [block][word][256 rows]
    constexpr int NW = NUB / 8;
    const int nw = (E.nu + 7) >> 3;
    const unsigned long long *cp = reinterpret_cast<const unsigned long long *>(E.codes) + ((size_t)blk * nw) * WG + t;
    if (vi && t < E.ndict) sdict[t] = E.dict[t];

    double v[NUB], xv[NUB];
    unsigned long long cw[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) cw[w] = (vi && live && w < nw) ? stream_load(&cp[(size_t)w * WG]) : 0ULL;
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
        v[u] = 0.0; xv[u] = 0.0;
        if (u < E.nu && live) {
            if (!vi) v[u] = stream_load(&vp[(size_t)u * WG]);
            const long j = (long)i + E.off[u];
            if (j >= 0 && j < E.ncols) xv[u] = a.xg[j];
        }
    }
    unsigned m = 0;
    double bval = 0.0, pre2 = 0.0;
    if (live) {
        m = (NUB == 8) ? (unsigned)stream_load(static_cast<const unsigned char *>(E.mask) + i)
                       : stream_load(static_cast<const unsigned *>(E.mask) + i);
        if (m & ((NUB == 8) ? 0x80u : 0x80000000u)) m = 0xFFFFFFFFu;          // not covered
        if (MT::sub || MODE == SM_RESIDUAL || MODE == SM_RESIDUAL_SUMSQ || MODE == SM_POLY_STEP ||
            MODE == SM_POLY_LAST || MODE == SM_JACOBI)
            bval = stream_load(&a.b[i]);
        if (MODE == SM_MATVEC_ACC) pre2 = a.out[i];
        else if (MODE == SM_POLY_LAST || MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1) pre2 = a.v2[i];
    }
    const bool covered = live && m != 0xFFFFFFFFu;
    if (!covered) m = 0;
    if (vi) {
        __syncthreads();                                      // dictionary in LDS
#pragma unroll
        for (int u = 0; u < NUB; ++u) v[u] = sdict[(unsigned)(cw[u >> 3] >> (8 * (u & 7))) & 0xFFu];
    }
    const double gscale = a.gscale;
    double acc = MT::sub ? bval : 0.0, diag = 0.0;
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
        if (u < E.nu && ((m >> u) & 1u)) {
            if (MT::jac && u == E.u0) { diag = v[u]; continue; }
            const double pr = v[u] * (gscale * xv[u]);
            acc = MT::sub ? (acc - pr) : (acc + pr);
        }
    }

    stencil_epilogue<MODE>(a, E, i, blk, t, covered, acc, diag, bval, pre2, red);
}

// ---------------------------------------------------------------------------
// The same application with TWO consecutive rows per lane (128 lanes per 256-row block): every streamed operand --
// the values of a slot, the gathered x, the masks, right-hand side, result -- is one 16-byte access for two rows.
// SQ counters of the one-row kernel at 500^3: 57 % of the wave cycles waiting for instruction ISSUE; the compute unit
// takes ~16 cycles per wave-level memory instruction whatever its width, and that kernel issues 17 of them per 64
// rows.  Half the instructions for the same bytes here; per-row arithmetic, order and results unchanged.
// ---------------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) PairU { double x, y; };        // two doubles at 8-byte alignment (gathers at odd offsets)
typedef unsigned long long v2ull __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr int WG2 = WG / 2;

template <int MODE, int NUB>
__global__ __launch_bounds__(WG2) void stencil2_kernel(StreamArgs a, StencilArgs E, int xcd_chunk)
{
    using MT = ModeTraits<MODE>;
    __shared__ double red[WG + 8];
    __shared__ double sdict[256];
    const int t = threadIdx.x;
    int blk;
    if (E.period_blocks > 0) {
        const int k = blockIdx.x & 7, s = blockIdx.x >> 3;
        const int p = s / E.seg_blocks, j = s - p * E.seg_blocks;
        const int q = k * E.seg_blocks + j;
        blk = p * E.period_blocks + q;
        if (q >= E.period_blocks || blk >= E.nblocks) return;
    } else {
        blk = remap_block(blockIdx.x, gridDim.x, xcd_chunk);
    }
    blk += E.blk_lo;
    const int i0 = blk * WG + 2 * t;                          // rows i0, i0 + 1
    const bool live0 = (i0 >= a.row_lo && i0 < a.row_hi), live1 = (i0 + 1 >= a.row_lo && i0 + 1 < a.row_hi);
    const bool both = live0 && live1, any = live0 || live1;
    const bool vi = E.codes != nullptr;                       // uniform
    const double *vp = E.vals + ((size_t)blk * E.nu) * WG + 2 * t;
    constexpr int NW = NUB / 8;
    const int nw = (E.nu + 7) >> 3;
    const unsigned long long *cp = reinterpret_cast<const unsigned long long *>(E.codes) + ((size_t)blk * nw) * WG + 2 * t;
    if (vi) { sdict[t] = (t < E.ndict) ? E.dict[t] : 0.0; sdict[t + WG2] = (t + WG2 < E.ndict) ? E.dict[t + WG2] : 0.0; }

    double v[2][NUB], xv[2][NUB];
    unsigned long long cw[2][NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        cw[0][w] = cw[1][w] = 0ULL;
        if (vi && any && w < nw) {
            const v2ull c2 = stream_load(reinterpret_cast<const v2ull *>(&cp[(size_t)w * WG]));
            cw[0][w] = c2.x; cw[1][w] = c2.y;
        }
    }
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
        v[0][u] = v[1][u] = 0.0; xv[0][u] = xv[1][u] = 0.0;
        if (u < E.nu && any) {
            if (!vi) {
                const v2d vv = stream_load(reinterpret_cast<const v2d *>(&vp[(size_t)u * WG]));
                v[0][u] = vv.x; v[1][u] = vv.y;
            }
            const long j = (long)i0 + E.off[u];
            if (j >= 0 && j + 1 < E.ncols) {
                const PairU xx = *reinterpret_cast<const PairU *>(&a.xg[j]);
                xv[0][u] = xx.x; xv[1][u] = xx.y;
            } else {
                if (j >= 0 && j < E.ncols) xv[0][u] = a.xg[j];
                if (j + 1 >= 0 && j + 1 < E.ncols) xv[1][u] = a.xg[j + 1];
            }
        }
    }
    unsigned m[2] = {0, 0};
    double bval[2] = {0.0, 0.0}, pre2[2] = {0.0, 0.0};
    constexpr bool need_b = MT::sub || MODE == SM_RESIDUAL || MODE == SM_RESIDUAL_SUMSQ || MODE == SM_POLY_STEP ||
                            MODE == SM_POLY_LAST || MODE == SM_JACOBI;
    if (any) {
        if (NUB == 8) {
            const unsigned short mm = stream_load(reinterpret_cast<const unsigned short *>(static_cast<const unsigned char *>(E.mask) + i0));
            m[0] = mm & 0xFFu; m[1] = mm >> 8;
        } else {
            const v2u mm = stream_load(reinterpret_cast<const v2u *>(static_cast<const unsigned *>(E.mask) + i0));
            m[0] = mm.x; m[1] = mm.y;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (m[r] & ((NUB == 8) ? 0x80u : 0x80000000u)) m[r] = 0xFFFFFFFFu;          // not covered
        if (need_b) {
            const v2d bb = stream_load(reinterpret_cast<const v2d *>(&a.b[i0]));
            bval[0] = bb.x; bval[1] = bb.y;
        }
        if (MODE == SM_MATVEC_ACC) { const v2d pp = *reinterpret_cast<const v2d *>(&a.out[i0]); pre2[0] = pp.x; pre2[1] = pp.y; }
        else if (MODE == SM_POLY_LAST || MODE == SM_JACOBI || MODE == SM_JACOBI_BSR1) {
            const v2d pp = *reinterpret_cast<const v2d *>(&a.v2[i0]); pre2[0] = pp.x; pre2[1] = pp.y;
        }
    }
    bool covered[2] = {live0 && m[0] != 0xFFFFFFFFu, live1 && m[1] != 0xFFFFFFFFu};
#pragma unroll
    for (int r = 0; r < 2; ++r) if (!covered[r]) m[r] = 0;
    if (vi) {
        __syncthreads();                                      // dictionary in LDS
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int u = 0; u < NUB; ++u) v[r][u] = sdict[(unsigned)(cw[r][u >> 3] >> (8 * (u & 7))) & 0xFFu];
    }
    const double gscale = a.gscale;
    double acc[2], diag[2] = {0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        acc[r] = MT::sub ? bval[r] : 0.0;
#pragma unroll
        for (int u = 0; u < NUB; ++u) {
            if (u < E.nu && ((m[r] >> u) & 1u)) {
                if (MT::jac && u == E.u0) { diag[r] = v[r][u]; continue; }
                const double pr = v[r][u] * (gscale * xv[r][u]);
                acc[r] = MT::sub ? (acc[r] - pr) : (acc[r] + pr);
            }
        }
    }

    double res[2] = {0.0, 0.0};
    if (MODE == SM_RESIDUAL_SUMSQ) {
        double sq[2] = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (covered[r]) { res[r] = bval[r] - acc[r]; sq[r] = res[r] * res[r]; }
        if (a.out) {
            if (covered[0] && covered[1]) __builtin_nontemporal_store(v2d{res[0], res[1]}, reinterpret_cast<v2d *>(&a.out[i0]));
            else { if (covered[0]) store_out(&a.out[i0], res[0]); if (covered[1]) store_out(&a.out[i0 + 1], res[1]); }
        }
        // the one-row kernel's summation tree, bit for bit: row r sits in lane r % 64 of wave r / 64, a shuffle tree per
        // wave, the four wave sums added in order
        *reinterpret_cast<v2d *>(&red[2 * t]) = v2d{sq[0], sq[1]};
        __syncthreads();
        double s0 = red[t], s1 = red[t + WG2];                // wave 0: rows 0..63 and 128..191, wave 1: 64..127 and 192..255
        for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); }
        __syncthreads();
        if ((t & 63) == 0) { red[WG + (t >> 6)] = s0; red[WG + 2 + (t >> 6)] = s1; }
        __syncthreads();
        if (t == 0) {
            double tot = 0.0;
            for (int k = 0; k < 4; ++k) tot += red[WG + k];
            a.out2[blk - E.blk_lo] = tot;
        }
        return;
    }
    bool wr[2] = {covered[0], covered[1]};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (MODE == SM_MATVEC) res[r] = acc[r];
        else if (MODE == SM_MATVEC_ACC) res[r] = pre2[r] + acc[r];
        else if (MODE == SM_RESIDUAL) res[r] = bval[r] - acc[r];
        else if (MODE == SM_POLY_STEP) { const double cr = a.c0 * bval[r]; res[r] = cr + acc[r]; }
        else if (MODE == SM_POLY_LAST) { const double cr = a.c0 * bval[r]; const double h = cr + acc[r]; res[r] = pre2[r] + h; }
        else if (MODE == SM_JACOBI) {
            if (diag[r] != 0.0) { const double q = (bval[r] - acc[r]) / diag[r]; const double t1 = (1.0 - a.c0) * pre2[r]; const double t2 = a.c0 * q; res[r] = t1 + t2; }
            else res[r] = pre2[r];
        } else if (MODE == SM_JACOBI_BSR1) {
            if (diag[r] != 0.0) { const double t1 = (1.0 - a.c0) * pre2[r]; const double t2 = (a.c0 * acc[r]) / diag[r]; res[r] = t1 + t2; }
            else res[r] = pre2[r];
        }
    }
    constexpr bool streamed = (MODE == SM_MATVEC || MODE == SM_MATVEC_ACC || MODE == SM_RESIDUAL || MODE == SM_POLY_LAST);
    if (wr[0] && wr[1]) {
        if (streamed) __builtin_nontemporal_store(v2d{res[0], res[1]}, reinterpret_cast<v2d *>(&a.out[i0]));
        else *reinterpret_cast<v2d *>(&a.out[i0]) = v2d{res[0], res[1]};
    } else {
        if (wr[0]) { if (streamed) store_out(&a.out[i0], res[0]); else a.out[i0] = res[0]; }
        if (wr[1]) { if (streamed) store_out(&a.out[i0 + 1], res[1]); else a.out[i0 + 1] = res[1]; }
    }
    (void)both;
}

static int g_stencil_coded = std::getenv("AMG_STENCIL_CODED") ? std::atoi(std::getenv("AMG_STENCIL_CODED")) : 1;   // 0: coded values through the generic kernel (A/B)
static int g_stencil_pairs = 1;     // 1: two rows per lane (stencil2_kernel), 0: one row per lane
void set_stencil_pairs(int on) { g_stencil_pairs = on; ++g_config_epoch; }

struct StencilRanges { int n; int range[STENCIL_RANGES][2]; };

// values of the CSR -> stencil layout; one thread per row
__global__ void stencil_build_kernel(int n, const int *Ap, const double *Ax, const int *pat, const int *dict_ptr,
                                     const int *dict_slot, const unsigned *pat_mask, int nu, double *vals,
                                     void *mask, StencilRanges R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pat[i];
    unsigned pm = pat_mask[p];
    for (int r = 0; r < R.n; ++r)
        if (i >= R.range[r][0] && i < R.range[r][1]) pm = 0x80000000u;      // inside an uncovered range
    if (nu <= 7) static_cast<unsigned char *>(mask)[i] = (pm & 0x80000000u) ? (unsigned char)0x80 : (unsigned char)pm;
    else static_cast<unsigned *>(mask)[i] = pm;
    if (pm & 0x80000000u) return;                                       // row stays with the pattern kernel
    const int s = Ap[i], e = Ap[i + 1], d = dict_ptr[p];
    double *vp = vals + ((size_t)(i / WG) * nu) * WG + (i % WG);
    for (int k = s; k < e; ++k) vp[(size_t)dict_slot[d + (k - s)] * WG] = Ax[k];
}

int launch_stencil_build(const DevCsr &M, const int *dict_slot, const unsigned *pat_mask, hipStream_t st)
{
    const int n = M.nrows;
    StencilRanges R;
    R.n = M.st_nranges;
    for (int r = 0; r < STENCIL_RANGES; ++r) { R.range[r][0] = M.st_range[r][0]; R.range[r][1] = M.st_range[r][1]; }
    hipLaunchKernelGGL(stencil_build_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, M.Ap, M.Ax, M.pat,
                       M.dict_ptr, dict_slot, pat_mask, M.st_nu, M.st_vals, M.st_mask, R);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "stencil build launch", __FILE__, __LINE__);
    return 0;
}

// ---- value index (DevCsr::st_codes) ----
__global__ void value_scan_kernel(const double *vals, long count, unsigned long long *table, int *overflow)
{
    const unsigned long long EMPTY = ~0ULL;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < count; k += (long)gridDim.x * blockDim.x) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[k]);
        if (bits == EMPTY) { *overflow = 1; continue; }                  // that NaN pattern is the sentinel
        unsigned h = (unsigned)((bits * 0x9E3779B97F4A7C15ULL) >> 54);   // 10 bits
        bool done = false;
        for (int probe = 0; probe < 1024 && !done; ++probe, h = (h + 1) & 1023u) {
            unsigned long long cur = table[h];
            if (cur == bits) { done = true; break; }
            if (cur == EMPTY) {
                cur = atomicCAS(&table[h], EMPTY, bits);
                if (cur == EMPTY || cur == bits) { done = true; break; }
            }
        }
        if (!done) *overflow = 1;
    }
}
// vals are [block][slot][256]; the code of (block, slot, t) goes to byte (slot & 7) of word [block][slot >> 3][t]
// mask8 != null (stencils of up to 7 offsets): a slot the row does not store gets the reserved code 255, so that a kernel
// can tell "absent" (skip the term) from a stored 0.0 (add it) without reading the row mask (stencil_coded_kernel)
__global__ void value_encode_kernel(const double *vals, long count, const double *dict, int ndict, unsigned char *codes,
                                    int nu, const unsigned char *mask8, long nrows)
{
    const int nw = (nu + 7) >> 3;
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < count; k += (long)gridDim.x * blockDim.x) {
        const long long bits = __double_as_longlong(vals[k]);
        int lo = 0, hi = ndict - 1, at = 0;
        while (lo <= hi) {                                               // dict sorted by bit pattern
            const int mid = (lo + hi) >> 1;
            const long long d = __double_as_longlong(dict[mid]);
            if (d == bits) { at = mid; break; }
            if (d < bits) lo = mid + 1; else hi = mid - 1;
        }
        const long blk = k / ((long)nu * WG);
        const int rem = (int)(k - blk * (long)nu * WG);
        const int slot = rem / WG, t = rem - slot * WG;
        if (mask8) {
            const long row = blk * WG + t;
            const unsigned m = row < nrows ? mask8[row] : 0u;
            if ((m & 0x80u) || !((m >> slot) & 1u)) at = 255;
        }
        codes[((blk * nw + (slot >> 3)) * WG + t) * 8 + (slot & 7)] = (unsigned char)at;
    }
}
int launch_value_scan(const double *vals, long count, unsigned long long *table, int *overflow, hipStream_t st)
{
    hipLaunchKernelGGL(value_scan_kernel, dim3(8192), dim3(256), 0, st, vals, count, table, overflow);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "value scan launch", __FILE__, __LINE__);
    return 0;
}
int launch_value_encode(const double *vals, long count, const double *dict_sorted, int ndict, unsigned char *codes, int nu,
                        hipStream_t st, const unsigned char *mask8, long nrows)
{
    hipLaunchKernelGGL(value_encode_kernel, dim3(8192), dim3(256), 0, st, vals, count, dict_sorted, ndict, codes, nu, mask8, nrows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "value encode launch", __FILE__, __LINE__);
    return 0;
}

static int g_stencil_form = 1;
void set_stencil_form(int on) { g_stencil_form = on; ++g_config_epoch; }
bool stencil_enabled() { return g_stencil_form != 0; }

static int stencil_own_blocks(const StreamArgs &a)
{
    if (a.row_hi <= a.row_lo) return 0;
    return (a.row_hi - 1) / WG - a.row_lo / WG + 1;
}

static int pattern_rpb(const StreamArgs &a)
{
    int rpb = a.rows_per_wg;
    if (rpb < 1 || rpb > WG) rpb = WG;
    return rpb;
}

// workgroups of launch_stencil = partial sums SM_RESIDUAL_SUMSQ writes: the stencil launch's blocks,
// then those of the pattern launches over the uncovered row ranges
int stencil_blocks(const StreamArgs &a, const DevCsr &M)
{
    int nb = stencil_own_blocks(a);
    const int rpb = pattern_rpb(a);
    for (int r = 0; r < M.st_nranges; ++r) {
        const int lo = std::max(a.row_lo, M.st_range[r][0]), hi = std::min(a.row_hi, M.st_range[r][1]);
        if (hi > lo) nb += (hi - lo + rpb - 1) / rpb;
    }
    return nb;
}

template <int MODE>
static int launch_stencil_mode(const StreamArgs &a, const DevCsr &M, hipStream_t st)
{
    const int nb = stencil_own_blocks(a);
    if (nb <= 0) return 0;
    StreamArgs b = a;
    if (b.gscale == 0.0) b.gscale = 1.0;
    StencilArgs E;
    E.vals = M.st_vals; E.mask = M.st_mask; E.nu = M.st_nu; E.u0 = M.st_u0;
    E.codes = (M.st_vi_on && M.st_codes) ? M.st_codes : nullptr; E.dict = M.st_dict; E.ndict = M.st_ndict;
    E.ranges = M.st_nranges;
    E.blk_lo = a.row_lo / WG; E.nblocks = nb; E.ncols = M.ncols;
    E.period_blocks = E.seg_blocks = 0;
    for (int u = 0; u < STENCIL_MAX; ++u) E.off[u] = M.st_off[u];
    int grid = nb;
    if (g_xcd_period && M.period_rows > 0) {
        const int S = (M.period_rows + WG / 2) / WG;
        const int G = (S + 7) / 8;
        if (S >= 64 && nb >= 4 * S && 3.0 * G * WG * 8.0 <= 2.0e6) {
            E.period_blocks = S;
            E.seg_blocks = G;
            grid = 8 * G * ((nb + S - 1) / S);
        }
    }
    // two rows per lane needs the vectors it streams 16-byte aligned
    // Measured (tools/pairs_size_ab.py, tools/pairs_ab.py): faster from ~30 M rows up with stencils of up to 7 offsets
    // (500^3: 1.74 vs 1.82 ms); slower on small levels (4 M rows, Jacobi mode: +12 %) and with the 16-slot
    // instantiation on coarse levels (+40 % at 10 us), where a launch has too few waves to need fewer instructions.
    // g_stencil_pairs = 2 forces it.  (Coded values -- the value index -- keep one row per lane: measured faster there; stencil_coded_kernel.)
    const bool big = (long)(a.row_hi - a.row_lo) >= 30000000L && M.st_nu <= 7;
    static const int pairs_coded = std::getenv("AMG_STENCIL_PAIRS_CODED") ? std::atoi(std::getenv("AMG_STENCIL_PAIRS_CODED")) : 0;   // (A/B)
    const bool pairs = (g_stencil_pairs >= 2 || (g_stencil_pairs == 1 && big)) && M.st_nu <= 16 && (E.codes == nullptr || pairs_coded) &&
                       (((uintptr_t)b.b | (uintptr_t)b.out | (uintptr_t)b.v2 | (uintptr_t)b.xg) & 15u) == 0;
    if (pairs && M.st_nu <= 7)
        hipLaunchKernelGGL((stencil2_kernel<MODE, 8>), dim3(grid), dim3(WG2), 0, st, b, E, g_xcd_chunk);
    else if (pairs)
        hipLaunchKernelGGL((stencil2_kernel<MODE, 16>), dim3(grid), dim3(WG2), 0, st, b, E, g_xcd_chunk);
    else if (E.codes != nullptr && M.st_nu <= 7 && g_stencil_coded)
        hipLaunchKernelGGL((stencil_coded_kernel<MODE>), dim3(grid), dim3(WG), 0, st, b, E, g_xcd_chunk);
    else if (M.st_nu <= 7)
        hipLaunchKernelGGL((stencil_kernel<MODE, 8>), dim3(grid), dim3(WG), 0, st, b, E, g_xcd_chunk);
    else if (M.st_nu <= 16)
        hipLaunchKernelGGL((stencil_kernel<MODE, 16>), dim3(grid), dim3(WG), 0, st, b, E, g_xcd_chunk);
    else
        hipLaunchKernelGGL((stencil_kernel<MODE, 32>), dim3(grid), dim3(WG), 0, st, b, E, g_xcd_chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "stencil launch", __FILE__, __LINE__);
    // rows outside the stencil form
    const int rpb = pattern_rpb(a);
    int done = nb;
    for (int r = 0; r < M.st_nranges; ++r) {
        StreamArgs c = a;
        c.row_lo = std::max(a.row_lo, M.st_range[r][0]);
        c.row_hi = std::min(a.row_hi, M.st_range[r][1]);
        if (c.row_hi <= c.row_lo) continue;
        if (MODE == SM_RESIDUAL_SUMSQ) c.out2 = a.out2 + done;
        int rc = launch_pattern((StreamMode)MODE, c, M, st);
        if (rc != 0) return rc;
        done += (c.row_hi - c.row_lo + rpb - 1) / rpb;
    }
    return 0;
}

int launch_stencil(StreamMode mode, const StreamArgs &a, const DevCsr &M, hipStream_t st)
{
    switch (mode) {
    case SM_MATVEC: return launch_stencil_mode<SM_MATVEC>(a, M, st);
    case SM_MATVEC_ACC: return launch_stencil_mode<SM_MATVEC_ACC>(a, M, st);
    case SM_RESIDUAL: return launch_stencil_mode<SM_RESIDUAL>(a, M, st);
    case SM_POLY_STEP: return launch_stencil_mode<SM_POLY_STEP>(a, M, st);
    case SM_POLY_LAST: return launch_stencil_mode<SM_POLY_LAST>(a, M, st);
    case SM_JACOBI: return launch_stencil_mode<SM_JACOBI>(a, M, st);
    case SM_JACOBI_BSR1: return launch_stencil_mode<SM_JACOBI_BSR1>(a, M, st);
    case SM_RESIDUAL_SUMSQ: return launch_stencil_mode<SM_RESIDUAL_SUMSQ>(a, M, st);
    default: break;
    }
    set_error("launch_stencil: mode not supported");
    return -1;
}

// 16-bit column codes (DevCsr::Aj16): one workgroup per row block of the stream kernel
__global__ __launch_bounds__(WG) void index16_build_kernel(int nrows, int rpb, const int *Ap, const int *Aj,
                                                          unsigned short *Aj16, int *wg_base, unsigned char *wg_flag)
{
    __shared__ int sset[16];
    __shared__ int sover;
    const int t = threadIdx.x, wg = blockIdx.x;
    const int r0 = wg * rpb;
    const int nr = min(rpb, nrows - r0);
    const int kbeg = Ap[r0], kend = Ap[r0 + nr];
    if (t < 16) sset[t] = -1;
    if (t == 0) sover = 0;
    __syncthreads();
    for (int k = kbeg + t; k < kend; k += WG) {
        const int w = Aj[k] >> 12;
        bool placed = false;
        for (int s = 0; s < 16 && !placed; ++s) {
            int cur = atomicCAS(&sset[s], -1, w);          // claims an empty slot, else returns its window
            placed = (cur == -1 || cur == w);
        }
        if (!placed) sover = 1;
    }
    __syncthreads();
    if (sover) {
        if (t == 0) wg_flag[wg] = 0;
        return;
    }
    for (int k = kbeg + t; k < kend; k += WG) {
        const int c = Aj[k], w = c >> 12;
        int slot = 0;
        for (int s = 0; s < 16; ++s) if (sset[s] == w) slot = s;
        Aj16[k] = (unsigned short)((slot << 12) | (c & 4095));
    }
    if (t < 16) wg_base[wg * 16 + t] = sset[t] >= 0 ? (sset[t] << 12) : 0;
    if (t == 0) wg_flag[wg] = 1;
}

static int g_index16 = 0;   // opt-in: measured -6 % on R_0, -1 % on A_1, +8 % on P_0 at 500^3
void set_index16(int on) { g_index16 = on; ++g_config_epoch; }
bool index16_enabled() { return g_index16 != 0; }

int launch_index16_build(DevCsr &M, int rpb, hipStream_t st)
{
    const int nwg = (M.nrows + rpb - 1) / rpb;
    if (nwg <= 0) return 0;
    hipLaunchKernelGGL(index16_build_kernel, dim3(nwg), dim3(WG), 0, st, M.nrows, rpb, M.Ap, M.Aj, M.Aj16, M.wg_base,
                       M.wg_flag);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "index16 build launch", __FILE__, __LINE__);
    return 0;
}

template <int MODE>
static int launch_stream_mode(const StreamArgs &a, hipStream_t st)
{
    int rows = a.row_hi - a.row_lo;
    if (rows <= 0) return 0;
    int rpb = a.rows_per_wg;
    if (rpb < 1 || rpb > WG) rpb = WG;
    int nb = (rows + rpb - 1) / rpb;
    StreamArgs b = a;
    if (b.gscale == 0.0) b.gscale = 1.0;
    if (!g_index16 || (a.row_lo % rpb) != 0) b.Aj16 = nullptr;     // row blocks must be the coded ones
    if constexpr (!ModeTraits<MODE>::gs) {
        if (g_stream_pipe && g_stream_variant && !b.Aj16) {
            // persistent grid: as many workgroups as the chip holds at once (a multiple of the 8 XCDs)
            static int per_cu = 0;
            if (per_cu == 0) {
                int nbk = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, csr_stream_pipe_kernel<MODE>, WG, 0) != hipSuccess || nbk < 1) nbk = 4;
                per_cu = nbk;
            }
            static int ncu = 0;
            if (ncu == 0) {
                int dev = 0; hipDeviceProp_t pr;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
                if (ncu < 8) ncu = 256;
            }
            int G = per_cu * ncu;
            G -= G % 8;
            if (G > nb) G = nb;
            hipLaunchKernelGGL((csr_stream_pipe_kernel<MODE>), dim3(G), dim3(WG), 0, st, b, g_xcd_chunk, rpb, nb);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "csr_stream_pipe launch", __FILE__, __LINE__);
            return 0;
        }
    }
    if (g_stream_variant)
        hipLaunchKernelGGL((csr_stream_kernel<MODE, 1>), dim3(nb), dim3(WG), 0, st, b, g_xcd_chunk, rpb);
    else
        hipLaunchKernelGGL((csr_stream_kernel<MODE, 0>), dim3(nb), dim3(WG), 0, st, b, g_xcd_chunk, rpb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "csr_stream launch", __FILE__, __LINE__);
    return 0;
}

// ---------------------------------------------------------------------------
// One dependency level of a scheduled Gauss-Seidel sweep, at most GS_LEVEL_MAXWG workgroups: the launch-per-level
// path of levels too wide for a chain.  Such a launch is pure latency (a few thousand rows, ~5 us in
// csr_stream_kernel: row pointers -> entries -> gathered operands, three dependent round trips).  Here every
// workgroup's first and last entry position arrive in the KERNEL ARGUMENTS (the host knows the level-ordered copy's row
// pointers), so the entries are requested at once, beside the row pointers, row numbers and diagonal positions:
// two round trips.  Same tile, same left-to-right row sums as csr_stream_kernel<SM_GS / SM_GS_BSR1>.
// ---------------------------------------------------------------------------
constexpr int GS_LEVEL_MAXWG = 224;
struct LevelEntries { int e[GS_LEVEL_MAXWG + 1]; };

template <bool BSR1>
__global__ __launch_bounds__(WG) void gs_level_kernel(StreamArgs a, LevelEntries H, int rpb)
{
    __shared__ double sp[TILE];
    const int t = threadIdx.x;
    const int blk = blockIdx.x;
    const int r0 = a.row_lo + blk * rpb;
    const int nr = min(rpb, a.row_hi - r0);
    const int kbeg = H.e[blk], kend = H.e[blk + 1];
    constexpr int U = TILE / WG;
    // first tile's entries, then the row's own data (neither depends on the other)
    int c[U];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = kbeg + u * WG + t;
        const bool ok = k < kend;
        c[u] = ok ? a.Aj[k] : 0;
        v[u] = ok ? a.Ax[k] : 0.0;
    }
    int my_s = kend, my_e = kend, row = r0 + t, dpos = -1;
    if (t < nr) {
        my_s = a.Ap[r0 + t];
        my_e = a.Ap[r0 + t + 1];
        if (a.rowmap) row = a.rowmap[r0 + t];
        dpos = a.diagpos[r0 + t];
    }
    double gs_d = 0.0, gs_b = 0.0;
    double acc = 0.0;
    for (int tile_lo = kbeg; tile_lo < kend; tile_lo += TILE) {
        const int tile_hi = min(tile_lo + TILE, kend);
        if (tile_lo != kbeg) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = tile_lo + u * WG + t;
                const bool ok = k < tile_hi;
                c[u] = ok ? a.Aj[k] : 0;
                v[u] = ok ? a.Ax[k] : 0.0;
            }
        }
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = a.xg[c[u]];
        if (tile_lo == kbeg && t < nr) {              // second round trip, beside the gathers
            gs_d = (dpos >= 0) ? a.Ax[dpos] : 0.0;
            gs_b = a.b[row];
            if (BSR1) acc = gs_b;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) sp[u * WG + t] = v[u] * xv[u];
        __syncthreads();
        {
            const int s = max(my_s, tile_lo), e2 = min(my_e, tile_hi);
            constexpr int RB = AMG_ROWSUM_BATCH;
            for (int k = s; k < e2; k += RB) {
                double p[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) p[u] = sp[min(k + u, e2 - 1) - tile_lo];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const bool take = (k + u < e2) && (k + u != dpos);
                    const double nxt = BSR1 ? (acc - p[u]) : (acc + p[u]);
                    acc = take ? nxt : acc;
                }
            }
        }
        __syncthreads();
    }
    if (t >= nr) return;
    if (kbeg >= kend) {                                // a workgroup of empty rows never entered the loop
        gs_d = (dpos >= 0) ? a.Ax[dpos] : 0.0;
        gs_b = a.b[row];
        if (BSR1) acc = gs_b;
    }
    if (gs_d != 0.0) a.out[row] = BSR1 ? (acc / gs_d) : ((gs_b - acc) / gs_d);
}

static int g_gs_level_hint = 1;
void set_gs_level_hint(int on) { g_gs_level_hint = on; ++g_config_epoch; }

// gp_host: the row pointers of the launch's operator on the host (Schedule::gp_host), or null
int launch_gs_level(const StreamArgs &a, bool bsr1, const int *gp_host, hipStream_t st)
{
    const int rows = a.row_hi - a.row_lo;
    if (rows <= 0) return 0;
    int rpb = a.rows_per_wg;
    if (rpb < 1 || rpb > WG) rpb = WG;
    const int nb = (rows + rpb - 1) / rpb;
    if (!gp_host || !g_gs_level_hint || nb > GS_LEVEL_MAXWG || a.Aj16 != nullptr || (a.gscale != 0.0 && a.gscale != 1.0))
        return launch_stream(bsr1 ? SM_GS_BSR1 : SM_GS, a, st);
    LevelEntries H;
    for (int g = 0; g <= nb; ++g) H.e[g] = gp_host[std::min(a.row_lo + g * rpb, a.row_hi)];
    for (int g = nb + 1; g <= GS_LEVEL_MAXWG; ++g) H.e[g] = H.e[nb];
    if (bsr1) hipLaunchKernelGGL((gs_level_kernel<true>), dim3(nb), dim3(WG), 0, st, a, H, rpb);
    else hipLaunchKernelGGL((gs_level_kernel<false>), dim3(nb), dim3(WG), 0, st, a, H, rpb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gs level launch", __FILE__, __LINE__);
    return 0;
}

int stream_blocks(const StreamArgs &a)
{
    int rows = a.row_hi - a.row_lo;
    int rpb = a.rows_per_wg;
    if (rpb < 1 || rpb > WG) rpb = WG;
    return rows > 0 ? (rows + rpb - 1) / rpb : 0;
}

int launch_stream(StreamMode mode, const StreamArgs &a, hipStream_t st)
{
    switch (mode) {
    case SM_MATVEC: return launch_stream_mode<SM_MATVEC>(a, st);
    case SM_MATVEC_ACC: return launch_stream_mode<SM_MATVEC_ACC>(a, st);
    case SM_RESIDUAL: return launch_stream_mode<SM_RESIDUAL>(a, st);
    case SM_POLY_FIRST: return launch_stream_mode<SM_POLY_FIRST>(a, st);
    case SM_POLY_STEP: return launch_stream_mode<SM_POLY_STEP>(a, st);
    case SM_POLY_LAST: return launch_stream_mode<SM_POLY_LAST>(a, st);
    case SM_JACOBI: return launch_stream_mode<SM_JACOBI>(a, st);
    case SM_JACOBI_BSR1: return launch_stream_mode<SM_JACOBI_BSR1>(a, st);
    case SM_GS: return launch_stream_mode<SM_GS>(a, st);
    case SM_GS_BSR1: return launch_stream_mode<SM_GS_BSR1>(a, st);
    case SM_RESIDUAL_SUMSQ: return launch_stream_mode<SM_RESIDUAL_SUMSQ>(a, st);
    }
    set_error("launch_stream: bad mode");
    return -1;
}

// ---------------------------------------------------------------------------
// thread-per-row Jacobi for strided row ranges (amg_core.jacobi with a
// row_step other than +-1): same arithmetic, rows are independent.
// ---------------------------------------------------------------------------
__global__ void jacobi_rows_kernel(const int *Ap, const int *Aj, const double *Ax,
                                   const double *temp, const double *b, double *x, int row_start,
                                   int count, int row_step, double omega)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    int i = row_start + t * row_step;
    double rsum = 0.0, diag = 0.0;
    for (int jj = Ap[i]; jj < Ap[i + 1]; ++jj) {
        int j = Aj[jj];
        if (i == j) diag = Ax[jj];
        else rsum = rsum + Ax[jj] * temp[j];
    }
    if (diag != 0.0) {
        double q = (b[i] - rsum) / diag;
        double t1 = (1.0 - omega) * temp[i];
        double t2 = omega * q;
        x[i] = t1 + t2;
    }
}

int launch_jacobi_rows(const DevCsr &A, const double *temp, const double *b, double *x,
                       int row_start, int count, int row_step, double omega, hipStream_t st)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(jacobi_rows_kernel, dim3((count + 255) / 256), dim3(256), 0, st, A.Ap, A.Aj,
                       A.Ax, temp, b, x, row_start, count, row_step, omega);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "jacobi_rows launch", __FILE__, __LINE__);
    return 0;
}

// ---------------------------------------------------------------------------
// vector kernels (grid-stride, 16-byte accesses where alignment allows)
// ---------------------------------------------------------------------------
static inline int vec_grid(long n)
{
    long nb = (n + 255) / 256;
    if (nb > 256L * 16) nb = 256L * 16;
    if (nb < 1) nb = 1;
    return (int)nb;
}

__global__ void scale_kernel(double *out, const double *in, double c, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = c * in[i];
}
__global__ void fill_kernel(double *out, double v, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = v;
}
__global__ void sor_combine_kernel(double *x, const double *xold, double omega, long n)
{
    // relaxation.py:166-168: x *= omega; x_old *= (1-omega); x += x_old
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double a = x[i] * omega;
        double b = xold[i] * (1.0 - omega);
        x[i] = a + b;
    }
}
__global__ void axpy_inplace_kernel(double *x, const double *h, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = x[i] + h[i];
}
__global__ void sub_kernel(double *out, const double *a, const double *b, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = a[i] - b[i];
}
__global__ void copy_strided_kernel(double *dst, const double *src, int start, int count, int step)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) {
        int i = start + t * step;
        dst[i] = src[i];
    }
}

#define LAUNCH_CHECK(name)                                                      \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) return hip_fail(e__, name, __FILE__, __LINE__);  \
        return 0;                                                               \
    } while (0)

int launch_scale(double *out, const double *in, double c, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(vec_grid(n)), dim3(256), 0, st, out, in, c, n);
    LAUNCH_CHECK("scale");
}
int launch_fill(double *out, double v, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(vec_grid(n)), dim3(256), 0, st, out, v, n);
    LAUNCH_CHECK("fill");
}
__global__ void scale_add_kernel(double *p, double beta, const double *z, long n)
{
    // krylov/_cg.py:155-156: p *= beta; p += z
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double t = p[i] * beta;
        p[i] = t + z[i];
    }
}
int launch_scale_add(double *p, double beta, const double *z, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scale_add_kernel, dim3(vec_grid(n)), dim3(256), 0, st, p, beta, z, n);
    LAUNCH_CHECK("scale_add");
}
int launch_sor_combine(double *x, const double *xold, double omega, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sor_combine_kernel, dim3(vec_grid(n)), dim3(256), 0, st, x, xold, omega, n);
    LAUNCH_CHECK("sor_combine");
}
int launch_axpy_inplace(double *x, const double *h, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_inplace_kernel, dim3(vec_grid(n)), dim3(256), 0, st, x, h, n);
    LAUNCH_CHECK("axpy_inplace");
}
int launch_sub(double *out, const double *a, const double *b, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sub_kernel, dim3(vec_grid(n)), dim3(256), 0, st, out, a, b, n);
    LAUNCH_CHECK("sub");
}
int launch_copy_strided(double *dst, const double *src, int start, int count, int step, hipStream_t st)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(copy_strided_kernel, dim3((count + 255) / 256), dim3(256), 0, st, dst, src,
                       start, count, step);
    LAUNCH_CHECK("copy_strided");
}

// ---------------------------------------------------------------------------
// 2-norm: deterministic two-stage tree reduction (fixed grid, fixed order)
// ---------------------------------------------------------------------------
constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_stage1(const double *x, long n, double *partial)
{
    __shared__ double smem[4];
    double s0 = 0.0, s1 = 0.0;
    long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        double a = x[i], b = x[i + stride];
        s0 += a * a;
        s1 += b * b;
    }
    if (i < n) { double a = x[i]; s0 += a * a; }
    double r = block_reduce_sum(s0 + s1, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void sumsq_stage2(const double *partial, int np, double *result)
{
    __shared__ double smem[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    double r = block_reduce_sum(s, smem);
    if (threadIdx.x == 0) *result = sqrt(r);
}

__global__ __launch_bounds__(256) void sum_sqrt_kernel(const double *partial, long np, double *result)
{
    __shared__ double smem[4];
    double s = 0.0;
    for (long i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    double r = block_reduce_sum(s, smem);
    if (threadIdx.x == 0) *result = sqrt(r);
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const double *partial, long np, double *out)
{
    // fixed assignment of partials to workgroups and threads: deterministic
    __shared__ double smem[4];
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < np; i += (long)gridDim.x * blockDim.x) s += partial[i];
    double r = block_reduce_sum(s, smem);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}
int launch_sum_sqrt(const double *partial, long np, double *scratch512, double *result_dev, hipStream_t st)
{
    // np can be ~5e5 (one partial per workgroup of the 500^3 operator): reduce in two steps
    if (np > 4096 && scratch512) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(512), dim3(256), 0, st, partial, np, scratch512);
        hipLaunchKernelGGL(sum_sqrt_kernel, dim3(1), dim3(256), 0, st, scratch512, 512L, result_dev);
    } else {
        hipLaunchKernelGGL(sum_sqrt_kernel, dim3(1), dim3(256), 0, st, partial, np, result_dev);
    }
    LAUNCH_CHECK("sum_sqrt");
}
__global__ void axpy_scaled_kernel(double *x, const double *r, double c, long n)
{
    // relaxation.py:663,668 with one coefficient: h = c*r; x += h
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double h = c * r[i];
        x[i] = x[i] + h;
    }
}
int launch_axpy_scaled(double *x, const double *r, double c, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_scaled_kernel, dim3(vec_grid(n)), dim3(256), 0, st, x, r, c, n);
    LAUNCH_CHECK("axpy_scaled");
}

int launch_norm2(const double *x, long n, double *scratch, double *result_dev, hipStream_t st)
{
    int nb = (int)((n + 255) / 256);
    if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sumsq_stage1, dim3(nb), dim3(256), 0, st, x, n, scratch);
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, scratch, nb, result_dev);
    LAUNCH_CHECK("norm2");
}

// dot product, same deterministic two-stage scheme
__global__ __launch_bounds__(256) void dot_stage1(const double *x, const double *y, long n, double *partial)
{
    __shared__ double smem[4];
    double s0 = 0.0;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) s0 += x[i] * y[i];
    double r = block_reduce_sum(s0, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void sum_stage2(const double *partial, int np, double *result)
{
    __shared__ double smem[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += partial[i];
    double r = block_reduce_sum(s, smem);
    if (threadIdx.x == 0) *result = r;
}
int launch_dot(const double *x, const double *y, long n, double *scratch, double *result_dev, hipStream_t st)
{
    int nb = (int)((n + 255) / 256);
    if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(dot_stage1, dim3(nb), dim3(256), 0, st, x, y, n, scratch);
    hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, st, scratch, nb, result_dev);
    LAUNCH_CHECK("dot");
}
__global__ void axpy_kernel(double *w, const double *v, double a, long n)   // w = w - a*v
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        w[i] = w[i] - a * v[i];
}
// w = w - (sign * num / den) * v with the two scalars read from device memory (AMLI step lengths, multilevel.py:523-537:
// the quotient is formed in fp64 exactly as the host would, sign = +-1 is exact)
__global__ void axpy_ratio_kernel(double *w, const double *v, const double *num, const double *den, double sign, long n)
{
    const double a = sign * (*num / *den);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        w[i] = w[i] - a * v[i];
}
int launch_axmy_ratio(double *w, const double *v, const double *num, const double *den, double sign, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_ratio_kernel, dim3(vec_grid(n)), dim3(256), 0, st, w, v, num, den, sign, n);
    LAUNCH_CHECK("axmy ratio");
}
int launch_axmy(double *w, const double *v, double a, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(vec_grid(n)), dim3(256), 0, st, w, v, a, n);
    LAUNCH_CHECK("axmy");
}
__global__ void divide_kernel(double *w, double a, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        w[i] = w[i] / a;
}
int launch_divide(double *w, double a, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(divide_kernel, dim3(vec_grid(n)), dim3(256), 0, st, w, a, n);
    LAUNCH_CHECK("divide");
}
__global__ void mul_elem_kernel(double *w, const double *d, long n)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        w[i] = w[i] * d[i];
}
int launch_mul_elem(double *w, const double *d, long n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mul_elem_kernel, dim3(vec_grid(n)), dim3(256), 0, st, w, d, n);
    LAUNCH_CHECK("mul_elem");
}
__global__ void combine_kernel(double *out, const double *V, const double *coef, int m, long n, long ld)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += V[(long)k * ld + i] * coef[k];
        out[i] = s;
    }
}
int launch_combine(double *out, const double *V, const double *coef_dev, int m, long n, long ld, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(combine_kernel, dim3(vec_grid(n)), dim3(256), 0, st, out, V, coef_dev, m, n, ld);
    LAUNCH_CHECK("combine");
}

// ---------------------------------------------------------------------------
// coarse solve: x = M b, M dense n x n (n <= a few hundred), stored transposed
// so that lane i streams Mt[k*n+i] coalesced; strict left-to-right row sums
// ---------------------------------------------------------------------------
__global__ void dense_apply_kernel(const double *Mt, const double *b, double *x, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    int k = 0;
    for (; k + 8 <= n; k += 8) {           // 8 independent loads in flight, sums stay in order
        double m[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { m[u] = Mt[(long)(k + u) * n + i]; v[u] = b[k + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = s + m[u] * v[u];
    }
    for (; k < n; ++k) s = s + Mt[(long)k * n + i] * b[k];
    x[i] = s;
}

int launch_dense_apply(const double *Mt, const double *b, double *x, int n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dense_apply_kernel, dim3((n + 63) / 64), dim3(64), 0, st, Mt, b, x, n);
    LAUNCH_CHECK("dense_apply");
}

constexpr int MAXBS = 16;

// ---------------------------------------------------------------------------
// bsr_stream: the block / point-BSR smoothers with the matrix STREAMED.  A workgroup owns `rpb`
// consecutive block rows (rpb*bs <= 256); their blocks are one contiguous slice of the BSR data
// array, streamed through LDS in tiles of whole blocks with lane-contiguous loads (each entry
// multiplied by its operand x[col*bs + c] on the way in); then thread (block row, r) forms, block
// by block, v = sum_c p[r][c] left to right and folds v into its row sum -- exactly the order of
// gemm + "rsum += v" in relaxation.h:697-709, 787-799 (and "rsum -= Axloc", :138-146, 317-325).
// The small dense epilogue (Dinv*t, or the point sweep over the diagonal block) goes through LDS.
// ---------------------------------------------------------------------------
template <int BMODE, int BS>      // BS > 0: compile-time block size (index arithmetic folds); 0: a.bs
__global__ __launch_bounds__(WG) void bsr_stream_kernel(BsrStreamArgs a, int rpb, int xcd_chunk)
{
    __shared__ double sp[TILE];
    // block column of every block in the tile: TILE / bs^2 blocks at most (a tile holds whole blocks); sized exactly when
    // the block size is a template constant -- 20 KB of LDS per workgroup instead of 27, i.e. 8 resident workgroups
    // per CU instead of 5
    __shared__ int sbj[BS > 1 ? TILE / (BS * BS) + 2 : TILE];
    __shared__ int sAp[WG + 1];
    __shared__ double st[WG];

    const int t = threadIdx.x;
    const int bs = BS > 0 ? BS : a.bs, B2 = bs * bs;
    // whole passes: consecutive row blocks to one XCD (its L2 then serves the neighbouring rows' operands); the
    // epilogue of SM_RESIDUAL_SUMSQ indexes its partial by blockIdx, which only has to be a permutation
    const int blk = xcd_chunk > 0 ? remap_block(blockIdx.x, gridDim.x, xcd_chunk) : (int)blockIdx.x;
    const int r0 = a.brow_lo + blk * rpb;
    const int nr = min(rpb, a.brow_hi - r0);
    for (int i = t; i <= nr; i += WG) sAp[i] = a.Ap[r0 + i];
    __syncthreads();
    const int bbeg = sAp[0], bend = sAp[nr];
    const int tile_blocks = (TILE - 1) / B2;     // one spare slot: the 16-byte loads may start one entry early

    const int li = t / bs, r = t - li * bs;
    const bool active = li < nr;
    const int prow = r0 + li;                                  // position in the (permuted) operator
    const int brow = active ? (a.rowmap ? a.rowmap[prow] : prow) : 0;   // block row in x / b numbering
    const int my_s = active ? sAp[li] : bend, my_e = active ? sAp[li + 1] : bend;
    const long ib = (long)brow * bs;
    constexpr bool point = (BMODE == BM_BSR_JACOBI || BMODE == BM_BSR_GS);
    double rsum = 0.0;
    if (point && active) rsum = a.b[ib + r];
    long dptr = -1;

    for (int tb = bbeg; tb < bend; tb += tile_blocks) {
        const int te = min(tb + tile_blocks, bend);
        const long ebase = (long)tb * B2, ecount = (long)(te - tb) * B2;
        // block columns of the tile into LDS (one load per block, not per entry), the values in one batch
        // of independent requests, then all gathers, then the products
        // Values by 16-byte loads: lane t of pair-batch h holds the entries 2(h*WG + t) - off + {0, 1}, where
        // off = ebase & 1 moves the origin down to a 16-byte boundary (arrays are padded at both ends of
        // what is read: entry ebase-1 exists whenever off == 1).
        constexpr int U = TILE / WG;
        const int off = (int)(ebase & 1);
        const double *vbase = a.Ax + (ebase - off);
        double av[U];
        long qv[U];
#pragma unroll
        for (int h = 0; h < U / 2; ++h) {
            const long j2 = 2 * ((long)h * WG + t);                   // position in the shifted stream
            v2d v = v2d{0.0, 0.0};
            if (j2 < ecount + off) v = *reinterpret_cast<const v2d *>(vbase + j2);
            av[2 * h] = v.x; av[2 * h + 1] = v.y;
            qv[2 * h] = j2 - off; qv[2 * h + 1] = j2 + 1 - off;       // entry index inside the tile (may be -1 / >= ecount)
        }
        for (int lb = t; lb < te - tb; lb += WG) sbj[lb] = a.Aj[tb + lb];
        __syncthreads();
        int colv[U], cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = qv[u];
            colv[u] = 0; cv[u] = 0;
            if (q >= 0 && q < ecount) {
                const int lb = (int)(q / B2);
                cv[u] = (int)(q - (long)lb * B2) % bs;
                colv[u] = sbj[lb];
            }
        }
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = qv[u];
            xv[u] = (q >= 0 && q < ecount) ? a.xin[(long)colv[u] * bs + cv[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = qv[u];
            if (q >= 0 && q < ecount) {
                if (BMODE == BM_SPMV) sp[q] = av[u] * (a.gscale * xv[u]);
                else sp[q] = av[u] * xv[u];
            }
        }
        __syncthreads();
        if (BMODE != BM_SPMV) {
            // the relaxation kernels sum in two levels (relaxation.h: gemm gives v = sum_c p[r][c] left to right per
            // block, then rsum += v block by block): the per-block sums are independent -- ALL threads form them,
            // each writing v over the first product of its block row -- and only the short chain over the blocks
            // stays with the row's own thread
            const int nrow_t = (te - tb) * bs;
            for (int idx = t; idx < nrow_t; idx += WG) {
                const int lb = idx / bs, rr = idx - lb * bs;
                double *p = &sp[(long)lb * B2 + rr * bs];
                double v = 0.0;
                for (int c = 0; c < bs; ++c) v = v + p[c];
                p[0] = v;
            }
            __syncthreads();
        }
        if (active) {
            // (LDS reads four blocks at a time, then the adds strictly in order: a read, a wait and an add per
            // operand -- what the plain loop compiles to -- serialises the LDS latency)
            const int s = max(my_s, tb), e = min(my_e, te);
            if (BMODE == BM_SPMV && BS > 0) {
                // scipy bsr_matvec: ONE running sum per scalar row, across the blocks and the
                // columns inside each block -- the order of the expanded CSR row
                for (int j0 = s; j0 < e; j0 += 4) {
                    double q[4][BS > 0 ? BS : 1];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double *p = &sp[(long)(min(j0 + u, e - 1) - tb) * B2 + r * bs];
#pragma unroll
                        for (int c = 0; c < (BS > 0 ? BS : 1); ++c) q[u][c] = p[c];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool take = j0 + u < e;
#pragma unroll
                        for (int c = 0; c < (BS > 0 ? BS : 1); ++c) { const double nxt = rsum + q[u][c]; rsum = take ? nxt : rsum; }
                    }
                }
            } else if (BMODE == BM_SPMV) {
                for (int jj = s; jj < e; ++jj) {
                    const double *p = &sp[(long)(jj - tb) * B2 + r * bs];
                    for (int c = 0; c < bs; ++c) rsum = rsum + p[c];
                }
            } else {
                for (int j0 = s; j0 < e; j0 += 4) {
                    double v[4];
                    int bc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int lb = min(j0 + u, e - 1) - tb;
                        bc[u] = sbj[lb];
                        v[u] = sp[(long)lb * B2 + r * bs];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j0 + u >= e) continue;
                        if (bc[u] == brow) { dptr = (long)(j0 + u) * B2; continue; }
                        rsum = point ? (rsum - v[u]) : (rsum + v[u]);
                    }
                }
            }
        }
        __syncthreads();
    }

    if (BMODE == BM_SPMV) {
        if (a.smode == SM_RESIDUAL_SUMSQ) {
            // per-workgroup partial of ||b - A x||^2 (the outer residual norm), r stored only when asked for
            double sq = 0.0;
            if (active) {
                const double rr = a.b[ib + r] - rsum;
                sq = rr * rr;
                if (a.xout) a.xout[ib + r] = rr;
            }
            __syncthreads();
            const double tot = block_reduce_sum(sq, sp);
            if (t == 0) a.out2[blockIdx.x] = tot;
            return;
        }
        if (!active) return;
        const long i = ib + r;
        if (a.smode == SM_MATVEC) a.xout[i] = rsum;
        else if (a.smode == SM_MATVEC_ACC) a.xout[i] = a.xout[i] + rsum;
        else if (a.smode == SM_RESIDUAL) a.xout[i] = a.b[i] - rsum;
        else if (a.smode == SM_POLY_STEP) { double cr = a.c0 * a.b[i]; a.xout[i] = cr + rsum; }
        else if (a.smode == SM_POLY_LAST) { double cr = a.c0 * a.b[i]; double h = cr + rsum; a.xout[i] = a.v2[i] + h; }
        return;
    } else if (BMODE == BM_BLOCK_JACOBI || BMODE == BM_BLOCK_GS) {
        if (active) st[t] = a.b[ib + r] - rsum;
        __syncthreads();
        if (active) {
            const double *D = a.Dinv + (long)brow * B2 + r * bs;
            const double *tv = &st[li * bs];
            double v = 0.0;
            for (int c = 0; c < bs; ++c) v = v + D[c] * tv[c];
            if (BMODE == BM_BLOCK_JACOBI) {
                double t1 = (1.0 - a.omega) * a.xin[ib + r];
                double t2 = a.omega * v;
                a.xout[ib + r] = t1 + t2;
            } else {
                a.xout[ib + r] = v;
            }
        }
    } else if (BMODE == BM_BSR_JACOBI) {
        // point Jacobi over the diagonal block (relaxation.h:339-351): rows independent (they read temp)
        if (active && dptr != -1) {
            const int step = a.intra_reverse ? -1 : 1;
            const int k0 = a.intra_reverse ? bs - 1 : 0, k1 = a.intra_reverse ? -1 : bs;
            double diag = 1.0;
            for (int kk = k0; kk != k1; kk += step) {
                if (kk == r) diag = a.Ax[dptr + r * bs + kk];
                else rsum = rsum - a.Ax[dptr + r * bs + kk] * a.xin[ib + kk];
            }
            if (diag != 0.0) {
                double t1 = (1.0 - a.omega) * a.xin[ib + r];
                double t2 = (a.omega * rsum) / diag;
                a.xout[ib + r] = t1 + t2;
            }
        }
    } else {
        // point Gauss-Seidel over the diagonal block (relaxation.h:151-163): sequential inside the
        // block, so one thread per block row finishes it from the row sums parked in LDS
        if (active) st[t] = rsum;
        __syncthreads();
        if (active && r == 0 && dptr != -1) {
            const int step = a.intra_reverse ? -1 : 1;
            const int k0 = a.intra_reverse ? bs - 1 : 0, k1 = a.intra_reverse ? -1 : bs;
            for (int k = k0; k != k1; k += step) {
                double rs = st[li * bs + k];
                double diag = 1.0;
                for (int kk = k0; kk != k1; kk += step) {
                    if (k == kk) diag = a.Ax[dptr + k * bs + kk];
                    else rs = rs - a.Ax[dptr + k * bs + kk] * a.xout[ib + kk];
                }
                if (diag != 0.0) a.xout[ib + k] = rs / diag;
            }
        }
    }
}

// 0 never, 1 when it pays (blocks of 3x3 and larger: measured -19 % at bs = 3, -9 % at bs = 6, break-even at
// bs = 2 against the CSR expansion, tools/bsr_spmv_ab.py), 2 always
static int g_bsr_spmv = 1;
void set_bsr_spmv(int on) { g_bsr_spmv = on; ++g_config_epoch; }
bool bsr_spmv_enabled(int bs) { return g_bsr_spmv == 2 || (g_bsr_spmv == 1 && bs >= 3); }
bool bsr_spmv_supports(StreamMode mode)
{
    return mode == SM_MATVEC || mode == SM_MATVEC_ACC || mode == SM_RESIDUAL || mode == SM_POLY_STEP || mode == SM_POLY_LAST ||
           mode == SM_RESIDUAL_SUMSQ;
}

// block rows per workgroup: as many as fit 256 threads, fewer when that would stage more than about one LDS
// tile of products
static int bsr_rows_per_wg(const BsrStreamArgs &a, long nblocks_hint)
{
    const int rows = a.brow_hi - a.brow_lo;
    const int B2 = a.bs * a.bs;
    int rpb = WG / a.bs;
    if (nblocks_hint > 0 && rows > 0) {
        double per_row = (double)nblocks_hint * B2 / (double)rows;
        // Passes over ALL block rows (Jacobi-type sweeps, operator applications) want workgroups that fill most of
        // their 256 lanes in the row phase: 4 LDS tiles of products each (measured on the C5 operator, bs = 3:
        // block Jacobi 3.6 -> 4.2 TB/s, r = b - A x 4.5 -> 5.0 TB/s).  A level of a scheduled Gauss-Seidel sweep is a
        // small launch that needs many workgroups instead: half a tile each (tools/bsr_tile_sweep.py).
        const bool whole = a.rowmap == nullptr;
        const double target = whole ? 4.0 * g_tile_target : 0.5 * g_tile_target;
        int want = (int)(target / (per_row > 1.0 ? per_row : 1.0));
        if (want < 1) want = 1;
        if (want < rpb) rpb = want;
    }
    return rpb;
}

int bsr_stream_blocks(const BsrStreamArgs &a, long nblocks_hint)
{
    const int rows = a.brow_hi - a.brow_lo;
    if (rows <= 0 || a.bs < 1 || a.bs > MAXBS) return 0;
    const int rpb = bsr_rows_per_wg(a, nblocks_hint);
    return (rows + rpb - 1) / rpb;
}

int launch_bsr_stream(BlockMode m, const BsrStreamArgs &a, long nblocks_hint, hipStream_t st)
{
    const int rows = a.brow_hi - a.brow_lo;
    if (rows <= 0) return 0;
    if (a.bs > MAXBS || a.bs < 1) { set_error("block kernels support blocksize 1..16"); return -5; }
    const int rpb = bsr_rows_per_wg(a, nblocks_hint);
    dim3 g((rows + rpb - 1) / rpb), b(WG);
    const int chunk = (a.rowmap == nullptr && g.x >= 4096) ? g_xcd_chunk : 0;
#define BSR_LAUNCH(MODE, BSV) hipLaunchKernelGGL((bsr_stream_kernel<MODE, BSV>), g, b, 0, st, a, rpb, chunk)
#define BSR_BY_BS(MODE)                                 \
    switch (a.bs) {                                     \
    case 2: BSR_LAUNCH(MODE, 2); break;                 \
    case 3: BSR_LAUNCH(MODE, 3); break;                 \
    case 4: BSR_LAUNCH(MODE, 4); break;                 \
    case 6: BSR_LAUNCH(MODE, 6); break;                 \
    default: BSR_LAUNCH(MODE, 0); break;                \
    }
    switch (m) {
    case BM_BSR_JACOBI: BSR_BY_BS(BM_BSR_JACOBI); break;
    case BM_BLOCK_JACOBI: BSR_BY_BS(BM_BLOCK_JACOBI); break;
    case BM_BSR_GS: BSR_BY_BS(BM_BSR_GS); break;
    case BM_BLOCK_GS: BSR_BY_BS(BM_BLOCK_GS); break;
    case BM_SPMV: BSR_BY_BS(BM_SPMV); break;
    }
#undef BSR_BY_BS
#undef BSR_LAUNCH
    LAUNCH_CHECK("bsr_stream kernel");
}

}  // namespace amg
