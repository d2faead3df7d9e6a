// Sliced form of irregular operators (DevCsr::sl_*, SELL-64-sigma) and its application kernel.
//
// csr_stream_kernel serves every operator without grid structure (the Galerkin operators A_1, A_2, ..., restriction,
// prolongation).  A workgroup of it makes three DEPENDENT round trips -- row pointer, entries, gathered operands --
// with barriers in between, and rocprofv3 shows ~80 % of its wave cycles waiting: it is bound
// by that chain, not by bytes.  Here a wave owns a slice of 64 rows, a lane one row; the slice's entries are stored
// entry-major (entry k of the 64 rows side by side, padded to the slice's longest row), so a lane's loads are
// coalesced with its neighbours', need no row pointer and no LDS, and ALL of a row's entries (up to 32 per pass) are
// requested before any is consumed: two round trips per slice, no barrier.  Rows are sorted by length inside windows
// of 256 (restrictions: 64) rows (longest first, ties in row order) so that slices are nearly rectangular.
// Every lane adds ITS row's products in stored order with separate multiply and add: bit-identical to the CSR kernel
// (scipy csr_matvec / relaxation.h row loops) in every mode.
#include "hier.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace amg {

namespace {

constexpr int SL_C = 64;            // rows per slice = lanes per wave
constexpr int SL_SIGMA_SQUARE = 256;   // rows per sorting window: square operators (A_l)
constexpr int SL_SIGMA_RECT = 64;      //                          restrictions (the gathers reach into the finer level: keep neighbours together)
constexpr int SL_PASS = 32;         // entries of a row requested at once (16: measured slower)
constexpr int SL_WG = 256;          // four slices per workgroup

__device__ __forceinline__ int remap(int b, int nb, int chunk)       // kernels.hip remap_block: consecutive blocks to one XCD
{
    if (chunk <= 0) return b;
    const int gsz = 8 * chunk, g = b / gsz, base = g * gsz;
    const int n_g = min(gsz, nb - base), l = b - base;
    const int xcd = l & 7, j = l >> 3, q = n_g >> 3, r = n_g & 7;
    return base + xcd * q + min(xcd, r) + j;
}

// window sort: key = ((length + 1) << 11) | (window size - 1 - position in window), sorted DESCENDING = longest row
// first, ties in row order.  Bitonic network over the keys of a window in LDS.
// Window size (tools/sell_ab.py with differently compiled libraries, 300^3 / 400^3 / 500^3): the wider the window, the
// less padding but the further apart the rows that end up side by side in a slice -- their gathers then share fewer
// cache lines.  2048: A_1 -11 % at 500^3 but +2 % / +10 % at 400^3 / 300^3 against the CSR kernel and R_0 up to +46 %;
// 256: A_1 -9 .. -12 % at all three sizes; restrictions (which gather from the finer level) want 64.
template <int SL_SIGMA>
__global__ __launch_bounds__(1024) void sell_sort_kernel(int row_lo, int n, const int *Ap, int *sl_row, unsigned short *sl_len, int *slice_w)
{          // rows row_lo .. row_lo + n - 1
    __shared__ unsigned key[SL_SIGMA];
    static_assert(SL_SIGMA <= 2048 && (SL_SIGMA & (SL_SIGMA - 1)) == 0, "window: a power of two up to 2048 (11-bit position in the sort key)");
    const int w0 = blockIdx.x * SL_SIGMA;
    for (int q = threadIdx.x; q < SL_SIGMA; q += (int)blockDim.x) {
        const int i = w0 + q;
        const unsigned len = (i < n) ? (unsigned)(Ap[row_lo + i + 1] - Ap[row_lo + i]) : 0u;
        key[q] = (i < n) ? (((len + 1u) << 11) | (unsigned)(SL_SIGMA - 1 - q)) : 0u;      // slots past the last row: key 0, sorted last
    }
    __syncthreads();
    for (int k = 2; k <= SL_SIGMA; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = threadIdx.x; q < SL_SIGMA; q += (int)blockDim.x) {
                const int p = q ^ j;
                if (p > q) {
                    const unsigned a = key[q], b = key[p];
                    const bool desc = (q & k) == 0;
                    if (desc ? (a < b) : (a > b)) { key[q] = b; key[p] = a; }
                }
            }
            __syncthreads();
        }
    for (int q = threadIdx.x; q < SL_SIGMA; q += (int)blockDim.x) {
        const unsigned kq = key[q];
        const bool is_row = kq != 0u;
        const int pos = SL_SIGMA - 1 - (int)(kq & 2047u);
        const int len = is_row ? (int)(kq >> 11) - 1 : 0;
        const int slot = w0 + q;
        sl_row[slot] = is_row ? row_lo + w0 + pos : -1;
        sl_len[slot] = (unsigned short)len;
        if ((q & (SL_C - 1)) == 0) slice_w[slot / SL_C] = len;                    // the slice's longest row comes first
    }
}

// entries of the slices, entry-major; padding entries: column 0, value 0 (requested, never used)
__global__ __launch_bounds__(SL_WG) void sell_fill_kernel(int nslices, const int *Ap, const int *Aj, const double *Ax, const int *sl_row,
                                                         const unsigned short *sl_len, const long *sl_off, int *sl_col, double *sl_val)
{
    const int s = blockIdx.x * (SL_WG / SL_C) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= nslices) return;
    const long off = sl_off[s];
    const int w = (int)((sl_off[s + 1] - off) >> 6);
    const int slot = s * SL_C + lane;
    const int row = sl_row[slot], len = sl_len[slot];
    const int k0 = row >= 0 ? Ap[row] : 0;
    for (int k = 0; k < w; ++k) {
        const bool have = k < len;
        sl_col[off + (long)k * SL_C + lane] = have ? Aj[k0 + k] : 0;
        sl_val[off + (long)k * SL_C + lane] = have ? Ax[k0 + k] : 0.0;
    }
}

struct SellArgs {
    const int *row;
    const unsigned short *len;
    const long *off;
    const int *col;
    const double *val;
    int nslices;
    const unsigned *code;        // 16-bit column codes (DevCsr::sl_code) or null
    const int *org;
    const long *coff;
    const unsigned char *flag16;
};

// 16-bit codes of one slice (one wave per slice): the distinct values of column >> 12 over the slice's real entries,
// in order of first appearance (row of 64 entries by row, lanes in order) -- at most 16, else the slice keeps its
// 32-bit columns; then the codes, two per word.
__global__ __launch_bounds__(SL_WG) void sell_code_kernel(int nslices, const unsigned short *sl_len, const long *sl_off, const int *sl_col,
                                                         const long *sl_coff, unsigned *sl_code, int *sl_org, unsigned char *sl_flag16)
{
    const int s = blockIdx.x * (SL_WG / SL_C) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= nslices) return;
    const long off = sl_off[s];
    const int w = (int)((sl_off[s + 1] - off) >> 6);
    const int len = sl_len[s * SL_C + lane];
    int tab = -1;                                   // lane t < 16 holds window t's value of column >> 12
    int cnt = 0;
    bool ok = true;
    for (int k = 0; k < w && ok; ++k) {
        const bool real = k < len;
        const int hi = real ? (sl_col[off + (long)k * SL_C + lane] >> 12) : -1;
        bool pending = real;
        while (true) {
            // drop the lanes whose window is already in the table
            bool found = false;
            for (int t = 0; t < cnt; ++t) found |= (__shfl(tab, t, 64) == hi);
            pending = pending && !found;
            const unsigned long long m = __ballot(pending);
            if (m == 0ULL) break;
            if (cnt >= 16) { ok = false; break; }
            const int first = __ffsll((long long)m) - 1;
            const int nv = __shfl(hi, first, 64);
            if (lane == cnt) tab = nv;
            ++cnt;
        }
    }
    if (lane < 16) sl_org[s * 16 + lane] = (ok && lane < cnt) ? (tab << 12) : 0;
    if (lane == 0) sl_flag16[s] = ok ? 1 : 0;
    if (!ok) return;
    const long coff = sl_coff[s];
    for (int k = 0; k < w; k += 2) {
        unsigned word = 0;
        for (int h = 0; h < 2; ++h) {
            // (every lane runs the table look-up -- the shuffles read lanes 0 .. 15, which must be active; a padding
            //  entry's column is 0 and its code is dropped)
            const int c = (k + h < w) ? sl_col[off + (long)(k + h) * SL_C + lane] : 0;
            int slot = 0;
            for (int t = 0; t < cnt; ++t) slot = (__shfl(tab, t, 64) == (c >> 12)) ? t : slot;
            const unsigned code = (k + h < len) ? (((unsigned)slot << 12) | ((unsigned)c & 4095u)) : 0u;
            word |= code << (16 * h);
        }
        sl_code[coff + (long)(k >> 1) * SL_C + lane] = word;
    }
}

template <int MODE, bool IDX16>
__global__ __launch_bounds__(SL_WG) void sell_kernel(StreamArgs a, SellArgs S, int xcd_chunk)
{
    const int blk = remap((int)blockIdx.x, (int)gridDim.x, xcd_chunk);
    const int s = blk * (SL_WG / SL_C) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= S.nslices) return;
    const long off = S.off[s];
    const int w = (int)((S.off[s + 1] - off) >> 6);                 // uniform over the wave
    const int slot = s * SL_C + lane;
    const int row = S.row[slot];
    const int len = S.len[slot];
    const double gscale = a.gscale;
    // the epilogue's streamed operands do not depend on the row sum: requested first
    double ep_b = 0.0, ep_v2 = 0.0;
    if (row >= 0) {
        if (MODE == SM_RESIDUAL || MODE == SM_POLY_FIRST || MODE == SM_POLY_STEP || MODE == SM_POLY_LAST) ep_b = a.b[row];
        if (MODE == SM_POLY_LAST) ep_v2 = a.v2[row];
        if (MODE == SM_MATVEC_ACC) ep_v2 = a.out[row];
    }
    const int *cp = S.col + off + lane;
    const double *vp = S.val + off + lane;
    bool coded = false;
    int org = 0;
    const unsigned *wp = nullptr;
    if constexpr (IDX16) {
        coded = S.flag16[s] != 0;                                   // uniform over the wave
        if (coded) {
            org = S.org[s * 16 + (lane & 15)];
            wp = S.code + S.coff[s] + lane;
        }
    }
    double acc = 0.0;
    for (int k0 = 0; k0 < w; k0 += SL_PASS) {
        int c[SL_PASS];
        double v[SL_PASS], xv[SL_PASS];
        if (IDX16 && coded) {
            unsigned wd[SL_PASS / 2];
#pragma unroll
            for (int u = 0; u < SL_PASS; u += 2) {
                wd[u / 2] = 0u;
                if (k0 + u < w) wd[u / 2] = __builtin_nontemporal_load(&wp[(long)((k0 + u) >> 1) * SL_C]);
            }
#pragma unroll
            for (int u = 0; u < SL_PASS; ++u) {
                v[u] = 0.0;
                if (k0 + u < w) v[u] = __builtin_nontemporal_load(&vp[(long)(k0 + u) * SL_C]);
            }
#pragma unroll
            for (int u = 0; u < SL_PASS; ++u) {
                const unsigned code = (wd[u / 2] >> (16 * (u & 1))) & 0xffffu;
                c[u] = __shfl(org, (int)(code >> 12), 64) + (int)(code & 4095u);
            }
        } else {
#pragma unroll
            for (int u = 0; u < SL_PASS; ++u) {
                c[u] = 0; v[u] = 0.0;
                if (k0 + u < w) {                                       // uniform: the whole wave requests or skips
                    c[u] = __builtin_nontemporal_load(&cp[(long)(k0 + u) * SL_C]);
                    v[u] = __builtin_nontemporal_load(&vp[(long)(k0 + u) * SL_C]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SL_PASS; ++u) xv[u] = (k0 + u < len) ? a.xg[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < SL_PASS; ++u) {
            const double pr = v[u] * (gscale * xv[u]);
            const double nxt = acc + pr;
            acc = (k0 + u < len) ? nxt : acc;
        }
    }
    if (row < 0) return;
    if (MODE == SM_MATVEC) {
        __builtin_nontemporal_store(acc, &a.out[row]);
    } else if (MODE == SM_MATVEC_ACC) {
        __builtin_nontemporal_store(ep_v2 + acc, &a.out[row]);
    } else if (MODE == SM_RESIDUAL) {
        __builtin_nontemporal_store(ep_b - acc, &a.out[row]);
    } else if (MODE == SM_POLY_FIRST) {
        const double r = ep_b - acc;
        a.out[row] = r;
        a.out2[row] = a.c0 * r;
    } else if (MODE == SM_POLY_STEP) {
        const double cr = a.c0 * ep_b;
        a.out[row] = cr + acc;
    } else if (MODE == SM_POLY_LAST) {
        const double cr = a.c0 * ep_b;
        const double h = cr + acc;
        __builtin_nontemporal_store(ep_v2 + h, &a.out[row]);
    }
}

int g_sell = 1;
int g_sell_idx16 = std::getenv("AMG_SELL_IDX16") ? std::atoi(std::getenv("AMG_SELL_IDX16")) : 1;

}   // namespace

void free_sell(DevCsr &M)
{
    if (M.sl_row) hipFree(M.sl_row);
    if (M.sl_len) hipFree(M.sl_len);
    if (M.sl_off) hipFree(M.sl_off);
    if (M.sl_col) hipFree(M.sl_col);
    if (M.sl_val) hipFree(M.sl_val);
    if (M.sl_code) hipFree(M.sl_code);
    if (M.sl_org) hipFree(M.sl_org);
    if (M.sl_coff) hipFree(M.sl_coff);
    if (M.sl_flag16) hipFree(M.sl_flag16);
    M.sl_row = nullptr; M.sl_len = nullptr; M.sl_off = nullptr; M.sl_col = nullptr; M.sl_val = nullptr;
    M.sl_code = nullptr; M.sl_org = nullptr; M.sl_coff = nullptr; M.sl_flag16 = nullptr; M.sl_frac16 = 0.0;
    M.sl_nslices = 0; M.sl_entries = 0; M.sl_lo = M.sl_hi = 0;
}

bool sell_supports(StreamMode mode)
{
    return mode == SM_MATVEC || mode == SM_MATVEC_ACC || mode == SM_RESIDUAL || mode == SM_POLY_FIRST || mode == SM_POLY_STEP ||
           mode == SM_POLY_LAST;
}
bool sell_enabled() { return g_sell != 0; }
void set_sell_form(int on) { g_sell = on; bump_config_epoch(); }
void set_sell_index16(int on) { g_sell_idx16 = on; bump_config_epoch(); }
bool sell_index16_enabled() { return g_sell_idx16 != 0; }

int launch_sell(StreamMode mode, const StreamArgs &a, const DevCsr &M, hipStream_t st)
{
    StreamArgs b = a;
    if (b.gscale == 0.0) b.gscale = 1.0;
    const bool idx16 = M.sl_code != nullptr && g_sell_idx16 != 0;
    SellArgs S{M.sl_row, M.sl_len, M.sl_off, M.sl_col, M.sl_val, M.sl_nslices, M.sl_code, M.sl_org, M.sl_coff, M.sl_flag16};
    const int grid = (M.sl_nslices + SL_WG / SL_C - 1) / (SL_WG / SL_C);
    // consecutive workgroups per XCD (speed only).  Measured at 400^3: 8 .. 128 equal within 1 %, 0 (round-robin over
    // the XCDs) 20 % slower, 512 2 % slower.  AMG_SELL_CHUNK overrides for A/B runs.
    static const int chunk_env = std::getenv("AMG_SELL_CHUNK") ? std::atoi(std::getenv("AMG_SELL_CHUNK")) : 32;
    const int chunk = grid >= 4096 ? chunk_env : 0;
    switch (mode) {
#define SELL_CASE(MODE) case MODE: if (idx16) hipLaunchKernelGGL((sell_kernel<MODE, true>), dim3(grid), dim3(SL_WG), 0, st, b, S, chunk); \
                        else hipLaunchKernelGGL((sell_kernel<MODE, false>), dim3(grid), dim3(SL_WG), 0, st, b, S, chunk); break
    SELL_CASE(SM_MATVEC);
    SELL_CASE(SM_MATVEC_ACC);
    SELL_CASE(SM_RESIDUAL);
    SELL_CASE(SM_POLY_FIRST);
    SELL_CASE(SM_POLY_STEP);
    SELL_CASE(SM_POLY_LAST);
#undef SELL_CASE
    default: set_error("launch_sell: mode not supported"); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sell launch", __FILE__, __LINE__);
    return 0;
}

// Built from the CSR arrays in HBM.  Only where it pays (measured on the 500^3 hierarchy, tools/sell_ab.py and the
// kernel trace): rows of 8 to 48 entries on average -- short rows (P_0: 3.4 entries) were 2x SLOWER here (a wave's 64 rows
// carry too few entries for its scattered row bookkeeping and result accesses) and stream well through the CSR kernel; longer rows (R_1: 150 entries, A_2: 67) need several dependent passes per
// slice with few slices to overlap them and were 5-70 % slower -- at least 2^16 rows, no row longer than 2046
// entries, and at most 15 % padding.
int build_sell(DevCsr &M, long *acct, int row_lo, int row_hi)
{
    const char *env = std::getenv("AMG_SELL");
    if (env && std::atoi(env) == 0) return 0;
    if (row_hi < 0) row_hi = M.nrows;
    if (row_lo < 0 || row_hi > M.nrows || row_hi <= row_lo || !M.Ap || !M.Aj || !M.Ax) return 0;
    const int n = row_hi - row_lo;
    const int sigma = (M.ncols >= 2L * M.nrows) ? SL_SIGMA_RECT : SL_SIGMA_SQUARE;
    const int nwin = (n + sigma - 1) / sigma;
    const int nslices = nwin * (sigma / SL_C);
    int *row = nullptr, *w_dev = nullptr;
    unsigned short *len = nullptr;
    // longest row must fit the 11 + 21-bit sort key and the 16-bit length; average row length 8 .. 48 (see above)
    long nnz_range = 0;
    {
        std::vector<int> hp((size_t)n + 1);
        AMG_HIP(hipMemcpy(hp.data(), M.Ap + row_lo, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost));
        int longest = 0;
        for (int i = 0; i < n; ++i) longest = std::max(longest, hp[(size_t)i + 1] - hp[(size_t)i]);
        if (longest > 2046) return 0;
        nnz_range = (long)hp[(size_t)n] - hp[0];
    }
    if (n < (1 << 16) || nnz_range < 8L * n || nnz_range > 48L * n) return 0;
    AMG_HIP(hipMalloc((void **)&row, sizeof(int) * (size_t)nslices * SL_C));
    AMG_HIP(hipMalloc((void **)&len, sizeof(unsigned short) * (size_t)nslices * SL_C));
    AMG_HIP(hipMalloc((void **)&w_dev, sizeof(int) * (size_t)nslices));
    if (sigma == SL_SIGMA_RECT) hipLaunchKernelGGL((sell_sort_kernel<SL_SIGMA_RECT>), dim3(nwin), dim3(64), 0, nullptr, row_lo, n, M.Ap, row, len, w_dev);
    else hipLaunchKernelGGL((sell_sort_kernel<SL_SIGMA_SQUARE>), dim3(nwin), dim3(256), 0, nullptr, row_lo, n, M.Ap, row, len, w_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sell sort launch", __FILE__, __LINE__);
    std::vector<int> hw((size_t)nslices);
    AMG_HIP(hipMemcpy(hw.data(), w_dev, sizeof(int) * (size_t)nslices, hipMemcpyDeviceToHost));
    hipFree(w_dev);
    std::vector<long> off((size_t)nslices + 1);
    off[0] = 0;
    for (int s = 0; s < nslices; ++s) off[(size_t)s + 1] = off[(size_t)s] + (long)hw[(size_t)s] * SL_C;
    const long entries = off[(size_t)nslices];
    if ((double)entries > 1.15 * (double)nnz_range) { hipFree(row); hipFree(len); return 0; }
    long *off_dev = nullptr;
    int *col = nullptr;
    double *val = nullptr;
    AMG_HIP(hipMalloc((void **)&off_dev, sizeof(long) * ((size_t)nslices + 1)));
    AMG_HIP(hipMalloc((void **)&col, sizeof(int) * (size_t)std::max(entries, 1L)));
    AMG_HIP(hipMalloc((void **)&val, sizeof(double) * (size_t)std::max(entries, 1L)));
    AMG_HIP(hipMemcpy(off_dev, off.data(), sizeof(long) * ((size_t)nslices + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sell_fill_kernel, dim3((nslices + SL_WG / SL_C - 1) / (SL_WG / SL_C)), dim3(SL_WG), 0, nullptr, nslices, M.Ap, M.Aj, M.Ax,
                       row, len, off_dev, col, val);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sell fill launch", __FILE__, __LINE__);
    AMG_HIP(hipDeviceSynchronize());
    M.sl_row = row; M.sl_len = len; M.sl_off = off_dev; M.sl_col = col; M.sl_val = val;
    M.sl_nslices = nslices; M.sl_entries = entries; M.sl_lo = row_lo; M.sl_hi = row_hi;
    if (g_sell_idx16 && M.ncols >= 4096) {
        // 16-bit column codes, two per word: code words of slice s start at coff[s] (ceil(width / 2) rows of 64 words)
        std::vector<long> coff((size_t)nslices + 1);
        coff[0] = 0;
        for (int s = 0; s < nslices; ++s) coff[(size_t)s + 1] = coff[(size_t)s] + (long)((hw[(size_t)s] + 1) / 2) * SL_C;
        AMG_HIP(hipMalloc((void **)&M.sl_coff, sizeof(long) * ((size_t)nslices + 1)));
        AMG_HIP(hipMalloc((void **)&M.sl_code, sizeof(unsigned) * (size_t)std::max(coff[(size_t)nslices], 1L)));
        AMG_HIP(hipMalloc((void **)&M.sl_org, sizeof(int) * (size_t)nslices * 16));
        AMG_HIP(hipMalloc((void **)&M.sl_flag16, (size_t)nslices));
        AMG_HIP(hipMemcpy(M.sl_coff, coff.data(), sizeof(long) * ((size_t)nslices + 1), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(sell_code_kernel, dim3((nslices + SL_WG / SL_C - 1) / (SL_WG / SL_C)), dim3(SL_WG), 0, nullptr, nslices, len, off_dev, col,
                           M.sl_coff, M.sl_code, M.sl_org, M.sl_flag16);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "sell code launch", __FILE__, __LINE__);
        std::vector<unsigned char> hf((size_t)nslices);
        AMG_HIP(hipMemcpy(hf.data(), M.sl_flag16, (size_t)nslices, hipMemcpyDeviceToHost));
        long coded = 0;
        for (unsigned char f : hf) coded += f ? 1 : 0;
        M.sl_frac16 = nslices ? (double)coded / (double)nslices : 0.0;
        if (acct) *acct += (long)(4L * coff[(size_t)nslices] + 65L * nslices + 8L * (nslices + 1));
        if (M.sl_frac16 < 0.5) {                                    // mostly fall-back slices: not worth the second array
            if (acct) *acct -= (long)(4L * coff[(size_t)nslices] + 65L * nslices + 8L * (nslices + 1));
            hipFree(M.sl_code); hipFree(M.sl_org); hipFree(M.sl_coff); hipFree(M.sl_flag16);
            M.sl_code = nullptr; M.sl_org = nullptr; M.sl_coff = nullptr; M.sl_flag16 = nullptr; M.sl_frac16 = 0.0;
        }
    }
    if (acct) *acct += (long)(sizeof(int) * (size_t)nslices * SL_C + sizeof(unsigned short) * (size_t)nslices * SL_C +
                              sizeof(long) * ((size_t)nslices + 1) + 12L * entries);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same for block operators (BASELINE configuration C5: BSR 3x3): DevBsr::bsl_*.  Block rows are sorted by block
// count inside windows of 32 slices, a slice holds NBR = 64 / bs block rows, a lane one SCALAR row (lane = block row
// in slice * bs + r).  Block k of the slice: its NBR block columns side by side, then its values as bs planes
// [c][lane] -- a lane's loads are coalesced with its neighbours', the three lanes of a block row read the same column
// and the same operands.  Arithmetic and order are bsr_stream_kernel's: BM_SPMV one running sum per scalar row across
// blocks and columns (scipy bsr_matvec); block Jacobi v = sum_c a[r][c] x[c] from 0 per block, rsum += v block by
// block, diagonal block skipped, then Dinv * (b - rsum) and the weighted update (relaxation.h:686-720).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int BSL_WIN = 32;          // slices per sorting window
constexpr int BSL_PASS = 8;          // blocks of a row requested at once

// windows: consecutive runs of WIN block rows, or (win_start / win_len given) listed runs of at most WIN block rows --
// the level-ordered copy of a Gauss-Seidel schedule is cut so that no window, hence no slice, straddles two levels
template <int NBR>
__global__ __launch_bounds__(1024) void bsell_sort_kernel(int nb, const int *Ap, int *brow, unsigned short *blen, int *slice_w,
                                                          const int *win_start, const int *win_len)
{
    constexpr int WIN = BSL_WIN * NBR;            // block rows per window (<= 1024)
    __shared__ unsigned key[1024];
    const int w0 = win_start ? win_start[blockIdx.x] : blockIdx.x * WIN;
    const int cnt = win_start ? win_len[blockIdx.x] : min(WIN, nb - w0);
    const int q = threadIdx.x;
    {
        const int i = w0 + q;
        const bool real = q < cnt && i < nb;
        const unsigned len = real ? (unsigned)(Ap[i + 1] - Ap[i]) : 0u;
        key[q] = real ? (((len + 1u) << 10) | (unsigned)(1023 - q)) : 0u;
    }
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = q ^ j;
            if (p > q) {
                const unsigned a = key[q], b = key[p];
                const bool desc = (q & k) == 0;
                if (desc ? (a < b) : (a > b)) { key[q] = b; key[p] = a; }
            }
            __syncthreads();
        }
    if (q < WIN) {
        const unsigned kq = key[q];
        const bool is_row = kq != 0u;
        const int pos = 1023 - (int)(kq & 1023u);
        const int len = is_row ? (int)(kq >> 10) - 1 : 0;
        const int slot = blockIdx.x * WIN + q;
        brow[slot] = is_row ? w0 + pos : -1;
        blen[slot] = (unsigned short)len;
        if (q % NBR == 0) slice_w[slot / NBR] = len;
    }
}

template <int BS>
__global__ __launch_bounds__(256) void bsell_fill_kernel(int nslices, const int *Ap, const int *Aj, const double *Ax, const int *brow,
                                                         const unsigned short *blen, const long *off, int *col, double *val)
{
    constexpr int NBR = 64 / BS, LW = NBR * BS;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= nslices || lane >= LW) return;
    const long o = off[s];
    const int w = (int)(off[s + 1] - o);
    const int br = lane / BS, r = lane - br * BS;
    const int row = brow[s * NBR + br], len = blen[s * NBR + br];
    const long k0 = row >= 0 ? Ap[row] : 0;
    for (int k = 0; k < w; ++k) {
        const bool have = k < len;
        if (r == 0) col[(o + k) * NBR + br] = have ? Aj[k0 + k] : 0;
        for (int c = 0; c < BS; ++c)
            val[((o + k) * BS + c) * LW + lane] = have ? Ax[(k0 + k) * (BS * BS) + r * BS + c] : 0.0;
    }
}

struct BsellArgs {
    const int *brow; const unsigned short *blen; const long *off; const int *col; const double *val; int nslices;
};

template <int BMODE, int BS>
__global__ __launch_bounds__(256) void bsell_kernel(BsrStreamArgs a, BsellArgs S, int xcd_chunk, int slice_lo, int slice_n)
{
    constexpr int NBR = 64 / BS, LW = NBR * BS, B2 = BS * BS;
    const int blk = remap((int)blockIdx.x, (int)gridDim.x, xcd_chunk);
    const int sl = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sl >= slice_n) return;
    const int s = slice_lo + sl;
    const long o = S.off[s];
    const int w = (int)(S.off[s + 1] - o);                          // uniform over the wave
    const bool lane_ok = lane < LW;
    const int br = lane_ok ? lane / BS : 0, r = lane_ok ? lane - br * BS : 0;
    const int brow = lane_ok ? S.brow[s * NBR + br] : -1;
    const int len = lane_ok ? (int)S.blen[s * NBR + br] : 0;
    const bool active = brow >= 0;
    const long ib = (long)brow * BS;
    const double gscale = (a.gscale == 0.0) ? 1.0 : a.gscale;
    double ep_b = 0.0, ep_v2 = 0.0, ep_x = 0.0;
    if (active) {
        if (BMODE == BM_BLOCK_JACOBI) { ep_b = a.b[ib + r]; ep_x = a.xin[ib + r]; }
        else if (BMODE == BM_BLOCK_GS) ep_b = a.b[ib + r];
        else {
            if (a.smode == SM_RESIDUAL || a.smode == SM_POLY_STEP || a.smode == SM_POLY_LAST) ep_b = a.b[ib + r];
            if (a.smode == SM_POLY_LAST) ep_v2 = a.v2[ib + r];
            if (a.smode == SM_MATVEC_ACC) ep_v2 = a.xout[ib + r];
        }
    }
    const int *cp = S.col + o * NBR + br;
    const double *vp = S.val + o * BS * LW + lane;
    double rsum = 0.0;
    for (int k0 = 0; k0 < w; k0 += BSL_PASS) {
        int bc[BSL_PASS];
        double v[BSL_PASS][BS], xv[BSL_PASS][BS];
#pragma unroll
        for (int u = 0; u < BSL_PASS; ++u) {
            bc[u] = 0;
#pragma unroll
            for (int c = 0; c < BS; ++c) v[u][c] = 0.0;
            if (k0 + u < w && lane_ok) {                            // (k0 + u < w is uniform)
                bc[u] = cp[(long)(k0 + u) * NBR];
#pragma unroll
                for (int c = 0; c < BS; ++c) v[u][c] = __builtin_nontemporal_load(&vp[((long)(k0 + u) * BS + c) * LW]);
            }
        }
#pragma unroll
        for (int u = 0; u < BSL_PASS; ++u)
#pragma unroll
            for (int c = 0; c < BS; ++c) xv[u][c] = (k0 + u < len) ? a.xin[(long)bc[u] * BS + c] : 0.0;
#pragma unroll
        for (int u = 0; u < BSL_PASS; ++u) {
            const bool take = k0 + u < len;
            if (BMODE == BM_SPMV) {
#pragma unroll
                for (int c = 0; c < BS; ++c) { const double nxt = rsum + v[u][c] * (gscale * xv[u][c]); rsum = take ? nxt : rsum; }
            } else {
                double vb = 0.0;
#pragma unroll
                for (int c = 0; c < BS; ++c) vb = vb + v[u][c] * xv[u][c];
                const double nxt = rsum + vb;
                rsum = (take && bc[u] != brow) ? nxt : rsum;        // the diagonal block is not part of the sum
            }
        }
    }
    if (BMODE == BM_SPMV) {
        if (!active) return;
        const long i = ib + r;
        if (a.smode == SM_MATVEC) a.xout[i] = rsum;
        else if (a.smode == SM_MATVEC_ACC) a.xout[i] = ep_v2 + rsum;
        else if (a.smode == SM_RESIDUAL) a.xout[i] = ep_b - rsum;
        else if (a.smode == SM_POLY_STEP) { const double cr = a.c0 * ep_b; a.xout[i] = cr + rsum; }
        else if (a.smode == SM_POLY_LAST) { const double cr = a.c0 * ep_b; const double h = cr + rsum; a.xout[i] = ep_v2 + h; }
        return;
    }
    // block Jacobi: x_i = (1 - omega) temp_i + omega * Dinv_i (b_i - rsum); the lanes of a block row exchange their t
    const double t = ep_b - rsum;
    double vD = 0.0;
#pragma unroll
    for (int c = 0; c < BS; ++c) {
        const double tc = __shfl(t, br * BS + c, 64);
        const double d = active ? a.Dinv[(long)brow * B2 + r * BS + c] : 0.0;
        vD = vD + d * tc;
    }
    if (active) {
        if (BMODE == BM_BLOCK_GS) a.xout[ib + r] = vD;                  // relaxation.h:756-810: x_i = Dinv_i (b_i - rsum)
        else {
            const double t1 = (1.0 - a.omega) * ep_x;
            const double t2 = a.omega * vD;
            a.xout[ib + r] = t1 + t2;
        }
    }
}

__global__ void bsell_remap_kernel(int nslots, const int *rowmap, int *brow)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nslots && brow[k] >= 0) brow[k] = rowmap[brow[k]];
}

template <int BS>
int build_bsell_bs(DevBsr &M, long *acct, const std::vector<int> *level_ptr = nullptr, const int *rowmap_dev = nullptr,
                   std::vector<int> *level_slice = nullptr)
{
    constexpr int NBR = 64 / BS, WIN = BSL_WIN * NBR;
    const int nb = M.nbrows;
    std::vector<int> ws, wl;
    if (level_ptr) {
        const int nl = (int)level_ptr->size() - 1;
        level_slice->assign((size_t)nl + 1, 0);
        for (int l = 0; l < nl; ++l) {
            (*level_slice)[(size_t)l] = (int)ws.size() * BSL_WIN;
            for (int w0 = (*level_ptr)[(size_t)l]; w0 < (*level_ptr)[(size_t)l + 1]; w0 += WIN) {
                ws.push_back(w0);
                wl.push_back(std::min(WIN, (*level_ptr)[(size_t)l + 1] - w0));
            }
        }
        (*level_slice)[(size_t)nl] = (int)ws.size() * BSL_WIN;
    }
    const int nwin = level_ptr ? (int)ws.size() : (nb + WIN - 1) / WIN;
    const int nslices = nwin * BSL_WIN;
    int *ws_dev = nullptr, *wl_dev = nullptr;
    if (level_ptr && nwin > 0) {
        AMG_HIP(hipMalloc((void **)&ws_dev, sizeof(int) * (size_t)nwin));
        AMG_HIP(hipMalloc((void **)&wl_dev, sizeof(int) * (size_t)nwin));
        AMG_HIP(hipMemcpy(ws_dev, ws.data(), sizeof(int) * (size_t)nwin, hipMemcpyHostToDevice));
        AMG_HIP(hipMemcpy(wl_dev, wl.data(), sizeof(int) * (size_t)nwin, hipMemcpyHostToDevice));
    }
    {
        std::vector<int> hp((size_t)nb + 1);
        AMG_HIP(hipMemcpy(hp.data(), M.Ap, sizeof(int) * ((size_t)nb + 1), hipMemcpyDeviceToHost));
        int longest = 0;
        for (int i = 0; i < nb; ++i) longest = std::max(longest, hp[(size_t)i + 1] - hp[(size_t)i]);
        if (longest > 60000) return 0;
    }
    int *brow = nullptr, *w_dev = nullptr;
    unsigned short *blen = nullptr;
    AMG_HIP(hipMalloc((void **)&brow, sizeof(int) * (size_t)nslices * NBR));
    AMG_HIP(hipMalloc((void **)&blen, sizeof(unsigned short) * (size_t)nslices * NBR));
    AMG_HIP(hipMalloc((void **)&w_dev, sizeof(int) * (size_t)nslices));
    hipLaunchKernelGGL((bsell_sort_kernel<NBR>), dim3(nwin), dim3(1024), 0, nullptr, nb, M.Ap, brow, blen, w_dev, (const int *)ws_dev, (const int *)wl_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "bsell sort launch", __FILE__, __LINE__);
    std::vector<int> hw((size_t)nslices);
    AMG_HIP(hipMemcpy(hw.data(), w_dev, sizeof(int) * (size_t)nslices, hipMemcpyDeviceToHost));
    hipFree(w_dev);
    std::vector<long> off((size_t)nslices + 1);
    off[0] = 0;
    for (int s = 0; s < nslices; ++s) off[(size_t)s + 1] = off[(size_t)s] + hw[(size_t)s];
    const long cols = off[(size_t)nslices];                          // block slot columns
    if (ws_dev) { hipFree(ws_dev); hipFree(wl_dev); }
    if ((double)cols * NBR > (level_ptr ? 1.3 : 1.15) * (double)M.nblocks) { hipFree(brow); hipFree(blen); return 0; }
    long *off_dev = nullptr; int *col = nullptr; double *val = nullptr;
    AMG_HIP(hipMalloc((void **)&off_dev, sizeof(long) * ((size_t)nslices + 1)));
    AMG_HIP(hipMalloc((void **)&col, sizeof(int) * (size_t)std::max(cols * NBR, 1L)));
    AMG_HIP(hipMalloc((void **)&val, sizeof(double) * (size_t)std::max(cols * BS * NBR * BS, 1L)));
    AMG_HIP(hipMemcpy(off_dev, off.data(), sizeof(long) * ((size_t)nslices + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((bsell_fill_kernel<BS>), dim3((nslices + 3) / 4), dim3(256), 0, nullptr, nslices, M.Ap, M.Aj, M.Ax, brow, blen, off_dev, col, val);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "bsell fill launch", __FILE__, __LINE__);
    if (rowmap_dev) {                                                // x / b / Dinv are indexed by the ORIGINAL block row
        const int nslots = nslices * NBR;
        hipLaunchKernelGGL(bsell_remap_kernel, dim3((nslots + 255) / 256), dim3(256), 0, nullptr, nslots, rowmap_dev, brow);
    }
    AMG_HIP(hipDeviceSynchronize());
    M.bsl_brow = brow; M.bsl_len = blen; M.bsl_off = off_dev; M.bsl_col = col; M.bsl_val = val; M.bsl_nslices = nslices;
    if (acct) *acct += (long)(6L * nslices * NBR + 8L * (nslices + 1) + 4L * cols * NBR + 8L * cols * BS * NBR * BS);
    return 0;
}

}   // namespace

void free_bsell(DevBsr &M)
{
    if (M.bsl_brow) hipFree(M.bsl_brow);
    if (M.bsl_len) hipFree(M.bsl_len);
    if (M.bsl_off) hipFree(M.bsl_off);
    if (M.bsl_col) hipFree(M.bsl_col);
    if (M.bsl_val) hipFree(M.bsl_val);
    M.bsl_brow = nullptr; M.bsl_len = nullptr; M.bsl_off = nullptr; M.bsl_col = nullptr; M.bsl_val = nullptr; M.bsl_nslices = 0;
}

// from the BSR arrays in HBM; blocks of 2x2 and 3x3, at least 2^15 block rows of 4 to 48 blocks on average
int build_bsell(DevBsr &M, long *acct)
{
    const char *env = std::getenv("AMG_SELL");
    if (env && std::atoi(env) == 0) return 0;
    if (M.nbrows < (1 << 15) || M.nblocks < 4L * M.nbrows || M.nblocks > 48L * M.nbrows || !M.Ap) return 0;
    if (M.bs == 3) return build_bsell_bs<3>(M, acct);
    if (M.bs == 2) return build_bsell_bs<2>(M, acct);
    return 0;
}

// the level-ordered copy of a block Gauss-Seidel schedule: slices cut at the level boundaries; level l is slices
// level_slice[l] .. level_slice[l+1]
int build_bsell_levels(DevBsr &M, const std::vector<int> &level_ptr, const int *rowmap_dev, std::vector<int> &level_slice, long *acct)
{
    const char *env = std::getenv("AMG_SELL");
    level_slice.clear();
    if (env && std::atoi(env) == 0) return 0;
    if (M.nbrows < (1 << 15) || M.nblocks < 4L * M.nbrows || M.nblocks > 48L * M.nbrows || !M.Ap) return 0;
    // only levels of thousands of block rows: a level of 2 500 block rows (C5 at 150^3) is a launch of 30 workgroups that
    // the streamed kernel finishes in 4.8 us -- measured 20 % slower from slices; AMG_SELL_LEVELS=1 forces it (tests)
    const char *force = std::getenv("AMG_SELL_LEVELS");
    if (!(force && std::atoi(force) == 1) && (long)(level_ptr.size() - 1) * 8192L > (long)M.nbrows) return 0;
    int rc = 0;
    if (M.bs == 3) rc = build_bsell_bs<3>(M, acct, &level_ptr, rowmap_dev, &level_slice);
    else if (M.bs == 2) rc = build_bsell_bs<2>(M, acct, &level_ptr, rowmap_dev, &level_slice);
    if (!M.bsl_val) level_slice.clear();
    return rc;
}

int launch_bsell_level(const DevBsr &M, BlockMode m, const BsrStreamArgs &a, int slice_lo, int slice_hi, hipStream_t st)
{
    if (slice_hi <= slice_lo) return 0;
    BsellArgs S{M.bsl_brow, M.bsl_len, M.bsl_off, M.bsl_col, M.bsl_val, M.bsl_nslices};
    const int n = slice_hi - slice_lo;
    const int grid = (n + 3) / 4;
    if (m != BM_BLOCK_GS) { set_error("launch_bsell_level: block Gauss-Seidel only"); return -1; }
    if (M.bs == 3) hipLaunchKernelGGL((bsell_kernel<BM_BLOCK_GS, 3>), dim3(grid), dim3(256), 0, st, a, S, 0, slice_lo, n);
    else if (M.bs == 2) hipLaunchKernelGGL((bsell_kernel<BM_BLOCK_GS, 2>), dim3(grid), dim3(256), 0, st, a, S, 0, slice_lo, n);
    else { set_error("launch_bsell_level: block size not supported"); return -1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "bsell level launch", __FILE__, __LINE__);
    return 0;
}

bool bsell_level_enabled() { return g_sell != 0; }

bool bsell_applies(const DevBsr &M, BlockMode m, const BsrStreamArgs &a)
{
    if (!M.bsl_val || !g_sell || a.rowmap || a.brow_lo != 0 || a.brow_hi != M.nbrows || a.Aj != M.Aj) return false;
    if (m == BM_BLOCK_JACOBI) return true;
    if (m != BM_SPMV) return false;
    return a.smode == SM_MATVEC || a.smode == SM_MATVEC_ACC || a.smode == SM_RESIDUAL || a.smode == SM_POLY_STEP || a.smode == SM_POLY_LAST;
}

int launch_bsell(const DevBsr &M, BlockMode m, const BsrStreamArgs &a, hipStream_t st)
{
    BsellArgs S{M.bsl_brow, M.bsl_len, M.bsl_off, M.bsl_col, M.bsl_val, M.bsl_nslices};
    const int grid = (M.bsl_nslices + 3) / 4;
    const int chunk = grid >= 4096 ? 32 : 0;
    if (M.bs == 3 && m == BM_SPMV) hipLaunchKernelGGL((bsell_kernel<BM_SPMV, 3>), dim3(grid), dim3(256), 0, st, a, S, chunk, 0, M.bsl_nslices);
    else if (M.bs == 3) hipLaunchKernelGGL((bsell_kernel<BM_BLOCK_JACOBI, 3>), dim3(grid), dim3(256), 0, st, a, S, chunk, 0, M.bsl_nslices);
    else if (M.bs == 2 && m == BM_SPMV) hipLaunchKernelGGL((bsell_kernel<BM_SPMV, 2>), dim3(grid), dim3(256), 0, st, a, S, chunk, 0, M.bsl_nslices);
    else if (M.bs == 2) hipLaunchKernelGGL((bsell_kernel<BM_BLOCK_JACOBI, 2>), dim3(grid), dim3(256), 0, st, a, S, chunk, 0, M.bsl_nslices);
    else { set_error("launch_bsell: block size not supported"); return -1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "bsell launch", __FILE__, __LINE__);
    return 0;
}

}   // namespace amg
