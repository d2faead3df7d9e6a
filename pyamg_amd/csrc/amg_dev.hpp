// Internal declarations shared by the translation units of libamgcore_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace amg {

// ---------------------------------------------------------------- errors
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define AMG_HIP(call)                                                         \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return amg::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------- device CSR
// Arrays are over-allocated by PAD entries so that 16-byte vector loads that
// start up to 3 entries before / end up to 3 entries after a row block stay in
// bounds.
constexpr int PAD = 16;

struct DevBsr;
struct DevCsr {
    int nrows = 0, ncols = 0;
    long nnz = 0;
    int *Ap = nullptr;
    int *Aj = nullptr;
    double *Ax = nullptr;
    bool owned = true;
    int longest_row = -1;          // entries of the longest row, found on first need (amg_hier_gs_natural)
    // The operator's own square blocks (a BSR level of the hierarchy), not owned: when set, applications
    // stream 8 B per entry + 4 B per block from there instead of 12 B per entry from the expanded CSR
    const DevBsr *blk = nullptr;
    // Offset-pattern form of the column indices (square operators on structured grids): row i's
    // columns are i + dict_off[dict_ptr[pat[i]] .. ], with a dictionary of a few distinct offset
    // tuples.  When present, the pattern kernel streams 8 B per entry + 4 B per row instead of
    // 12 B per entry; results are identical (same entries, same order).
    int *pat = nullptr;        // [nrows] pattern id of every row
    int *dict_ptr = nullptr;   // [npat+1]
    int *dict_off = nullptr;   // [ndict]
    int npat = 0, ndict = 0;
    int period_rows = 0;       // largest |column - row| of the dictionary: the stride of the slowest grid axis
    // Stencil form (a second copy of the values, built on the device when every row's offsets are
    // an increasing subset of one union stencil U of at most STENCIL_MAX offsets): slot u of row i
    // holds a(i, i+U[u]) if bit u of mask[i] is set.  Values are laid out [block][slot][256 rows],
    // so every load of the kernel is issued up front, coalesced, with no row pointer, no column
    // index and no LDS staging; x[i+U[u]] is gathered speculatively and masked.
    // Sliced form of an irregular operator (SELL-64-sigma; built on the device by build_sell): the rows of every window
    // of 2048 are sorted by length (longest first, ties in row order), cut into slices of 64, and a slice's entries
    // are stored entry-major -- entry k of its 64 rows side by side -- padded to the slice's longest row.  One wave per
    // slice, one lane per row: no row pointer, no LDS, no barrier; every lane sums ITS row's entries in stored
    // order, so results are bit-identical to the CSR kernel's.
    int *sl_row = nullptr;         // [sl_nslices * 64] row of the slot (-1: padding slot)
    unsigned short *sl_len = nullptr;  // [sl_nslices * 64] entries of that row
    long *sl_off = nullptr;        // [sl_nslices + 1] first entry of the slice in sl_col / sl_val (multiples of 64)
    int *sl_col = nullptr;
    double *sl_val = nullptr;
    int sl_nslices = 0;
    // 16-bit column codes of the sliced form (lossless): the columns a slice's 64 rows reference fall into a few narrow
    // clusters; when at most 16 aligned windows of 4096 columns cover them, entry k of a row is stored as
    // (window slot << 12) | (column & 4095), two codes per 32-bit word, and the slice's 16 window origins travel in one
    // 64-byte load: 10 B per entry instead of 12.  Slices that need more windows keep reading sl_col (flag per slice).
    unsigned *sl_code = nullptr;       // [sl_coff[nslices]] pairs of codes, pair-major per slice: [k / 2][64 lanes]
    int *sl_org = nullptr;             // [nslices * 16] window origins (multiples of 4096)
    long *sl_coff = nullptr;           // [nslices + 1] first code word of the slice
    unsigned char *sl_flag16 = nullptr;// [nslices] 1 = this slice's entries are coded
    double sl_frac16 = 0.0;            // fraction of the slices that are coded
    long sl_entries = 0;           // padded entries stored
    int sl_lo = 0, sl_hi = 0;      // rows the sliced form covers (all of them; a partitioned level: its interior rows)
    double *st_vals = nullptr;     // [nblocks256 * st_nu * 256]
    void *st_mask = nullptr;       // [nrows] uint8 when |U| <= 7, else uint32; top bit = row not covered
    int st_nu = 0;                 // |U|
    // Rows that use a rare offset (halo columns of a rank-local operator, a few irregular rows) are
    // left out of the stencil form instead of widening U for everybody: up to STENCIL_RANGES
    // contiguous row ranges that the offset-pattern kernel applies after the stencil launch.
    int st_nranges = 0;
    int st_range[8][2] = {{0}};
    // Value index on top of the stencil form (automatic since r3, amg_hier_value_index / amg_set_value_index): when the operator holds at
    // most 255 distinct values (constant-coefficient stencils), slot u of row i stores a one-byte code
    // into a dictionary kept in LDS instead of the 8-byte value -- the product uses the very same double.
    unsigned char *st_codes = nullptr;   // 8 slots per 64-bit word: [nblocks256][ceil(st_nu / 8)][256] words
    double *st_dict = nullptr;           // [256]
    int st_ndict = 0;
    bool st_vi_on = false;
    // 16-bit column codes for csr_stream_kernel (irregular operators: coarse A_l, P_l, R_l).  The
    // columns one workgroup's rows reference fall into a few narrow clusters; when at most 16
    // aligned windows of 4096 columns cover them, entry k is stored as (window slot << 12) | (column
    // & 4095) and the 16 window origins sit in LDS: 2 B per entry instead of 4.  Workgroups that
    // need more windows keep reading Aj (flag per workgroup).  Same columns, same order, same bits.
    unsigned short *Aj16 = nullptr;    // [nnz]
    int *wg_base = nullptr;            // [nwg * 16]
    unsigned char *wg_flag = nullptr;  // [nwg] 1 = this workgroup's entries are coded
    int i16_rpb = 0;                   // rows per workgroup the coding was built for
    double i16_frac = 0.0;             // fraction of the entries that are coded
    int st_u0 = -1;                // slot of offset 0 (the diagonal), -1 if absent
    int st_off[32] = {0};          // U, increasing
};
constexpr int STENCIL_MAX = 31;
constexpr int STENCIL_RANGES = 8;

constexpr int PAT_MAX = 255;       // distinct row patterns kept in LDS
constexpr int PAT_DICT_MAX = 2048; // total offsets in the dictionary

struct DevBsr {   // block rows, for the block / point-BSR relaxation kernels
    int nbrows = 0, bs = 1;
    long nblocks = 0;
    int *Ap = nullptr;
    int *Aj = nullptr;
    double *Ax = nullptr;
    // Sliced form of the block operator (sell.hip build_bsell; bs = 2, 3): block rows sorted by block count inside
    // windows, slices of 64 / bs block rows, one lane per SCALAR row; block k of the slice's rows side by side
    // (columns [k][block row], values [k][c][lane]).  Whole passes (operator applications, block Jacobi sweeps) run
    // from it without row pointer, LDS or barrier; same summation order, same bits as bsr_stream_kernel.
    int *bsl_brow = nullptr;           // [nslices * (64 / bs)] block row of the slot (-1: padding)
    unsigned short *bsl_len = nullptr; // blocks of that block row
    long *bsl_off = nullptr;           // [nslices + 1] first block SLOT COLUMN of the slice: slice s holds (off[s+1]-off[s]) blocks per row
    int *bsl_col = nullptr;            // [off[nslices] * (64 / bs)]
    double *bsl_val = nullptr;         // [off[nslices] * bs * (64 / bs) * bs]
    int bsl_nslices = 0;
};

// ---------------------------------------------------------------- stream kernel
// Row epilogues of the CSR-stream kernel.  `s` is the strict left-to-right sum
// over the stored row (scipy csr_matvec order); for the relaxation modes the
// diagonal entry is excluded from the sum (pyamg/amg_core/relaxation.h).
enum StreamMode {
    SM_MATVEC = 0,      // out[i]  = s                                  (scipy A*x, y pre-zeroed)
    SM_MATVEC_ACC,      // out[i]  = out[i] + s                         (scipy csr_matvec accumulate; x += P*cx)
    SM_RESIDUAL,        // out[i]  = b[i] - s                           (multilevel.py:496)
    SM_POLY_FIRST,      // out[i]  = b[i] - s ; out2[i] = c0*out[i]     (relaxation.py:661-663)
    SM_POLY_STEP,       // out[i]  = c0*b[i] + s                        (relaxation.py:666, b = residual)
    SM_POLY_LAST,       // out[i]  = v2[i] + (c0*b[i] + s)              (relaxation.py:666,668 fused, v2 = x)
    SM_JACOBI,          // out[i]  = (1-w)*v2[i] + w*((b[i]-s)/d)       (relaxation.h:202-239; xg = v2 = temp)
    SM_JACOBI_BSR1,     // rs=b[i]; rs-=p..; out[i] = (1-w)*v2[i] + w*rs/d   (relaxation.h:268-360, bs=1)
    SM_GS,              // out[row] = (b[row]-s)/d                      (relaxation.h:34-62, one level)
    SM_GS_BSR1,         // rs=b[row]; rs-=p..; out[row] = rs/d          (relaxation.h:90-173, bs=1)
    SM_RESIDUAL_SUMSQ,  // out2[block] = sum over the block's rows of (b[i]-s)^2; r stored only if out != nullptr
                        // (the outer residual norm, multilevel.py:461, without the 16n vector round trip)
};

struct StreamArgs {
    const int *Ap;
    const int *Aj;
    const double *Ax;
    int row_lo, row_hi;      // rows [row_lo,row_hi) of the (possibly permuted) matrix
    const double *xg;        // gathered vector
    const double *b;         // streamed rhs / residual
    const double *v2;        // second streamed vector
    double *out;
    double *out2;
    double c0;               // coefficient / omega
    const int *rowmap;       // GS levels: original row of permuted row i (rhs/out index), else null
    const int *diagpos;      // GS levels: position of the diagonal entry of permuted row i (-1: none)
    long nnz_total;          // entries in Aj/Ax (bound for 16-byte loads)
    int rows_per_wg;         // rows handled by one workgroup (1..256; 0 -> 256)
    const unsigned short *Aj16;   // 16-bit column codes (DevCsr::Aj16) or null
    const int *wg_base;
    const unsigned char *wg_flag;
    double gscale;           // operand scaling: products are a_ij * (gscale * xg[j]); 0 is read as 1.
                             // (polynomial smoother: gather c0*r straight from r, relaxation.py:663-666)
};

// variant: 0 = scalar (8 B / 4 B per lane) loads, 1 = 16-byte vector loads
int launch_stream(StreamMode mode, const StreamArgs &a, hipStream_t st);
// same modes (except the GS ones) through the offset-pattern form of M's column indices
int launch_pattern(StreamMode mode, const StreamArgs &a, const DevCsr &M, hipStream_t st);
bool pattern_supports(StreamMode mode);
int launch_stencil(StreamMode mode, const StreamArgs &a, const DevCsr &M, hipStream_t st);
// the sliced form (DevCsr::sl_*): whole-operator applications in the modes sell_supports() names
bool sell_supports(StreamMode mode);
bool sell_enabled();
void set_sell_form(int on);
void set_sell_index16(int on);
bool sell_index16_enabled();              // 1 (default): sliced forms built from now on also get 16-bit column codes and the kernel reads those
int launch_sell(StreamMode mode, const StreamArgs &a, const DevCsr &M, hipStream_t st);
int build_sell(DevCsr &M, long *acct, int row_lo = 0, int row_hi = -1);   // from the CSR arrays already in HBM (rows [lo, hi), default all); leaves M untouched if not worth it
void free_sell(DevCsr &M);
int stencil_blocks(const StreamArgs &a, const DevCsr &M);   // workgroups (= SM_RESIDUAL_SUMSQ partials) of launch_stencil
int launch_stencil_build(const DevCsr &M, const int *dict_slot, const unsigned *pat_mask, hipStream_t st);
// value index: distinct values of st_vals into a 1024-slot table (EMPTY = all ones), then the byte codes
int launch_value_scan(const double *vals, long count, unsigned long long *table, int *overflow, hipStream_t st);
int launch_value_encode(const double *vals, long count, const double *dict_sorted, int ndict, unsigned char *codes, int nu,
                        hipStream_t st, const unsigned char *mask8 = nullptr, long nrows = 0);
int config_epoch();                      // bumped by every set_* knob below
void bump_config_epoch();
// Gauss-Seidel: runs of dependency levels of at most gs_chain_max_rows() rows swept by ONE workgroup in
// one launch (barrier between levels, next level's rows prefetched) instead of a launch per level
int launch_gs_chain(const DevCsr &G, const int *rowmap, const int *diagpos, const int *level_ptr_dev, int l_first,
                    int nlevels, int width, bool reverse, bool bsr1, double *x, const double *b, hipStream_t st);
// second generation of the chained sweep: operands produced by the last CHAIN2_D levels travel through LDS, everything
// else (entries, diagonal, right-hand side, older operands) is prefetched CHAIN2_D levels ahead from the padded copy
constexpr int CHAIN2_WG = 512;      // rows per level at most
constexpr int CHAIN2_PF = 12;       // off-diagonal entries per row at most (copies are padded to 4, 8 or 12 slots)
constexpr int CHAIN2_D = 2;         // prefetch distance in levels = levels whose results are passed through LDS
constexpr int CHAIN2_LMAX = 4096;   // levels per launch
constexpr int CHAIN2_EMPTY = -2147483647 - 1;
int launch_perm_gather(const int *rowmap, const double *x, const double *b, const double *diag, double *xp, double *bp, double *bd,
                       int n, hipStream_t st);
int launch_perm_scatter(const int *rowmap, const double *xp, double *x, int n, hipStream_t st);
int launch_gs_chain2(const int *lp, const double *val, const int *code, const int *off, double *dummy, int pf,
                     int l_first, int nlevels, int width, bool reverse, bool bsr1, double *x, const double *bd, int nzero, hipStream_t st);
int gs_chain_max_rows();
// runs of levels with few but long rows (SA coarse levels): entry-parallel products through LDS, one workgroup
// of `width` (128 / 256 / 512) threads; a level holds at most `width` rows and width * gs_chainl_entries_per_lane() entries
int launch_gs_chain_long(const DevCsr &G, const int *rowmap, const int *diagpos, const int *level_ptr_dev, int l_first,
                         int nlevels, int width, bool reverse, bool bsr1, double *x, const double *b, hipStream_t st,
                         const int *ring_code = nullptr);      // ring_code: per-entry codes of the LDS hand-off variant (Schedule::cl_code_*)
int gs_chainl_max_levels();
// one dependency level by a launch of its own: with the level-ordered copy's row pointers on the host the entry
// ranges travel in the kernel arguments (two memory round trips instead of three); falls back to launch_stream
int launch_gs_level(const StreamArgs &a, bool bsr1, const int *gp_host, hipStream_t st);
void set_gs_level_hint(int on);
int gs_chainl_max_rows();
int gs_chainl_entries_per_lane();
bool gs_chain_enabled();
int gs_chain_generation();
void set_gs_chain(int on);
bool stencil_enabled();
int launch_index16_build(DevCsr &M, int rpb, hipStream_t st);   // fills Aj16 / wg_base / wg_flag (already allocated)
bool index16_enabled();
void set_index16(int on);
void set_stencil_pairs(int on);          // 1: two rows per lane in the stencil form (16-byte accesses), 0: one
void set_stencil_form(int on);           // 0: dispatch pattern operators to csr_pattern_kernel instead
void set_xcd_period(int on);             // plane-periodic block->XCD mapping of pattern operators (default on)
int stream_blocks(const StreamArgs &a);   // workgroups launch_stream will use (partials of SM_RESIDUAL_SUMSQ)
int launch_sum_sqrt(const double *partial, long np, double *scratch512, double *result_dev, hipStream_t st);   // sqrt(sum), fixed order
int launch_axpy_scaled(double *x, const double *r, double c, long n, hipStream_t st);        // x += c*r
void set_stream_variant(int v);
void set_stream_pipe(int on);              // 1 (default): persistent, software-pipelined CSR stream kernel
bool stream_pipe_enabled();
int stream_variant();
void set_xcd_chunk(int c);
void set_tile_target(int t);
int rows_per_wg_for(long nnz, long rows);

// thread-per-row fallbacks for non-unit strides / index lists (exact same arithmetic)
int launch_jacobi_rows(const DevCsr &A, const double *temp, const double *b, double *x,
                       int row_start, int count, int row_step, double omega, hipStream_t st);

// ---------------------------------------------------------------- vector kernels
int launch_scale(double *out, const double *in, double c, long n, hipStream_t st);          // out = c*in
int launch_fill(double *out, double v, long n, hipStream_t st);                               // out = v
int launch_scale_add(double *p, double beta, const double *z, long n, hipStream_t st);          // p = beta*p + z
int launch_sor_combine(double *x, const double *xold, double omega, long n, hipStream_t st); // x = w*x + (1-w)*xold
int launch_axpy_inplace(double *x, const double *h, long n, hipStream_t st);                 // x += h
int launch_sub(double *out, const double *a, const double *b, long n, hipStream_t st);       // out = a - b
int launch_copy_strided(double *dst, const double *src, int start, int count, int step, hipStream_t st);
// ||x||_2 into *result_dev (deterministic two-stage reduction); scratch >= 1024 doubles
int launch_norm2(const double *x, long n, double *scratch, double *result_dev, hipStream_t st);
int launch_dot(const double *x, const double *y, long n, double *scratch, double *result_dev, hipStream_t st);
int launch_axmy(double *w, const double *v, double a, long n, hipStream_t st);      // w -= a*v
int launch_axmy_ratio(double *w, const double *v, const double *num, const double *den, double sign, long n, hipStream_t st);   // w -= (sign * *num / *den) * v, scalars in device memory
int launch_divide(double *w, double a, long n, hipStream_t st);                     // w /= a
int launch_mul_elem(double *w, const double *d, long n, hipStream_t st);            // w *= d (elementwise)
int launch_combine(double *out, const double *V, const double *coef_dev, int m, long n, long ld, hipStream_t st);
// x = M b with M given transposed (Mt[k*n+i] = M[i][k]); sequential sum per row
int launch_dense_apply(const double *Mt, const double *b, double *x, int n, hipStream_t st);

// ---------------------------------------------------------------- BSR kernels
enum BlockMode { BM_BSR_JACOBI, BM_BLOCK_JACOBI, BM_BSR_GS, BM_BLOCK_GS,
                 BM_SPMV };   // operator application straight from the blocks (scipy bsr_matvec order); epilogue = smode

// streamed variant over a contiguous range of block rows of a (possibly level-permuted) BSR operator
struct BsrStreamArgs {
    const int *Ap; const int *Aj; const double *Ax;   // BSR arrays of the operator being streamed
    int bs;
    int brow_lo, brow_hi;   // block rows [lo, hi) of that operator
    const int *rowmap;      // permuted operator: original block row of each row (x / b / Dinv index); else null
    int intra_reverse;
    const double *xin;      // operand vector (temp for Jacobi, the live x for GS)
    double *xout;
    const double *b;
    const double *Dinv;
    double omega;
    // BM_SPMV: epilogue (SM_MATVEC, SM_MATVEC_ACC, SM_RESIDUAL, SM_POLY_STEP, SM_POLY_LAST), its second
    // streamed vector / coefficient, and the on-the-fly operand scaling of the polynomial smoother
    int smode;
    const double *v2;
    double c0, gscale;
    double *out2;           // SM_RESIDUAL_SUMSQ: one partial sum of squares per workgroup
};
int bsr_stream_blocks(const BsrStreamArgs &a, long nblocks_hint);   // workgroups launch_bsr_stream will use
bool bsr_spmv_supports(StreamMode mode);
bool bsr_spmv_enabled(int bs);
void set_bsr_spmv(int on);
int launch_bsr_stream(BlockMode m, const BsrStreamArgs &a, long nblocks_hint, hipStream_t st);
// the sliced block form (DevBsr::bsl_*): whole passes in BM_SPMV (epilogues MATVEC, MATVEC_ACC, RESIDUAL, POLY_STEP,
// POLY_LAST) and BM_BLOCK_JACOBI
int build_bsell(DevBsr &M, long *acct);
void free_bsell(DevBsr &M);
bool bsell_applies(const DevBsr &M, BlockMode m, const BsrStreamArgs &a);
int launch_bsell(const DevBsr &M, BlockMode m, const BsrStreamArgs &a, hipStream_t st);
// level-ordered copy of a block Gauss-Seidel schedule: slices cut at level boundaries, one launch per level
int build_bsell_levels(DevBsr &M, const std::vector<int> &level_ptr, const int *rowmap_dev, std::vector<int> &level_slice, long *acct);
int launch_bsell_level(const DevBsr &M, BlockMode m, const BsrStreamArgs &a, int slice_lo, int slice_hi, hipStream_t st);
bool bsell_level_enabled();

// ---------------------------------------------------------------- dataflow Gauss-Seidel (gsflow.hip)
// A whole sequence of directional sweeps as ONE persistent launch: the unknowns in level-order numbering, every
// dependency level cut into chunks of 64 rows (one wave, one lane per row, the off-diagonal entries slot-major), chunks
// dealt to the resident waves in sweep order.  A sweep reads the previous sweep's values from one buffer and publishes
// its own into another that starts out filled with a signalling-NaN sentinel; an operand is ready when it no longer
// reads as the sentinel, so a row starts the moment ITS operands exist -- no barrier, no launch per level.
constexpr int FLOW_MAXSEQ = 4;          // directional sweeps per launch (buffers: FLOW_MAXSEQ + 1)
struct FlowChunk { int row0, nrows, nslots, lvl_lo, lvl_hi, off, pad0, pad1; };   // nslots: slots per LANE; off: first slot row of the chunk (x 64 entries)
struct FlowForm {
    bool ready = false;
    int n = 0, ncols = 0, nchunks = 0, nlevels = 0, lpr = 1;    // ncols >= n: columns n .. ncols-1 are frozen operands (a partitioned level's halo); lpr: lanes sharing a row (1 .. 64, by the longest row: 8 slots per lane)
    long slot_rows = 0;                 // rows of 64 (value, column) pairs stored
    int *rowmap = nullptr;              // [n] original row of level-order position k (rows of a level sorted by length)
    FlowChunk *meta = nullptr;          // [nchunks]
    int *col = nullptr;                 // [slot_rows * 64] operand position (n: the permanent 0.0 of padded slots)
    double *val = nullptr;              // [slot_rows * 64]
    double *diag = nullptr;             // [n]
    int *gate_f = nullptr, *gate_b = nullptr;   // [n] forward / backward sweep: the row's latest operand produced at least two levels earlier (n: none)
    double *bp = nullptr;               // [n] right-hand side in level-order numbering (gathered per application)
    double *X = nullptr;                // [(FLOW_MAXSEQ + 1) * xstride] iterate buffers, entry n of each = 0.0
    long xstride = 0;
    long bytes = 0;
    void release();
};
// from the level-ordered copy of a schedule (host arrays): leaves F.ready false when the form does not apply
int build_flow_form(FlowForm &F, int n, int ntasks, const std::vector<int> &level_ptr, const std::vector<int> &rowmap,
                    const std::vector<int> &gp, const std::vector<int> &gj, const std::vector<double> &gx, int ncols = 0);
int gs_flow_sweep(const FlowForm &F, bool bsr1, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st);
// block Gauss-Seidel (relaxation.h:756-810) the same way: a lane per SCALAR row, LPR lanes per scalar row, FLOW_SEG
// blocks per lane; positions count block rows, the iterate buffers hold bs scalars per block row
struct BlockFlowForm {
    bool ready = false;
    int nb = 0, bs = 0, nchunks = 0, nlevels = 0, lpr = 1, seg = 8;     // seg blocks per lane, lpr lanes per scalar row
    long slot_rows = 0;
    int *rows = nullptr;                // [nb] original block row of position k
    FlowChunk *meta = nullptr;
    int *col = nullptr;                 // [slot_rows * (64 / bs)] block position of the slot's operand (nb: the permanent zero block)
    double *val = nullptr;              // [slot_rows * bs * 64]: slot, column inside the block, lane
    int *gate_f = nullptr, *gate_b = nullptr;   // [nb]
    double *bp = nullptr;               // [nb * bs]
    double *X = nullptr;                // [(FLOW_MAXSEQ + 1) * xstride]
    long xstride = 0;
    long bytes = 0;
    void release();
};
int build_block_flow_form(BlockFlowForm &F, int nb, int bs, int ntasks, const std::vector<int> &level_ptr, const std::vector<int> &rows,
                          const std::vector<int> &gp, const std::vector<int> &gj, const std::vector<double> &gx);
int block_flow_sweep(const BlockFlowForm &F, const double *Dinv, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st);
// Gauss-Seidel sweeps in the operator's own row order from its CSR arrays (gsflow.hip: gs_natural_kernel); -40: a row is too long
int gs_natural_sweeps(const int *Ap, const int *Aj, const double *Ax, int n, int longest_row, double *x, const double *b,
                      const unsigned char *dirs, int nsweeps, hipStream_t st, const int *order_fwd = nullptr, const int *order_bwd = nullptr,
                      const int *tstart_fwd = nullptr, const int *tstart_bwd = nullptr, int ntasks_fwd = 0, int ntasks_bwd = 0);
int gs_flow_mode();                     // 0 off, 1 where it measured faster (default), 2 wherever the form exists
void set_gs_flow(int mode);
void set_gs_flow_lookahead(int levels);
int gs_flow_status();                   // host-synchronous: nonzero when a wave of a dataflow sweep ran out of its time budget (resets the flag)

}  // namespace amg
