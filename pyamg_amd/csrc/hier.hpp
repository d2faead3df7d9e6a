// Device-resident multigrid hierarchy (host-side engine).
#pragma once
#include "amg_dev.hpp"
#include "../../include/amgcore_hip.h"
#include "comm.hpp"

#include <memory>

namespace amg {

// Dependency-level schedule of a sequential sweep (Gauss-Seidel family).
// Tasks = the sequence of rows the reference's loop visits.  Two tasks conflict
// when one writes an unknown the other reads or writes; level(t) = 1 + max level
// of the earlier tasks it conflicts with.  Running levels in ascending order
// (tasks of one level concurrently) reproduces the sequential sweep bit for
// bit; because the conflict relation is symmetric, running the levels in
// DESCENDING order reproduces the reversed sweep.
struct Schedule {
    int ntasks = 0;
    std::vector<int> level_ptr;   // host, size nlevels+1, offsets into the task order
    // CSR flavour: rows copied in level order (coalesced streaming per level)
    DevCsr G;                     // permuted rows, ORIGINAL column indices
    int *rowmap = nullptr;        // device: original row of permuted row
    int *diagpos = nullptr;       // device: position of the diagonal entry in G (-1 if none)
    int *level_ptr_dev = nullptr; // device copy of level_ptr (the chained sweep walks the levels on the GPU)
    // maximal runs [first, last) of consecutive levels that are all narrow enough for ONE workgroup:
    // such a run is swept by a single launch (gs_chain_kernel) instead of one launch per level
    std::vector<std::pair<int, int>> chains;
    std::vector<int> chain_width;      // threads of each chain's workgroup (64 .. 512, by its widest level)
    // The chained sweep's own copy of those runs (gs_chain2_kernel, kernels.hip): per level the rows' off-diagonal
    // entries padded to CHAIN2_PF slots and stored slot-major (coalesced, no row pointer), the diagonal and the row
    // per row, and for every entry an operand CODE per sweep direction: >= 0 the operand's level-order position (it is
    // final in memory long before it is needed and is prefetched), < 0 a slot of the workgroup's LDS ring (the
    // operand was produced by one of the last CHAIN2_D levels of this very launch).
    std::vector<int> gp_host;          // row pointers of G on the host: a level launch gets its workgroups' entry ranges as kernel arguments
    int *cl_code_f = nullptr, *cl_code_b = nullptr;   // long-row chain with LDS hand-off: per entry of G the column, or ~(ring slot); forward / backward sweep
    bool chain_long = false;           // the chains are runs of levels with few but LONG rows: gs_chainl_kernel (entry-parallel, products through LDS)
    bool chain2 = false;
    int c2_pf = 0;                 // slots per row of the copy: 4, 8 or 12
    int *c2_code_f = nullptr, *c2_code_b = nullptr, *c2_off = nullptr;
    double *c2_diag = nullptr, *c2_val = nullptr, *c2_dummy = nullptr;   // diag: by level-order position; dummy: where idle lanes store (512 doubles)
    // LEVEL-ORDER numbering (unknown k = k-th row of the schedule; only when the schedule lists every unknown once):
    // the sweeps of the second-generation chain run on xp = x[rowmap], bp = b[rowmap] with G's columns renumbered
    bool perm = false;
    int *perm_Aj = nullptr;
    double *xp = nullptr, *bp = nullptr, *bd = nullptr;      // bd: (right-hand side, diagonal) pairs for the chained sweep
    // BSR flavour: block rows listed in level order
    int *rows = nullptr;          // device
    DevBsr Gb;                    // BSR flavour with values: block rows copied in level order (streamed)
    std::vector<int> gb_level_slice;   // sliced block form of Gb (sell.hip): first slice of every level (empty: not built)
    // dataflow form (gsflow.hip): the whole sequence of directional sweeps as one persistent launch
    FlowForm flow;
    BlockFlowForm bflow;               // the same for block Gauss-Seidel over Gb
    bool flow_auto = false;            // the default picks it for this schedule (wide or long-row levels that would be launches)
    int nlevels() const { return (int)level_ptr.size() - 1; }
    void release();
    void drop_level_copies();          // the dataflow form serves this schedule: free the level-ordered copies of the other paths
    long level_copy_bytes = 0;         // HBM held by those copies (0 after drop_level_copies)
};

int build_levels(int n, const int *Ap, const int *Aj, const int *tasks, int ntasks,
                 std::vector<int> &level_ptr, std::vector<int> &order);
// allow_flow: the dataflow form may be built (and, where the default picks it, replaces the level-ordered copies);
// false for row-partitioned hierarchies (several ranks may share a device)
// ncols > n: columns n .. ncols-1 exist but no listed row writes them (a partitioned level's halo): frozen operands
int build_csr_schedule(const int *Ap, const int *Aj, const double *Ax, int n, const int *tasks,
                       int ntasks, Schedule &S, hipStream_t st, bool allow_flow = true, int ncols = 0);
// block_flow: the schedule is for BLOCK Gauss-Seidel (relaxation.h:756-810) and may be served by the dataflow form alone
int build_block_schedule(const int *Ap, const int *Aj, int nb, const int *tasks, int ntasks,
                         Schedule &S, hipStream_t st, const double *Ax = nullptr, int bs = 0, bool independent = false,
                         bool block_flow = false);
int sweep_block_schedule(const Schedule &S, BlockMode mode, const double *Dinv, const double *xin, double *x, const double *b,
                         double omega, bool reverse, hipStream_t st);
// block Gauss-Seidel: all directional sweeps of a smoother application (seq[k] != 0: backward); the dataflow form where built
int block_gs_sweeps(const Schedule &S, const double *Dinv, double *x, const double *b, const unsigned char *seq, int nseq, hipStream_t st,
                    bool allow_flow = true);

struct Smoother {
    // AMG_SM_CALLBACK: the relaxation is driven from outside (the device-resident Krylov smoothers of
    // pyamg_amd/krylov.py): cb(user, level, x, b) is called with DEVICE pointers and enqueues on the hierarchy's stream
    amg_relax_callback cb = nullptr;
    void *cb_user = nullptr;
    int kind = AMG_SM_NONE;
    int iterations = 1;
    int sweep = AMG_SWEEP_FORWARD;
    double omega = 1.0;
    std::vector<double> coef;
    int bs = 1;
    double *Dinv = nullptr;          // device
    std::vector<int> indices;        // gauss_seidel_indexed
    DevBsr Ablk;                     // re-blocked A for block smoothers (owned) ...
    bool Ablk_owned = false;
    std::shared_ptr<Schedule> sched; // GS-type smoothers
    // schwarz: subdomains + inverse blocks (device), dependency levels of the subdomain tasks
    int nsd = 0;
    std::vector<int> hSj, hSp;       // host copies for the level analysis at finalize
    int *sw_Sj = nullptr, *sw_Sp = nullptr, *sw_Tp = nullptr, *sw_order = nullptr;
    double *sw_Tx = nullptr, *sw_scratch = nullptr;
    std::vector<int> sw_level_ptr;
    // normal-equation family: aux[0] = A by columns (CSC arrays; gauss_seidel_nr sweeps it, jacobi_ne
    // gathers through it), aux[1] = A with sorted rows when the level's own copy is not (the residual
    // gauss_seidel_nr starts from is a CSC product: terms in ascending column order).  Dinv holds
    // 1/diag(A A^H) or 1/diag(A^H A); the task levels reuse sw_order / sw_level_ptr.
    DevCsr aux[2];
};

// A level of a ROW-PARTITIONED hierarchy (one process per GPU): this rank owns n_own consecutive entries of the
// level's vector space; its vectors are [owned | halo], the halo being the off-rank entries its rows of A_l, R_l
// and P_{l-1} gather, grouped by owner.  Local operators keep their rows' stored order and only renumber columns,
// so every row sum is the single-GPU one.
struct Partition {
    int n_own = 0, n_halo = 0;
    int channel = -1;              // halo exchange plan (comm.hpp); -1: nothing to exchange on this level
    int *send_idx = nullptr;       // device: owned entries the peers need, grouped by destination rank
    int i0 = 0, i1 = 0;            // rows [i0, i1) of A read no halo entry: they run while the halo is in flight
    bool overlap = false;
    // entering the replicated part of the hierarchy: R computes this rank's slice of the coarse right-hand side,
    // all ranks gather the slices
    int gather_channel = -1;
    int gather_rows = 0;
    int *gather_idx = nullptr;     // device: (0..gather_rows) repeated once per rank
    double *rslice = nullptr;
};

struct Level {
    int fmt = AMG_FMT_CSR, R = 1, C = 1;   // container of A in the reference hierarchy
    Partition part;
    DevCsr A, P, Rm;
    DevBsr Ab;                             // A's own blocks when fmt == BSR and R == C > 1
    bool hasA = false, hasP = false, hasR = false;
    Smoother sm[2];
    // work vectors (device), length n
    double *x = nullptr, *xalt = nullptr, *b = nullptr, *r = nullptr, *h = nullptr, *h2 = nullptr;
    double *amli[4] = {nullptr, nullptr, nullptr, nullptr};   // p0, p1, A*p, A*p_j (AMLI cycles, lazily)
    std::shared_ptr<Schedule> sched_csr, sched_blk;   // natural-order schedules, shared pre/post
    std::shared_ptr<Schedule> sched_bgs;              // block Gauss-Seidel over the level's own blocks, shared pre/post
};

}  // namespace amg

namespace amg {
// schwarz.hip
int schwarz_levels(int nrows, const int *Ap, const int *Aj, const int *Sj, const int *Sp,
                   const std::vector<int> &tasks, std::vector<int> &level_ptr, std::vector<int> &order);
int launch_schwarz_level(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b,
                         const double *Tx, const int *Tp, const int *Sj, const int *Sp, double *scratch,
                         const int *doms, int count, hipStream_t st);
// ne.hip
int ne_touch_levels(int nvec, const int *Ap, const int *Aj, int ntasks, std::vector<int> &level_ptr,
                    std::vector<int> &order);
int launch_gs_ne_level(const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, const double *Dinv,
                       double omega, const int *rows, int count, hipStream_t st);
int launch_gs_nr_level(const int *Ap, const int *Aj, const double *Ax, double *x, double *r, const double *Dinv,
                       double omega, const int *cols, int count, hipStream_t st);
int launch_jacobi_ne_gather(const int *Tp, const int *Trow, const double *Tval, const double *delta, double omega,
                            double *temp, int n, hipStream_t st);
}  // namespace amg

namespace amg {
// One captured iteration (cycle + residual norm) for a given buffer state: Jacobi-type smoothers
// swap x/xalt as they go, so the launch sequence depends on which buffer each level's x lives in.
struct GraphEntry {
    std::vector<double *> state_in, state_out;   // (x, xalt) of every level before / after
    int cyc = 0;
    bool x_zero = false;
    bool kept_in = false, kept_out = false;     // amg_hier::r_kept before / after
    int seen = 0;                                // times this state was run eagerly
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
};
}  // namespace amg

struct amg_hier {
    int device = 0;
    int nlevels = 0;
    std::vector<amg::Level> lv;
    hipStream_t stream = nullptr;
    bool finalized = false;
    // coarse solve
    int coarse_kind = 0;              // 0 none (zero correction), 1 dense, 2 smoother, 3 host callback
    amg_coarse_callback coarse_cb = nullptr;   // x = solve(A_coarse, b) on HOST vectors (Krylov names, callables: <= a few hundred rows)
    void *coarse_cb_user = nullptr;
    std::vector<double> coarse_hb, coarse_hx;
    double *coarse_Mt = nullptr;
    int coarse_n = 0;
    amg::Smoother coarse_sm;
    // scratch
    double *norm_scratch = nullptr;   // 1024 partials
    double *pcg[4] = {nullptr, nullptr, nullptr, nullptr};   // x, r, p, A*p of the device PCG (lazily)
    double *sumsq_partials = nullptr; // one partial per workgroup of the level-0 residual kernel
    long sumsq_cap = 0;
    double *res_dev = nullptr;        // residual history on device
    int res_cap = 0;
    // Arnoldi workspace for setup-time spectral-radius estimates
    double *arn_V = nullptr;          // (arn_m + 1) vectors of length arn_n
    double *arn_dinv = nullptr;
    double *arn_coef = nullptr;
    int arn_m = 0;
    long arn_n = 0;
    long arn_bytes = 0;               // HBM held by the workspace (part of dev_bytes while it lives)
    // hipGraph replay of the iteration (launch-bound hierarchies: small levels, level-scheduled GS)
    std::vector<amg::GraphEntry> graphs;
    int use_graphs = 1;
    bool has_callbacks = false;                  // callback smoothers / coarse solver: iterations run eagerly
    int graph_epoch = 0;                         // amg::config_epoch() the cached graphs were captured under
    int keep_residual = 1;                       // hand the outer residual to the next pre-smoother
    bool r_kept = false;                         // lv[0].r == lv[0].b - A*lv[0].x right now (solve loop only)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    long dev_bytes = 0;
    // row-partitioned mode (amg_hier_set_comm): not owned
    amg_comm *comm = nullptr;
    int reduce_channel = -1;                     // 1 x 1 channel for the residual-norm all-reduce
    // a PARTITIONED coarsest level with a dense coarse solver: all ranks gather the right-hand side, apply the
    // dense operator redundantly (bit-identical everywhere) and keep their slice of the result
    int coarse_gather_channel = -1, coarse_lo = 0;
    int *coarse_gather_idx = nullptr;
    double *coarse_full_b = nullptr, *coarse_full_x = nullptr;
    int overlap = 1;                             // interior rows while the halo is in flight
};
