"""Row-partitioned multi-GPU solve: one process per GPU, every level of the
hierarchy split into contiguous row blocks, neighbour halos moved with
torch.distributed (backend "nccl" = RCCL over xGMI), one scalar all-reduce per
iteration for the residual norm.

The reference has no distributed path; the arithmetic contract is that of the
single-GPU cycle (pyamg/multilevel.py:316-548): every row is summed exactly as
before, its off-rank operands arriving through a halo appended to the local
vector, so Jacobi / polynomial (Chebyshev) cycles produce iterates that are
BIT-IDENTICAL to the one-GPU run whatever the number of ranks; only the
residual norm differs in the last bits (per-rank partial sums are all-reduced).
Gauss-Seidel sweeps are inherently sequential across ranks: they are offered as
HYBRID sweeps -- Gauss-Seidel inside a rank (lexicographic `gauss_seidel`, or an
index list such as the multicolour ordering: `gauss_seidel_indexed`), Jacobi
across ranks: the halo is refreshed once per directional sweep and frozen
during it (BASELINE configuration C4).  Their iterates differ from the one-GPU
sweep by construction; the oracle is the partition-emulating CPU run
(tests/test_distributed_cpu.py::_hybrid_sweep).

Layout per rank and level l (vector space V_l):  [ owned entries | halo ]
 * owned = the contiguous index range bounds[l][rank] .. bounds[l][rank+1]
 * halo  = the sorted off-rank indices any local row of A_l, R_l or P_{l-1}
           gathers from V_l (grouped by owner because owners are contiguous),
           so ONE exchange plan per level serves all three operators.
Local operators keep their rows' entry order and only renumber columns
(owned -> 0..n_own-1, halo -> n_own + position), so summation order is kept.

The compute backend is pluggable: `HipBackend` (the product: csr_stream kernels
through include/amgcore_hip.h section 3 on torch CUDA tensors).  tests/ plug a
CPU backend built on the oracle to exercise partitioning, plans and exchange
sequencing with gloo where no GPU exists.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sparse

MATVEC, MATVEC_ACC, RESIDUAL, POLY_FIRST, POLY_STEP, POLY_LAST, JACOBI, JACOBI_BSR1 = range(8)


def split_rows(n, world):
    return np.array([(n * p) // world for p in range(world + 1)], dtype=np.int64)


def coarse_bounds(P, fine_bounds):
    """Row partition of the next-coarser level that FOLLOWS the fine one through the prolongator: rank k's coarse
    rows start at the coarse unknown its first fine row interpolates from most strongly (the aggregate that row sits
    in for smoothed aggregation, the nearest C-point for classical interpolation).  Aggregates / C-points are numbered
    in the order of the fine rows, so the fine rows [f_k, f_k+1) then interpolate almost only from [c_k, c_k+1) and
    the halos of P, R and the coarse operator stay surface-sized, where an even split of the coarse rows drifts away
    from the fine one by whole planes of the grid."""
    Ap, Aj, Ax, _ = _csr_view(P)
    n, nc = P.shape
    world = len(fine_bounds) - 1
    out = np.zeros(world + 1, dtype=np.int64)
    out[world] = nc
    for k in range(1, world):
        f = int(fine_bounds[k])
        c = nc
        while f < n:                                   # first row at or after the cut that has entries
            s, e = int(Ap[f]), int(Ap[f + 1])
            if e > s:
                c = int(Aj[s + int(np.argmax(np.abs(np.asarray(Ax[s:e]))))])
                break
            f += 1
        out[k] = min(max(c, int(out[k - 1])), nc)
    # where the coarse numbering does not follow the fine one (deep levels; aggregates formed in the clean-up pass
    # are numbered last) the cut would starve some ranks: keep the even split there
    even = split_rows(nc, world)
    if np.any(np.abs(np.diff(out) - np.diff(even)) > 0.2 * np.maximum(np.diff(even), 1)):
        return even
    return out


class _Lazy(object):
    """memory-mapped global CSR arrays with the small interface local_rows needs"""

    def __init__(self, Ap, Aj, Ax, shape, bsr):
        self.indptr, self.indices, self.data, self.shape, self._bsr = Ap, Aj, Ax, shape, bsr
        self.nnz = int(Ap[-1])


def _csr_view(M):
    """(indptr, indices, data, is_bsr11) of a CSR or BSR(1,1) matrix"""
    if isinstance(M, _Lazy):
        return M.indptr, M.indices, M.data, M._bsr
    if sparse.isspmatrix_bsr(M):
        if M.blocksize != (1, 1):
            raise NotImplementedError("the partitioned path supports scalar (CSR / BSR(1,1)) operators")
        return M.indptr, M.indices, M.data.reshape(-1), True
    if not sparse.isspmatrix_csr(M):
        M = sparse.csr_matrix(M)
    return M.indptr, M.indices, M.data, False


def local_rows(M, lo, hi):
    """rows [lo, hi) of a global operator: (Ap rebased, Aj global, Ax)"""
    Ap, Aj, Ax, _ = _csr_view(M)
    s, e = int(Ap[lo]), int(Ap[hi])
    return (np.asarray(Ap[lo:hi + 1], dtype=np.int64) - s, np.asarray(Aj[s:e], dtype=np.int64),
            np.asarray(Ax[s:e], dtype=np.float64))


def local_rows_of(M, rows, col_map=None):
    """the listed rows of a global operator, in list order, entries of a row in STORED order (the summation order):
    (Ap rebased, Aj -- relabelled through col_map when given --, Ax)"""
    Ap, Aj, Ax, _ = _csr_view(M)
    rows = np.asarray(rows, dtype=np.int64)
    Ap = np.asarray(Ap, dtype=np.int64)
    start = Ap[rows]
    lens = Ap[rows + 1] - start
    out_p = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(lens, out=out_p[1:])
    # position k of the output reads entry start[row of k] + (k - out_p[row of k])
    src = np.repeat(start - out_p[:-1], lens) + np.arange(int(out_p[-1]), dtype=np.int64)
    cols = np.asarray(Aj)[src].astype(np.int64)
    if col_map is not None:
        cols = col_map[cols]
    return out_p, cols, np.asarray(Ax)[src].astype(np.float64)


def owners_by_aggregate(levels, owner0, world):
    """Ownership of every level by INDEX SET for an arbitrary numbering: level 0 as given (owner0[i] = rank of unknown i),
    a coarse unknown goes to the rank that owns the fine row interpolating from it most strongly (its aggregate's
    root for smoothed aggregation, the C-point itself for classical interpolation) -- aggregates are never split from
    their strongest member, whatever the numbering."""
    owners = [np.asarray(owner0, dtype=np.int64)]
    for l in range(len(levels) - 1):
        P = levels[l].get("P")
        Ap, Aj, Ax, _ = _csr_view(P)
        n, nc = P.shape
        Ap = np.asarray(Ap, dtype=np.int64)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
        w = np.abs(np.asarray(Ax, dtype=np.float64))
        cols = np.asarray(Aj, dtype=np.int64)
        # strongest fine row of every coarse column (ties: the lowest row)
        order = np.lexsort((rows, -w, cols))
        first = np.ones(len(order), dtype=bool)
        first[1:] = cols[order][1:] != cols[order][:-1]
        oc = np.zeros(nc, dtype=np.int64)
        oc[cols[order][first]] = owners[l][rows[order][first]]
        owners.append(oc)
    return owners


class HipBackend(object):
    """csr_stream kernels + vector kernels on torch CUDA tensors (the product path)."""

    def __init__(self, device):
        import torch
        from . import _lib
        self.torch = torch
        self._lib = _lib
        self.L = _lib.lib()
        self.device = int(device)
        if _lib.device_count() <= self.device:
            raise _lib.AmgDeviceError("no HIP device %d (no CPU fallback)" % self.device)
        torch.cuda.set_device(self.device)
        self.dev = torch.device("cuda", self.device)
        self.scratch = torch.zeros(1100, dtype=torch.float64, device=self.dev)
        self._mats = []

    def stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def vec(self, n):
        return self.torch.zeros(max(int(n), 1), dtype=self.torch.float64, device=self.dev)

    def ivec(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.dev)

    def from_host(self, t, a):
        t[:len(a)].copy_(self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)))

    def to_host(self, t, n):
        return t[:n].cpu().numpy()

    def mat(self, nrows, ncols, Ap, Aj, Ax):
        Ap = np.ascontiguousarray(Ap, dtype=np.intc)
        Aj = np.ascontiguousarray(Aj, dtype=np.intc)
        Ax = np.ascontiguousarray(Ax, dtype=np.float64)
        h = self.L.amg_mat_create(self.device, int(nrows), int(ncols), self._lib.ip(Ap), self._lib.ip(Aj),
                                  self._lib.dp(Ax))
        if not h:
            raise self._lib.AmgError(self.L.amg_last_error().decode())
        self._mats.append(h)
        return h

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def apply(self, m, mode, xg, b, v2, out, out2, c0, gscale=1.0):
        self._lib.check(self.L.amg_mat_apply(m, mode, self._p(xg), self._p(b), self._p(v2), self._p(out),
                                             self._p(out2), float(c0), float(gscale), self.stream()))

    def apply_rows(self, m, mode, lo, hi, xg, b, v2, out, out2, c0, gscale=1.0):
        if hi > lo:
            self._lib.check(self.L.amg_mat_apply_rows(m, mode, int(lo), int(hi), self._p(xg), self._p(b), self._p(v2),
                                                      self._p(out), self._p(out2), float(c0), float(gscale),
                                                      self.stream()))

    def axpy_scaled(self, x, r, c, n):
        self._lib.check(self.L.amg_dev_axpy_scaled(x.data_ptr(), r.data_ptr(), float(c), int(n), self.stream()))

    def scale(self, out, inp, c, n):
        self._lib.check(self.L.amg_dev_scale(out.data_ptr(), inp.data_ptr(), float(c), int(n), self.stream()))

    def axpy(self, x, h, n):
        self._lib.check(self.L.amg_dev_axpy(x.data_ptr(), h.data_ptr(), int(n), self.stream()))

    def gather(self, out, inp, idx, n):
        self._lib.check(self.L.amg_dev_gather(out.data_ptr(), inp.data_ptr(), idx.data_ptr(), int(n), self.stream()))

    def build_gs(self, m, order):
        if order is None:
            self._lib.check(self.L.amg_mat_build_gs(m, None, 0))
        else:
            o = np.ascontiguousarray(order, dtype=np.intc)
            self._lib.check(self.L.amg_mat_build_gs(m, self._lib.ip(o), len(o)))

    def gs_sweep(self, m, x, b, reverse, bsr1):
        self._lib.check(self.L.amg_mat_gs_sweep(m, x.data_ptr(), b.data_ptr(), int(reverse), int(bsr1), self.stream()))

    def sumsq(self, x, n, out):
        """out[0] = sum of squares of x[:n] (device tensor of 1 double)"""
        self._lib.check(self.L.amg_dev_dot(x.data_ptr(), x.data_ptr(), int(n), self.scratch.data_ptr(),
                                           out.data_ptr(), self.stream()))

    def dense(self, Mt, b, x, n):
        self._lib.check(self.L.amg_dev_dense_apply(Mt.data_ptr(), b.data_ptr(), x.data_ptr(), int(n), self.stream()))

    def zero(self, t, n):
        t[:n].zero_()

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def close(self):
        for h in self._mats:
            self.L.amg_mat_destroy(h)
        self._mats = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Level(object):
    pass


class DistributedSolver(object):
    """multilevel_solver.solve() on a row-partitioned hierarchy.

    levels : list of dicts {A, P, R, pre, post} with GLOBAL scipy operators (P/R/pre/post absent
             on the coarsest level) and smoother descriptors as in smoothing.py's `.desc`
    coarse_dense : dense coarse operator (x = M b) or None
    group : torch.distributed group for the halos / all-reduce (device tensors)
    host_group : group able to move CPU tensors (gloo) for the setup-time index exchange
    owners : None (contiguous row blocks: level 0 cut evenly, coarser levels following it through P), or a list with
             one integer array per level -- owners[l][i] = rank that owns unknown i of level l -- for OWNERSHIP BY INDEX
             SET (arbitrary numberings; `owners_by_aggregate` derives the coarse levels from level 0).  A rank's local
             numbering is then its owned indices in ascending order: internally the level is relabelled so that every
             rank's set becomes a contiguous block (rows reordered, column indices relabelled, the entries of a row kept
             in stored order -- hence the same row sums), and everything below works on blocks.  `owned(l)` lists a rank's
             indices; solve() takes and returns the entries of b / x at owned(0), in that order.
    """

    def __init__(self, levels, coarse_dense, backend, rank, world, group=None, host_group=None,
                 replicate_below=500000, owners=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.be = backend
        self.rank, self.world = rank, world
        self.group, self.host_group = group, host_group if host_group is not None else group
        import os as _os
        self.overlap = _os.environ.get("AMG_DIST_OVERLAP", "1") != "0"
        self.nlevels = len(levels)
        # level 0 is cut evenly; every coarser level follows the cut of the level above through P (coarse_bounds).
        # AMG_DIST_EVEN_SPLIT=1 cuts every level evenly instead (the round-1 behaviour, kept for A/B).
        self.perm = None                 # index-set ownership: perm[l][k] = original index of relabelled unknown k
        self.bounds = [split_rows(levels[0]["A"].shape[0], world)]
        even = _os.environ.get("AMG_DIST_EVEN_SPLIT", "0") != "0"
        for l in range(1, self.nlevels):
            if even or levels[l - 1].get("P") is None:
                self.bounds.append(split_rows(levels[l]["A"].shape[0], world))
            else:
                self.bounds.append(coarse_bounds(levels[l - 1]["P"], self.bounds[l - 1]))
        if owners is not None:
            if len(owners) != self.nlevels:
                raise ValueError("owners: one array per level")
            self.perm, self.bounds = [], []
            for l in range(self.nlevels):
                o = np.asarray(owners[l], dtype=np.int64)
                n_l = levels[l]["A"].shape[0]
                if o.shape != (n_l,) or (n_l and (o.min() < 0 or o.max() >= world)):
                    raise ValueError("owners[%d]: one rank in 0..%d per unknown" % (l, world - 1))
                self.perm.append(np.argsort(o, kind="stable"))          # a rank's indices in ascending order
                self.bounds.append(np.concatenate(([0], np.cumsum(np.bincount(o, minlength=world)))).astype(np.int64))
        # Coarse levels at or below `replicate_below` unknowns are REPLICATED: every rank holds them
        # whole and computes them redundantly (bit-identical everywhere), so they need no halo
        # exchange at all -- one all-gather of the restricted right-hand side enters the replicated
        # part of the hierarchy, nothing is needed to leave it.  Exchange latency, not bandwidth,
        # is what the small levels cost.
        sizes = [L["A"].shape[0] for L in levels]
        self.first_rep = self.nlevels
        if world > 1:
            for l in range(self.nlevels - 1, 0, -1):
                if sizes[l] <= replicate_below:
                    self.first_rep = l
                else:
                    break
        self.lv = []
        self.keep_residual = _os.environ.get("AMG_KEEP_RESIDUAL", "1") != "0"
        self._r_kept = False
        # Engine: with the HIP backend the whole partitioned cycle runs in libamgcore_hip.so (hier.hip with a
        # communicator, comm.hip): transport "peer" = IPC-mapped arenas, GPU-to-GPU pushes and flag kernels on the
        # hierarchy's stream (graph-capturable); "rccl" = grouped ncclSend/ncclRecv + ncclAllReduce from C++.
        # "python" keeps the cycle in this file with torch.distributed collectives (what the CPU tests drive).
        want = _os.environ.get("AMG_DIST_TRANSPORT", "peer" if isinstance(backend, HipBackend) else "python")
        self.transport = want if isinstance(backend, HipBackend) else "python"
        self.native = None
        self._build(levels, coarse_dense)

    # ------------------------------------------------------------------ setup
    def _build(self, levels, coarse_dense):
        torch, dist = self.torch, self.dist
        r, W = self.rank, self.world
        nl = self.nlevels
        fr = self.first_rep
        own = [(int(b[r]), int(b[r + 1])) if l < fr else (0, int(b[-1])) for l, b in enumerate(self.bounds)]
        # rows of R_l a rank computes: its slice of the coarse rows, also at the transition into the
        # replicated part (the slices are then all-gathered); everything inside the replicated part
        rrows = [((int(self.bounds[l + 1][r]), int(self.bounds[l + 1][r + 1])) if l + 1 <= fr else own[l + 1])
                 for l in range(nl - 1)]
        if fr < nl and fr >= 1:
            rrows[fr - 1] = (int(self.bounds[fr][r]), int(self.bounds[fr][r + 1]))
        self.rrows = rrows
        # index-set ownership: partitioned levels are relabelled (replicated ones keep their numbering)
        relabel = [None] * nl
        if self.perm is not None:
            for l in range(min(fr, nl)):
                inv = np.empty(len(self.perm[l]), dtype=np.int64)
                inv[self.perm[l]] = np.arange(len(self.perm[l]), dtype=np.int64)
                relabel[l] = inv
            for l in range(fr, nl):
                self.perm[l] = None
        self._relabel = relabel

        def rows_of(M, l_rows, lo, hi, l_cols):
            if relabel[l_rows] is None and relabel[l_cols] is None:
                return local_rows(M, lo, hi)
            ids = np.arange(lo, hi, dtype=np.int64) if relabel[l_rows] is None else self.perm[l_rows][lo:hi]
            return local_rows_of(M, ids, relabel[l_cols])
        loc = []
        for l, L in enumerate(levels):
            d = {"A": rows_of(L["A"], l, own[l][0], own[l][1], l)}
            d["bsr"] = bool(_csr_view(L["A"])[3])
            if l < nl - 1:
                d["P"] = rows_of(L["P"], l, own[l][0], own[l][1], l + 1)        # fine rows, coarse columns (V_{l+1})
                d["R"] = rows_of(L["R"], l + 1, rrows[l][0], rrows[l][1], l)    # coarse rows, fine columns (V_l)
            loc.append(d)
        # union halo of every vector space
        halos = []
        for l in range(nl):
            lo, hi = own[l]
            cols = [loc[l]["A"][1]]
            if l < nl - 1:
                cols.append(loc[l]["R"][1])
            if l > 0:
                cols.append(loc[l - 1]["P"][1])
            c = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)
            c = c[(c < lo) | (c >= hi)]
            halos.append(np.unique(c))
        # tell every owner what we need (setup-time, host tensors)
        for l in range(nl):
            lv = _Level()
            lo, hi = own[l]
            lv.n_own = hi - lo
            H = halos[l]
            lv.n_halo = len(H)
            owner = np.searchsorted(self.bounds[l], H, side="right") - 1
            recv_counts = np.bincount(owner, minlength=W).astype(np.int64)
            send_counts = np.zeros(W, dtype=np.int64)
            if W > 1:
                tc = torch.from_numpy(recv_counts.copy())
                ts = torch.zeros(W, dtype=torch.int64)
                dist.all_to_all_single(ts, tc, group=self.host_group)
                send_counts = ts.numpy().copy()
                req = torch.from_numpy(H.astype(np.int64))
                got = torch.zeros(int(send_counts.sum()), dtype=torch.int64)
                dist.all_to_all_single(got, req, [int(v) for v in send_counts], [int(v) for v in recv_counts],
                                       group=self.host_group)
                send_idx = got.numpy() - lo
                if len(send_idx) and (send_idx.min() < 0 or send_idx.max() >= lv.n_own):
                    raise RuntimeError("halo request outside the owner's range")
            else:
                send_idx = np.zeros(0, dtype=np.int64)
            lv.recv_counts = [int(v) for v in recv_counts]
            lv.send_counts = [int(v) for v in send_counts]
            lv.n_send = int(send_counts.sum())
            lv.send_idx_host = np.ascontiguousarray(send_idx, dtype=np.intc)
            native = self.transport != "python"
            lv.send_idx = self.be.ivec(send_idx) if (lv.n_send and not native) else None
            lv.sendbuf = self.be.vec(lv.n_send) if not native else None
            lv.halo_ids = H
            lv.comm = (lv.n_halo + lv.n_send) > 0
            if W > 1:
                flag = torch.tensor([1.0 if lv.comm else 0.0], dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.host_group)
                lv.comm = bool(flag.item() > 0.5)      # every rank joins the exchange or none does
            self.lv.append(lv)
        # local operators with renumbered columns
        def renum(cols, l):
            lo, hi = own[l]
            out = np.empty(len(cols), dtype=np.int64)
            m = (cols >= lo) & (cols < hi)
            out[m] = cols[m] - lo
            out[~m] = self.lv[l].n_own + np.searchsorted(halos[l], cols[~m])
            return out
        if self.transport != "python":
            self._comm = self._h = None
            try:
                self._build_native(levels, coarse_dense, loc, own, rrows, renum)
            except Exception:
                # a half-built engine must not outlive the failed constructor (its IPC arena would stay exported)
                from . import _lib
                Lb = _lib.lib()
                if self._h:
                    Lb.amg_hier_destroy(self._h)
                if self._comm:
                    Lb.amg_comm_destroy(self._comm)
                self._comm = self._h = None
                self.native = None
                raise
            return
        for l, L in enumerate(levels):
            lv = self.lv[l]
            n_ext = lv.n_own + lv.n_halo
            Ap, Aj, Ax = loc[l]["A"]
            Ajl = renum(Aj, l)
            lv.A = self.be.mat(lv.n_own, n_ext, Ap, Ajl, Ax)
            # rows [i0, i1) read no halo entry: they can run while the halo exchange is in flight
            lv.i0, lv.i1 = 0, lv.n_own
            if lv.n_halo and lv.n_own:
                touches = np.zeros(lv.n_own + 1, dtype=np.int64)
                hal = np.nonzero(Ajl >= lv.n_own)[0]
                rows_h = np.unique(np.searchsorted(Ap, hal, side="right") - 1)
                if len(rows_h):
                    # largest halo-free window: between the last boundary row of the leading run and the
                    # first of the trailing run, taken at the biggest gap
                    gaps = np.diff(np.concatenate(([-1], rows_h, [lv.n_own])))
                    k = int(np.argmax(gaps))
                    lv.i0 = int(rows_h[k - 1]) + 1 if k > 0 else 0
                    lv.i1 = int(rows_h[k]) if k < len(rows_h) else lv.n_own
                del touches
            lv.overlap = (lv.i1 - lv.i0) >= 0.5 * max(lv.n_own, 1) and lv.n_halo > 0
            lv.A_bsr = loc[l]["bsr"]
            lv.nnzA = len(Ax)
            if l < nl - 1:
                nxt = self.lv[l + 1]
                Pp, Pj, Px = loc[l]["P"]
                lv.P = self.be.mat(lv.n_own, nxt.n_own + nxt.n_halo, Pp, renum(Pj, l + 1), Px)
                Rp, Rj, Rx = loc[l]["R"]
                lv.R = self.be.mat(rrows[l][1] - rrows[l][0], n_ext, Rp, renum(Rj, l), Rx)
                lv.rslice = self.be.vec(rrows[l][1] - rrows[l][0]) if (l + 1 == fr and W > 1) else None
                lv.pre, lv.post = L.get("pre"), L.get("post")
                for s in (lv.pre, lv.post):
                    nm = None if s is None else s.get("name")
                    if nm not in (None, "jacobi", "polynomial", "gauss_seidel", "gauss_seidel_indexed"):
                        raise NotImplementedError(
                            "smoother %r has no partitioned form; offered: jacobi / polynomial (chebyshev, "
                            "richardson) / gauss_seidel and gauss_seidel_indexed as HYBRID sweeps (Gauss-Seidel "
                            "inside a rank, Jacobi across ranks) / None" % (nm,))
                    if nm in ("gauss_seidel", "gauss_seidel_indexed") and W > 1 and not getattr(self, "allow_hybrid", True):
                        raise NotImplementedError("hybrid Gauss-Seidel disabled")
                    if nm == "gauss_seidel_indexed":
                        idx = np.asarray(s["indices"], dtype=np.int64)
                        if self._relabel[l] is not None:
                            idx = self._relabel[l][idx]                  # the list in the relabelled numbering, order kept
                        lo_, hi_ = own[l]
                        s["_local_order"] = (idx[(idx >= lo_) & (idx < hi_)] - lo_).astype(np.intc)
                        self.be.build_gs(lv.A, s["_local_order"])
                    elif nm == "gauss_seidel" and not getattr(lv, "_gs_natural", False):
                        self.be.build_gs(lv.A, None)
                        lv._gs_natural = True
            for nm in ("x", "xalt", "b", "r", "h", "h2"):
                setattr(lv, nm, self.be.vec(n_ext))
        # coarse dense operator: replicated, applied redundantly on the gathered coarse rhs
        self.coarse_Mt = None
        nc = levels[-1]["A"].shape[0]
        self.nc = nc
        if coarse_dense is not None:
            Mt = np.ascontiguousarray(np.asarray(coarse_dense, dtype=np.float64).T)
            self.coarse_Mt = self.be.vec(nc * nc)
            self.be.from_host(self.coarse_Mt, Mt.ravel())
            self.coarse_full_b = self.be.vec(nc)
            self.coarse_full_x = self.be.vec(nc)
        self.nnz_coarse = int(levels[-1]["A"].nnz)
        self.acc = self.be.vec(1)

    # ------------------------------------------------------------------ native engine (C++ cycle + exchange)
    @staticmethod
    def _halo_free_window(Ap, Ajl, n_own):
        """largest run of rows [i0, i1) that read no halo column (they can run while the halo is in flight)"""
        hal = np.nonzero(Ajl >= n_own)[0]
        if not len(hal):
            return 0, n_own
        rows_h = np.unique(np.searchsorted(Ap, hal, side="right") - 1)
        gaps = np.diff(np.concatenate(([-1], rows_h, [n_own])))
        k = int(np.argmax(gaps))
        i0 = int(rows_h[k - 1]) + 1 if k > 0 else 0
        i1 = int(rows_h[k]) if k < len(rows_h) else n_own
        return i0, i1

    def _build_native(self, levels, coarse_dense, loc, own, rrows, renum):
        from . import _lib
        from .multilevel import _desc_struct
        torch, dist = self.torch, self.dist
        Lb = _lib.lib()
        W, r, nl, fr = self.world, self.rank, self.nlevels, self.first_rep
        be = self.be
        comm = Lb.amg_comm_create(r, W, be.device, 1 if self.transport == "rccl" else 0)
        if not comm:
            raise _lib.AmgError(Lb.amg_last_error().decode())
        self._comm = comm

        def gather_rows(row):
            """every rank's row of a count matrix -> W x W (dst-major), identical on all ranks"""
            row = np.asarray(row, dtype=np.int64)
            if W == 1:
                return row.reshape(1, 1)
            out = [torch.zeros(W, dtype=torch.int64) for _ in range(W)]
            dist.all_gather(out, torch.from_numpy(row.copy()), group=self.host_group)
            return np.stack([t.numpy() for t in out])

        def add_channel(matrix):
            m = np.ascontiguousarray(matrix, dtype=np.intc)
            ch = Lb.amg_comm_add_channel(comm, _lib.ip(m))
            if ch < 0:
                raise _lib.AmgError(Lb.amg_last_error().decode())
            return ch
        halo_ch = [-1] * nl
        for l in range(nl):
            lv = self.lv[l]
            if W > 1 and lv.comm:
                halo_ch[l] = add_channel(gather_rows(lv.recv_counts))
        gather_ch, gather_rows_n = -1, 0
        if W > 1 and 1 <= fr <= nl - 1:
            sizes = np.diff(self.bounds[fr]).astype(np.int64)             # slice of every source rank
            gather_ch = add_channel(np.tile(sizes[None, :], (W, 1)))
            gather_rows_n = int(sizes[r])
        coarse_ch = -1
        if W > 1 and fr > nl - 1 and coarse_dense is not None:
            sizes = np.diff(self.bounds[-1]).astype(np.int64)             # the coarsest level is partitioned too
            coarse_ch = add_channel(np.tile(sizes[None, :], (W, 1)))
        reduce_ch = add_channel(np.ones((W, W), dtype=np.int64))
        handle = np.zeros(64, dtype=np.uint8)
        _lib.check(Lb.amg_comm_commit(comm, handle.ctypes.data))
        if self.transport == "rccl":
            import os as _os
            libpath = _os.environ.get("AMG_RCCL_LIB") or _os.path.join(_os.path.dirname(torch.__file__), "lib", "librccl.so")
            if not _os.path.exists(libpath):
                libpath = "librccl.so"
            uid = np.zeros(128, dtype=np.uint8)
            if r == 0:
                _lib.check(Lb.amg_comm_rccl_unique_id(libpath.encode(), uid.ctypes.data))
            if W > 1:
                t = torch.from_numpy(uid)
                dist.broadcast(t, src=0, group=self.host_group)
                uid = t.numpy()
            _lib.check(Lb.amg_comm_rccl_init(comm, libpath.encode(), uid.ctypes.data))
        else:
            if W > 1:
                out = [torch.zeros(64, dtype=torch.uint8) for _ in range(W)]
                dist.all_gather(out, torch.from_numpy(handle.copy()), group=self.host_group)
                handles = np.concatenate([t.numpy() for t in out])
            else:
                handles = handle
            handles = np.ascontiguousarray(handles, dtype=np.uint8)
            err = None
            try:
                _lib.check(Lb.amg_comm_connect(comm, handles.ctypes.data))
            except Exception as e:          # noqa: BLE001 -- every rank must learn about it (no rank may wait forever)
                err = e
            if W > 1:
                # doubles as the barrier "every arena is mapped everywhere before anybody pushes"
                flag = torch.tensor([0.0 if err else 1.0], dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.host_group)
                if flag.item() < 0.5 and err is None:
                    err = RuntimeError("a peer could not map the IPC arenas")
            if err is not None:
                raise err
        h = Lb.amg_hier_create(nl, be.device)
        if not h:
            raise _lib.AmgError(Lb.amg_last_error().decode())
        self._h = h
        _lib.check(Lb.amg_hier_set_comm(h, comm, reduce_ch))
        keep = []

        def set_mat(l, which, nrows, ncols, Ap, Aj, Ax, bsr11):
            Ap = np.ascontiguousarray(Ap, dtype=np.intc)
            Aj = np.ascontiguousarray(Aj, dtype=np.intc)
            Ax = np.ascontiguousarray(Ax, dtype=np.float64)
            _lib.check(Lb.amg_hier_set_matrix(h, l, which, 1 if bsr11 else 0, int(nrows), int(ncols), 1, 1,
                                              Ap.ctypes.data, Aj.ctypes.data, Ax.ctypes.data, 0))
        for l, L in enumerate(levels):
            lv = self.lv[l]
            n_ext = lv.n_own + lv.n_halo
            Ap, Aj, Ax = loc[l]["A"]
            Ajl = renum(Aj, l)
            set_mat(l, 0, lv.n_own, n_ext, Ap, Ajl, Ax, loc[l]["bsr"])
            i0, i1 = self._halo_free_window(Ap, Ajl, lv.n_own) if lv.n_halo else (0, lv.n_own)
            lv.i0, lv.i1 = i0, i1
            _lib.check(Lb.amg_hier_set_partition(h, l, lv.n_own, lv.n_halo, halo_ch[l], _lib.ip(lv.send_idx_host), i0, i1))
            lv.nnzA = len(Ax)
            lv.A_bsr = loc[l]["bsr"]
            if l < nl - 1:
                nxt = self.lv[l + 1]
                if l + 1 == fr and gather_ch >= 0:
                    _lib.check(Lb.amg_hier_set_gather(h, l, gather_ch, gather_rows_n))
                Pp, Pj, Px = loc[l]["P"]
                set_mat(l, 1, lv.n_own, nxt.n_own + nxt.n_halo, Pp, renum(Pj, l + 1), Px, False)
                Rp, Rj, Rx = loc[l]["R"]
                set_mat(l, 2, rrows[l][1] - rrows[l][0], n_ext, Rp, renum(Rj, l), Rx, False)
                for which, side in ((0, "pre"), (1, "post")):
                    sdesc = L.get(side)
                    nm = None if sdesc is None else sdesc.get("name")
                    if nm not in (None, "jacobi", "polynomial", "gauss_seidel", "gauss_seidel_indexed"):
                        raise NotImplementedError(
                            "smoother %r has no partitioned form; offered: jacobi / polynomial (chebyshev, richardson) / "
                            "gauss_seidel and gauss_seidel_indexed as HYBRID sweeps / None" % (nm,))
                    d = None if sdesc is None else dict(sdesc)
                    if nm == "gauss_seidel_indexed":
                        idx = np.asarray(d["indices"], dtype=np.int64)
                        if self._relabel[l] is not None:
                            idx = self._relabel[l][idx]
                        lo_, hi_ = own[l]
                        d["indices"] = (idx[(idx >= lo_) & (idx < hi_)] - lo_).astype(np.intc)
                    _lib.check(Lb.amg_hier_set_smoother(h, l, which, _desc_struct(d, keep)))
            lv.pre, lv.post = L.get("pre"), L.get("post")
        nc = levels[-1]["A"].shape[0]
        self.nc = nc
        self.nnz_coarse = int(levels[-1]["A"].nnz)
        if coarse_dense is not None:
            M = np.ascontiguousarray(np.asarray(coarse_dense, dtype=np.float64))
            _lib.check(Lb.amg_hier_set_coarse_dense(h, _lib.dp(M), M.shape[0]))
        if coarse_ch >= 0:
            _lib.check(Lb.amg_hier_set_coarse_gather(h, coarse_ch, int(self.bounds[-1][r])))
        _lib.check(Lb.amg_hier_finalize(h))
        self.native = (Lb, h, comm)
        self._b_local = None

    def _native_solve(self, b_local, x_local, tol, maxiter, cycle, x_zero, fixed):
        from . import _lib
        Lb, h, comm = self.native
        res = np.zeros(maxiter + 2, dtype=np.float64)
        nres = C.c_int(0)
        flags = (1 if x_zero else 0) | (2 if fixed else 0)
        cyc = {"V": 0, "W": 1, "F": 2}.get(cycle)
        if cyc is None:
            raise NotImplementedError("AMLI cycles are not implemented on the partitioned path")
        _lib.check(Lb.amg_hier_solve(h, b_local.ctypes.data, x_local.ctypes.data, float(tol), int(maxiter), cyc,
                                     _lib.dp(res), C.byref(nres), flags))
        _lib.check(Lb.amg_hier_comm_check(h))
        return res[:nres.value]

    def operator_form(self, l):
        """storage form level l's local A is applied from: 0 CSR, 1 offset-pattern, 2 stencil"""
        if self.native is not None:
            return self.native[0].amg_hier_operator_form(self.native[1], int(l))
        return self.be.L.amg_mat_form(self.lv[l].A)

    def last_solve_ms(self):
        Lb, h, comm = self.native
        return Lb.amg_hier_last_solve_ms(h)

    def device_bytes(self):
        Lb, h, comm = self.native
        return int(Lb.amg_hier_device_bytes(h))

    def close(self, collective=True):
        """collective (default): every rank's arena stays mapped in its peers until all have stopped exchanging -- two
        barriers, so EVERY rank must call it.  collective=False: tear down this rank's objects alone (a transport that
        only some ranks could set up: the ranks whose constructor failed have nothing to close and join no barrier)."""
        if self.native is not None:
            Lb, h, comm = self.native
            if self.world > 1 and collective:
                self.dist.barrier(group=self.host_group)
            Lb.amg_hier_destroy(h)
            if self.world > 1 and collective:
                self.dist.barrier(group=self.host_group)
            Lb.amg_comm_destroy(comm)
            self.native = None

    # ------------------------------------------------------------------ communication
    def xapply(self, l, mode, v, b, v2, out, out2, c0, gscale=1.0):
        """exchange(l, v) followed by A_l applied with v as the gathered operand; when the level has a
        large halo-free row window, that window runs while the halo is in flight"""
        lv = self.lv[l]
        if self.world == 1 or not lv.comm or not (self.overlap and lv.overlap):
            self.exchange(l, v)
            self.be.apply(lv.A, mode, v, b, v2, out, out2, c0, gscale)
            return
        if lv.n_send:
            self.be.gather(lv.sendbuf, v, lv.send_idx, lv.n_send)
        work = self.dist.all_to_all_single(v[lv.n_own:lv.n_own + lv.n_halo], lv.sendbuf[:lv.n_send],
                                           lv.recv_counts, lv.send_counts, group=self.group, async_op=True)
        self.be.apply_rows(lv.A, mode, lv.i0, lv.i1, v, b, v2, out, out2, c0, gscale)
        work.wait()
        self.be.apply_rows(lv.A, mode, 0, lv.i0, v, b, v2, out, out2, c0, gscale)
        self.be.apply_rows(lv.A, mode, lv.i1, lv.n_own, v, b, v2, out, out2, c0, gscale)

    def exchange(self, l, v):
        """refresh the halo part of v (a V_l vector) from its owners"""
        lv = self.lv[l]
        if self.world == 1 or not lv.comm:
            return
        if lv.n_send:
            self.be.gather(lv.sendbuf, v, lv.send_idx, lv.n_send)
        self.dist.all_to_all_single(v[lv.n_own:lv.n_own + lv.n_halo], lv.sendbuf[:lv.n_send],
                                    lv.recv_counts, lv.send_counts, group=self.group)

    def global_norm(self, v, n):
        self.be.sumsq(v, n, self.acc)
        if self.world > 1:
            self.dist.all_reduce(self.acc, group=self.group)
        return float(np.sqrt(self.be.to_host(self.acc, 1)[0]))

    # ------------------------------------------------------------------ smoothers (relaxation.py)
    def relax(self, l, s, xname, bvec, x_zero, r_ready=False):
        lv = self.lv[l]
        if s is None or s.get("name") is None:
            return
        n = lv.n_own
        it = int(s.get("iterations", 1))
        if s["name"] in ("gauss_seidel", "gauss_seidel_indexed"):
            # hybrid sweep: halo refreshed once per directional sweep and frozen during it
            # (on a replicated level there is no halo and the sweep is the exact sequential one)
            x = getattr(lv, xname)
            bsr1 = lv.A_bsr and s["name"] == "gauss_seidel"
            sweep = s.get("sweep", "forward")
            for _ in range(it):
                if sweep in ("forward", "symmetric"):
                    self.exchange(l, x)
                    self.be.gs_sweep(lv.A, x, bvec, False, bsr1)
                if sweep in ("backward", "symmetric"):
                    self.exchange(l, x)
                    self.be.gs_sweep(lv.A, x, bvec, True, bsr1)
            return
        if s["name"] == "jacobi":
            for _ in range(it):
                x, xalt = getattr(lv, xname), lv.xalt
                self.xapply(l, JACOBI_BSR1 if lv.A_bsr else JACOBI, x, bvec, x, xalt, None, s["omega"])
                setattr(lv, xname, xalt)
                lv.xalt = x
            return
        co = s["coefficients"]
        for k in range(it):
            # h = c0*r is gathered from r on the fly (one rounding either way); see hier.hip relax()
            x = getattr(lv, xname)
            if x_zero:
                rvec = bvec
            elif r_ready and k == 0:
                rvec = lv.r                # b - A x of this very x, left there by the residual norm
            else:
                self.xapply(l, RESIDUAL, x, bvec, None, lv.r, None, 0.0)
                rvec = lv.r
            if len(co) == 1:
                self.be.axpy_scaled(x, rvec, co[0], n)
            else:
                if len(co) == 2:
                    self.xapply(l, POLY_LAST, rvec, rvec, x, x, None, co[1], co[0])
                else:
                    hh, hn = lv.h, lv.h2
                    self.xapply(l, POLY_STEP, rvec, rvec, None, hh, None, co[1], co[0])
                    for c in co[2:-1]:
                        self.xapply(l, POLY_STEP, hh, rvec, None, hn, None, c)
                        hh, hn = hn, hh
                    self.xapply(l, POLY_LAST, hh, rvec, x, x, None, co[-1])
            x_zero = False

    def coarse_solve(self):
        lv = self.lv[-1]
        if self.nnz_coarse == 0 or self.coarse_Mt is None:
            self.be.zero(lv.x, lv.n_own)
            return
        nc = self.nc
        if self.first_rep <= self.nlevels - 1:
            # the coarsest level is replicated: b is already whole on every rank
            self.be.dense(self.coarse_Mt, lv.b, lv.x, nc)
            return
        if self.world > 1:
            counts = [int(self.bounds[-1][p + 1] - self.bounds[-1][p]) for p in range(self.world)]
            self._all_gather_uneven(lv.b[:lv.n_own], counts)
        else:
            self.coarse_full_b[:nc].copy_(lv.b[:nc])
        self.be.dense(self.coarse_Mt, self.coarse_full_b, self.coarse_full_x, nc)
        lo = int(self.bounds[-1][self.rank])
        lv.x[:lv.n_own].copy_(self.coarse_full_x[lo:lo + lv.n_own])

    def _all_gather_uneven(self, mine, counts, out=None):
        # all_to_all with every rank sending its whole slice to everybody
        W = self.world
        inp = mine.repeat(W) if mine.numel() else mine
        if out is None:
            out = self.coarse_full_b[:self.nc]
        self.dist.all_to_all_single(out, inp, counts, [int(mine.numel())] * W, group=self.group)

    # ------------------------------------------------------------------ cycle (multilevel.py:473-548)
    def cycle(self, l, cyc, x_zero, r_ready=False):
        lv, nx = self.lv[l], self.lv[l + 1]
        self.relax(l, lv.pre, "x", lv.b, x_zero, r_ready)
        self.xapply(l, RESIDUAL, lv.x, lv.b, None, lv.r, None, 0.0)
        self.exchange(l, lv.r)
        if lv.rslice is not None:
            # entering the replicated part: each rank restricts its slice of the coarse rows, then all gather
            self.be.apply(lv.R, MATVEC, lv.r, None, None, lv.rslice, None, 0.0)
            counts = [int(self.bounds[l + 1][p + 1] - self.bounds[l + 1][p]) for p in range(self.world)]
            lo_, hi_ = self.rrows[l]
            self._all_gather_uneven(lv.rslice[:hi_ - lo_], counts, out=nx.b[:nx.n_own])
        else:
            self.be.apply(lv.R, MATVEC, lv.r, None, None, nx.b, None, 0.0)
        self.be.zero(nx.x, nx.n_own)
        if l == self.nlevels - 2:
            self.coarse_solve()
        elif cyc == "V":
            self.cycle(l + 1, "V", True)
        elif cyc == "W":
            self.cycle(l + 1, cyc, True)
            self.cycle(l + 1, cyc, False)
        elif cyc == "F":
            self.cycle(l + 1, cyc, True)
            self.cycle(l + 1, "V", False)
        else:
            raise NotImplementedError("AMLI cycles are not implemented on the device path")
        self.exchange(l + 1, nx.x)
        self.be.apply(lv.P, MATVEC_ACC, nx.x, None, None, lv.x, None, 0.0)
        self.relax(l, lv.post, "x", lv.b, False)

    def _keeps_residual(self):
        # the residual of the convergence test stays in lv[0].r; a polynomial pre-smoother on level 0
        # starts from exactly that vector (relaxation.py:655), so it need not be formed twice
        pre = self.lv[0].pre if self.nlevels > 1 else None
        return bool(self.keep_residual and pre is not None and pre.get("name") == "polynomial"
                    and int(pre.get("iterations", 1)) >= 1)

    def residual_norm(self):
        lv = self.lv[0]
        self.xapply(0, RESIDUAL, lv.x, lv.b, None, lv.r, None, 0.0)
        self._r_kept = self._keeps_residual()
        return self.global_norm(lv.r, lv.n_own)

    def set_problem(self, b_local, x0_local=None):
        lv = self.lv[0]
        self.be.from_host(lv.b, b_local)
        if x0_local is None:
            self.be.zero(lv.x, lv.n_own)
        else:
            self.be.from_host(lv.x, x0_local)
        self._r_kept = False

    def iterate(self, cyc, x_zero):
        r_ready, self._r_kept = self._r_kept, False
        if self.nlevels == 1:
            self.coarse_solve()
        else:
            self.cycle(0, cyc, x_zero, r_ready)

    def owned(self, l=0):
        """the indices of level l's unknowns this rank owns, in its local order (ascending)"""
        lo, hi = int(self.bounds[l][self.rank]), int(self.bounds[l][self.rank + 1])
        if l >= self.first_rep:
            return np.arange(int(self.bounds[l][-1]), dtype=np.int64)
        if self.perm is not None and self.perm[l] is not None:
            return np.asarray(self.perm[l][lo:hi], dtype=np.int64)
        return np.arange(lo, hi, dtype=np.int64)

    def solve(self, b_local, x0_local=None, tol=1e-5, maxiter=100, cycle="V", fixed=False):
        """multilevel.py:316-471 on the local slices; returns (x_local, residuals)"""
        cycle = str(cycle).upper()
        if self.native is not None:
            x_zero = x0_local is None or not np.any(x0_local)
            if self.world > 1:                      # x0 == 0 must hold on EVERY rank for the static shortcut
                flag = self.torch.tensor([1.0 if x_zero else 0.0], dtype=self.torch.float64)
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.host_group)
                x_zero = bool(flag.item() > 0.5)
            b1 = np.ascontiguousarray(b_local, dtype=np.float64)
            x1 = np.zeros(self.lv[0].n_own) if x0_local is None else np.array(x0_local, dtype=np.float64)
            res = self._native_solve(b1, x1, tol, maxiter, cycle, x_zero, fixed)
            return x1, [float(v) for v in res]
        self.set_problem(b_local, x0_local)
        lv = self.lv[0]
        normb = self.global_norm(lv.b, lv.n_own)
        if normb != 0:
            tol = tol * normb
        res = [self.residual_norm()]
        x_zero = x0_local is None or not np.any(x0_local)
        # x0 == 0 must hold on EVERY rank for the static shortcut
        if self.world > 1:
            flag = self.torch.tensor([1.0 if x_zero else 0.0], dtype=self.torch.float64)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.host_group)
            x_zero = bool(flag.item() > 0.5)
        if fixed:
            res.extend(self.run_fixed(maxiter, cycle, x_zero))
        else:
            while len(res) <= maxiter and res[-1] > tol:
                self.iterate(cycle, x_zero)
                x_zero = False
                res.append(self.residual_norm())
        return self.be.to_host(self.lv[0].x, lv.n_own), res

    def run_fixed(self, steps, cycle="V", x_zero=False):
        """exactly `steps` iterations (cycle + residual norm) with no host synchronisation inside:
        the squared norms are all-reduced into a device history and read once at the end"""
        lv = self.lv[0]
        hist = self.be.vec(steps)
        for k in range(steps):
            self.iterate(cycle, x_zero)
            x_zero = False
            self.xapply(0, RESIDUAL, lv.x, lv.b, None, lv.r, None, 0.0)
            self._r_kept = self._keeps_residual()
            self.be.sumsq(lv.r, lv.n_own, hist[k:k + 1])
        if self.world > 1 and steps:
            self.dist.all_reduce(hist[:steps], group=self.group)
        return [float(v) for v in np.sqrt(self.be.to_host(hist, steps))]


def levels_from_ml(ml):
    """global level dicts (+ dense coarse operator) from a pyamg_amd.multilevel_solver"""
    levels = []
    for lvl in ml.levels:
        L = {"A": lvl.A}
        if hasattr(lvl, "P"):
            L["P"], L["R"] = lvl.P, lvl.R
            for side, fn in (("pre", getattr(lvl, "presmoother", None)), ("post", getattr(lvl, "postsmoother", None))):
                d = getattr(fn, "desc", None)
                if fn is not None and d is None:
                    raise NotImplementedError("smoother without a device descriptor")
                L[side] = dict(d) if d is not None else None
        levels.append(L)
    kind, M = ml.coarse_solver.device_form(ml.levels[-1].A)
    if kind not in ("dense", "none"):
        raise NotImplementedError("partitioned path: dense coarse solvers only")
    return levels, (M if kind == "dense" else None)


# --------------------------------------------------------------------------- shipping a hierarchy between ranks
def save_levels(path, levels, coarse):
    """rank 0 -> shared directory (e.g. /dev/shm/...): flat .npy files the other ranks memory-map"""
    import json
    import os
    os.makedirs(path, exist_ok=True)
    meta = {"nlevels": len(levels), "levels": []}
    for l, L in enumerate(levels):
        m = {}
        for nm in ("A", "P", "R"):
            if nm in L:
                Ap, Aj, Ax, isb = _csr_view(L[nm])
                np.save(os.path.join(path, "%s%d_p.npy" % (nm, l)), np.asarray(Ap))
                np.save(os.path.join(path, "%s%d_j.npy" % (nm, l)), np.asarray(Aj))
                np.save(os.path.join(path, "%s%d_x.npy" % (nm, l)), np.asarray(Ax))
                m[nm] = {"shape": list(L[nm].shape), "bsr": bool(isb)}
        for side in ("pre", "post"):
            if side in L:
                d = L[side]
                if d is None:
                    m[side] = None
                    continue
                m[side] = {}
                for k, v in d.items():
                    if isinstance(v, np.ndarray) and v.size > 64:
                        # (an index list of a multicolour ordering has one entry per unknown: a file of its own)
                        fn = "%s%d_%s.npy" % (side, l, k)
                        np.save(os.path.join(path, fn), v)
                        m[side][k] = {"__npy__": fn}
                    else:
                        m[side][k] = v if not isinstance(v, np.ndarray) else v.tolist()
        meta["levels"].append(m)
    if coarse is not None:
        np.save(os.path.join(path, "coarse.npy"), np.asarray(coarse))
    meta["coarse"] = coarse is not None
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)


def load_levels(path):
    import json
    import os
    meta = json.load(open(os.path.join(path, "meta.json")))
    levels = []
    for l, m in enumerate(meta["levels"]):
        L = {}
        for nm in ("A", "P", "R"):
            if nm in m:
                Ap = np.load(os.path.join(path, "%s%d_p.npy" % (nm, l)), mmap_mode="r")
                Aj = np.load(os.path.join(path, "%s%d_j.npy" % (nm, l)), mmap_mode="r")
                Ax = np.load(os.path.join(path, "%s%d_x.npy" % (nm, l)), mmap_mode="r")
                L[nm] = _Lazy(Ap, Aj, Ax, tuple(m[nm]["shape"]), m[nm]["bsr"])
        for side in ("pre", "post"):
            if side in m:
                d = m[side]
                if d is not None:
                    d = {k: (np.load(os.path.join(path, v["__npy__"]), mmap_mode="r") if isinstance(v, dict) and "__npy__" in v else v)
                         for k, v in d.items()}
                L[side] = d
        levels.append(L)
    coarse = np.load(os.path.join(path, "coarse.npy")) if meta["coarse"] else None
    return levels, coarse
