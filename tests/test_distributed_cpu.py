"""CPU, gloo, world_size 2 and 3: the row-partitioned multi-rank cycle (pyamg_amd/distributed.py) --
partition bounds, per-level halo plans, exchange sequencing, replicated coarse solve, all-reduced
residual norm -- with the oracle as the local compute backend.  The iterates gathered from the
ranks must equal the single-process oracle solve BIT FOR BIT (Jacobi / polynomial cycles are
partition invariant); the residual norms may differ in the last bits (all-reduced partial sums)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_io
import oracle_lib


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, cycle, out_dir, replicate_below=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from pyamg_amd.distributed import DistributedSolver, load_levels, save_levels, split_rows
        g = golden_io.load_hier(case)
        # ship the hierarchy through the shared-directory format, as the multi-GPU bench does
        shared = os.path.join(out_dir, "hier")
        if rank == 0:
            save_levels(shared, g["levels"], g["coarse_pinv"])
        dist.barrier()
        levels, coarse = load_levels(shared)
        S = DistributedSolver(levels, coarse, OracleBackend(), rank, world, replicate_below=replicate_below)
        n = g["levels"][0]["A"].shape[0]
        bnd = split_rows(n, world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x0 = g["x0"][lo:hi] if np.any(g["x0"]) else None
        x, res = S.solve(g["b"][lo:hi], x0, tol=g["meta"]["tol"], maxiter=g["meta"]["maxiter"], cycle=cycle)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array(res))
            np.save(os.path.join(out_dir, "halo.npy"), np.array([lv.n_halo for lv in S.lv]))
            np.save(os.path.join(out_dir, "overlap.npy"), np.array([int(S.overlap and lv.overlap) for lv in S.lv]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world,rep", [("sa_jacobi_2d", 2, 0), ("sa_cheb2_3d", 2, 0), ("sa_cheb2_3d", 3, 0),
                                            ("sa_jacobi_2d", 2, 300), ("sa_cheb2_3d", 3, 600), ("sa_cheb2_3d", 2, 20),
                                            ("sa_cheb2_3d", 4, 600), ("sa_cheb2_3d", 8, 600), ("sa_jacobi_2d", 8, 0),
                                            ("sa_mixed_W_2d", None, 0)])
def test_partitioned_cycle_equals_single_process(case, world, rep, tmp_path):
    g = golden_io.load_hier(case)
    m = g["meta"]
    if world is None:
        # this hierarchy uses SOR (sequential): the partitioned path must refuse it, not approximate it
        from cpu_backend import OracleBackend
        from pyamg_amd.distributed import DistributedSolver
        with pytest.raises(NotImplementedError):
            DistributedSolver(g["levels"], g["coarse_pinv"], OracleBackend(), 0, 1)
        return
    port = _free_port()
    # rep = replicate_below: 0 partitions every level; larger values replicate the coarse levels
    # (sa_jacobi_2d: 2304/392/49/6 rows; sa_cheb2_3d: 4096/517/13)
    mp.spawn(_worker, args=(world, port, case, m["cycle"], str(tmp_path), rep), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    res = np.load(tmp_path / "res.npy")
    H = oracle_lib.Hierarchy(g["levels"], g["coarse_pinv"])
    xo, reso = H.solve(g["b"], x0=g["x0"], tol=m["tol"], maxiter=m["maxiter"], cycle=m["cycle"])
    assert len(res) == len(reso) == len(g["residuals"])
    assert np.array_equal(x, xo), np.abs(x - xo).max()
    golden_io.assert_history(res, g["residuals"], g["levels"][0]["A"], g["x"], g["b"])
    assert np.load(tmp_path / "halo.npy")[0] > 0          # the ranks really exchanged halos
    if world <= 3:                                        # (with 8 ranks a slab is all boundary: no window)
        assert np.load(tmp_path / "overlap.npy")[0] == 1  # ... with the interior rows overlapped on level 0


def test_split_rows_and_local_rows():
    from pyamg_amd.distributed import local_rows, split_rows
    b = split_rows(10, 3)
    assert list(b) == [0, 3, 6, 10]
    g = golden_io.load_hier("sa_jacobi_2d")
    P = g["levels"][0]["P"]                                   # BSR(1,1)
    Ap, Aj, Ax = local_rows(P, 5, 9)
    Pc = P.tocsr()
    assert Ap[0] == 0 and Ap[-1] == Pc.indptr[9] - Pc.indptr[5]
    assert np.array_equal(Aj, P.indices[P.indptr[5]:P.indptr[9]])


# ---------------------------------------------------------------------------
# Hybrid Gauss-Seidel (configuration C4): GS inside a rank, Jacobi across ranks.  Oracle =
# partition-emulating run built from the reference kernel's row_start/row_stop slicing on a
# frozen copy, written out independently here.
# ---------------------------------------------------------------------------
def _hybrid_sweep(lib, A, x, b, bounds, reverse, bsr, indices=None):
    """indices: the index list of gauss_seidel_indexed (multicolour ordering) -- every partition relaxes ITS entries of
    the list, in list order, on the frozen copy (relaxation.h:395-430 with the partition's sub-list)"""
    from oracle_lib import dp, ip
    Ac = A.tocsr() if not bsr else A
    Ap = np.ascontiguousarray(A.indptr, dtype=np.intc); Aj = np.ascontiguousarray(A.indices, dtype=np.intc)
    Ax = np.ascontiguousarray(np.ravel(A.data), dtype=np.float64)
    frozen = x.copy()
    for p in range(len(bounds) - 1):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        xp = frozen.copy()
        rs, re, rt = (hi - 1, lo - 1, -1) if reverse else (lo, hi, 1)
        if indices is not None:
            idx = np.asarray(indices, dtype=np.int64)
            sub = np.ascontiguousarray(idx[(idx >= lo) & (idx < hi)], dtype=np.intc)
            Acs = A.tocsr()
            Sp = np.ascontiguousarray(Acs.indptr, dtype=np.intc); Sj = np.ascontiguousarray(Acs.indices, dtype=np.intc)
            Sx = np.ascontiguousarray(Acs.data, dtype=np.float64)
            rs, re, rt = (len(sub) - 1, -1, -1) if reverse else (0, len(sub), 1)
            lib.oracle_gauss_seidel_indexed(ip(Sp), ip(Sj), dp(Sx), dp(xp), dp(b), ip(sub), rs, re, rt)
        elif bsr:
            lib.oracle_bsr_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(xp), dp(b), rs, re, rt, 1)
        else:
            lib.oracle_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(xp), dp(b), rs, re, rt)
        x[lo:hi] = xp[lo:hi]


def _hybrid_cycle(lib, levels, coarse, bounds, l, x, b):
    import scipy.sparse as sps
    L = levels[l]
    A = L["A"]; bsr = sps.isspmatrix_bsr(A)
    for side in ("pre", "post"):
        if side == "post":
            r = b - A * x
            cb = L["R"] * r
            cx = np.zeros_like(cb)
            if l == len(levels) - 2:
                for i in range(len(cb)):
                    s = 0.0
                    for k in range(len(cb)):
                        s += coarse[i, k] * cb[k]
                    cx[i] = s
            else:
                _hybrid_cycle(lib, levels, coarse, bounds, l + 1, cx, cb)
            x += L["P"] * cx
        sm = L[side]
        idx = sm.get("indices") if sm.get("name") == "gauss_seidel_indexed" else None
        for _ in range(int(sm.get("iterations", 1))):
            if sm["sweep"] in ("forward", "symmetric"):
                _hybrid_sweep(lib, A, x, b, bounds[l], False, bsr, idx)
            if sm["sweep"] in ("backward", "symmetric"):
                _hybrid_sweep(lib, A, x, b, bounds[l], True, bsr, idx)


def _worker_hybrid(rank, world, port, case, out_dir, rep=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from pyamg_amd.distributed import DistributedSolver, split_rows
        g = golden_io.load_hier(case)
        S = DistributedSolver(g["levels"], g["coarse_pinv"], OracleBackend(), rank, world, replicate_below=rep)
        n = g["levels"][0]["A"].shape[0]
        bnd = split_rows(n, world); lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(g["b"][lo:hi], None, tol=0.0, maxiter=3, cycle="V", fixed=True)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
    finally:
        dist.destroy_process_group()


def hybrid_bounds(g, world, rep):
    """partition of every level as DistributedSolver makes it: contiguous row blocks, except that levels at or
    below `rep` unknowns (from the first such level down) are replicated = one block"""
    from pyamg_amd.distributed import split_rows
    sizes = [L["A"].shape[0] for L in g["levels"]]
    first_rep = len(sizes)
    for l in range(len(sizes) - 1, 0, -1):
        if sizes[l] <= rep:
            first_rep = l
        else:
            break
    # level 0 evenly, every coarser partitioned level following the level above through P (distributed.coarse_bounds)
    from pyamg_amd.distributed import coarse_bounds
    bounds = [split_rows(sizes[0], world)]
    for l in range(1, len(sizes)):
        bounds.append(coarse_bounds(g["levels"][l - 1]["P"], bounds[l - 1]))
    return [bounds[l] if l < first_rep else np.array([0, n]) for l, n in enumerate(sizes)]


@pytest.mark.parametrize("case,rep", [("sa_gs_3d", 0), ("rs_gs_2d", 0), ("rs_gs_2d", 450)])
def test_hybrid_gauss_seidel_matches_partition_emulation(case, rep, tmp_path):
    world = 2
    g = golden_io.load_hier(case)
    mp.spawn(_worker_hybrid, args=(world, _free_port(), case, str(tmp_path), rep), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    lib = oracle_lib.load()
    bounds = hybrid_bounds(g, world, rep)
    xe = np.zeros_like(g["b"])
    for _ in range(3):
        _hybrid_cycle(lib, g["levels"], g["coarse_pinv"], bounds, 0, xe, np.ascontiguousarray(g["b"]))
    assert np.array_equal(x, xe), np.abs(x - xe).max()
    # and it is a different iteration from sequential GS (so the test cannot pass vacuously)
    H = oracle_lib.Hierarchy(g["levels"], g["coarse_pinv"])
    xs, _ = H.solve(g["b"], tol=0.0, maxiter=3)
    assert not np.array_equal(x, xs)
    assert np.linalg.norm(x - xs) < 0.5 * np.linalg.norm(xs)


def multicolour_hierarchy(path, grid=(14, 13, 12)):
    """BASELINE configuration C4's smoother at a size the emulation affords: SA on a 3-D Poisson operator with
    multicolour Gauss-Seidel (gauss_seidel_indexed over a greedy colouring), written where the rank processes find it"""
    from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
    from pyamg_amd.distributed import levels_from_ml, save_levels
    np.random.seed(0)
    sm = ("multicolor_gauss_seidel", {"sweep": "symmetric"})
    ml = smoothed_aggregation_solver(poisson(grid), presmoother=sm, postsmoother=sm, max_coarse=30)
    levels, coarse = levels_from_ml(ml)
    save_levels(path, levels, coarse)
    b = np.random.RandomState(4).rand(levels[0]["A"].shape[0])
    np.save(os.path.join(path, "b.npy"), b)
    return levels, coarse, b


def _worker_multicolour(rank, world, port, path, rep):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from pyamg_amd.distributed import DistributedSolver, load_levels, split_rows
        levels, coarse = load_levels(path)
        b = np.load(os.path.join(path, "b.npy"))
        S = DistributedSolver(levels, coarse, OracleBackend(), rank, world, replicate_below=rep)
        bnd = split_rows(len(b), world); lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(b[lo:hi], None, tol=0.0, maxiter=3, cycle="V", fixed=True)
        np.save(os.path.join(path, "x_%d.npy" % rank), x)
    finally:
        dist.destroy_process_group()


def multicolour_emulation(levels, coarse, b, world, rep, cycles=3):
    g = {"levels": levels}
    bounds = hybrid_bounds(g, world, rep)
    xe = np.zeros_like(b)
    for _ in range(cycles):
        _hybrid_cycle(oracle_lib.load(), levels, coarse, bounds, 0, xe, np.ascontiguousarray(b))
    return xe


@pytest.mark.parametrize("world", [2, 3])
def test_hybrid_multicolour_gauss_seidel_matches_partition_emulation(world, tmp_path):
    """C4 as SURVEY 8(e) specifies it: multicolour Gauss-Seidel inside a rank, Jacobi across ranks; oracle = every
    partition relaxing its part of the index list with the reference kernel on a frozen copy"""
    levels, coarse, b = multicolour_hierarchy(str(tmp_path))
    mp.spawn(_worker_multicolour, args=(world, _free_port(), str(tmp_path), 100), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    xe = multicolour_emulation(levels, coarse, b, world, 100)
    assert np.array_equal(x, xe), np.abs(x - xe).max()
    x1 = multicolour_emulation(levels, coarse, b, 1, 100)          # one rank: the plain multicolour sweep -- a different iteration
    assert not np.array_equal(x, x1) and np.linalg.norm(x - x1) < 0.5 * np.linalg.norm(x1)


def _permuted_hierarchy(case, seed):
    """a reference-built hierarchy with every level renumbered by a random permutation (rows reordered, columns
    relabelled, the entries of a row in stored order): contiguous row blocks are then arbitrary subsets of the grid"""
    import scipy.sparse as sps
    g = golden_io.load_hier(case)
    rng = np.random.RandomState(seed)
    L = g["levels"]
    q = [rng.permutation(lv["A"].shape[0]) for lv in L]               # new index k holds old unknown q[l][k]
    q[-1] = np.arange(L[-1]["A"].shape[0])                            # (the dense coarse operator keeps its numbering)
    inv = [np.argsort(p) for p in q]

    def relabel(M, lr, lc):
        M = sps.csr_matrix(M)
        Ap, Aj, Ax = local_rows_of_host(M, q[lr], inv[lc])
        return sps.csr_matrix((Ax, Aj, Ap), shape=M.shape)
    out = []
    for l, lv in enumerate(L):
        d = {"A": relabel(lv["A"], l, l)}
        if "P" in lv:
            d["P"] = relabel(lv["P"], l, l + 1)
            d["R"] = relabel(lv["R"], l + 1, l)
            d["pre"], d["post"] = lv["pre"], lv["post"]
        out.append(d)
    return g, out, q


def local_rows_of_host(M, rows, col_map):
    from pyamg_amd.distributed import local_rows_of
    return local_rows_of(M, rows, col_map)


def _worker_index_sets(rank, world, port, path, case, seed):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from pyamg_amd.distributed import DistributedSolver, owners_by_aggregate, split_rows
        g, levels, q = _permuted_hierarchy(case, seed)
        n = levels[0]["A"].shape[0]
        # ownership of level 0: the block of the ORIGINAL numbering an unknown sits in -- an index set in the permuted one
        bnd = split_rows(n, world)
        owner0 = np.searchsorted(bnd, q[0], side="right") - 1
        owners = owners_by_aggregate(levels, owner0, world)
        S = DistributedSolver(levels, g["coarse_pinv"], OracleBackend(), rank, world, replicate_below=40, owners=owners)
        mine = S.owned(0)
        assert np.array_equal(np.sort(mine), mine) and np.all(owner0[mine] == rank)
        b = np.asarray(g["b"])[q[0]]
        x, res = S.solve(b[mine], None, tol=0.0, maxiter=4, cycle="V", fixed=True)
        np.save(os.path.join(path, "x_%d.npy" % rank), x)
        np.save(os.path.join(path, "i_%d.npy" % rank), mine)
        np.save(os.path.join(path, "halo_%d.npy" % rank), np.array([lv.n_halo for lv in S.lv]))
        # the same ranks with contiguous blocks of the permuted numbering: what ownership by range would exchange
        S2 = DistributedSolver(levels, g["coarse_pinv"], OracleBackend(), rank, world, replicate_below=40)
        np.save(os.path.join(path, "halo_range_%d.npy" % rank), np.array([lv.n_halo for lv in S2.lv]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["sa_jacobi_2d", "sa_cheb2_3d"])
def test_ownership_by_index_set_on_a_permuted_hierarchy(case, tmp_path):
    """a hierarchy in an ARBITRARY numbering (every level randomly permuted), partitioned over 8 ranks by
    index set -- level 0 by the grid block an unknown sits in, coarser levels by `owners_by_aggregate` -- gives the
    single-process iterates bit for bit (a rank's local numbering is its owned indices in ascending order, entries of a
    row keep their stored order), and exchanges far smaller halos than contiguous blocks of that numbering would."""
    world = 8
    seed = 7
    mp.spawn(_worker_index_sets, args=(world, _free_port(), str(tmp_path), case, seed), nprocs=world, join=True)
    g, levels, q = _permuted_hierarchy(case, seed)
    n = levels[0]["A"].shape[0]
    x = np.zeros(n)
    for r in range(world):
        x[np.load(tmp_path / ("i_%d.npy" % r))] = np.load(tmp_path / ("x_%d.npy" % r))
    b = np.asarray(g["b"])[q[0]]
    xs, _ = oracle_lib.Hierarchy(levels, g["coarse_pinv"]).solve(b, tol=0.0, maxiter=4)
    assert np.array_equal(x, xs), np.abs(x - xs).max()
    halo = sum(int(np.load(tmp_path / ("halo_%d.npy" % r))[0]) for r in range(world))
    halo_range = sum(int(np.load(tmp_path / ("halo_range_%d.npy" % r))[0]) for r in range(world))
    assert halo < 0.5 * halo_range, (halo, halo_range)


def test_coarse_bounds_follow_the_prolongator():
    """distributed.coarse_bounds: a coarse cut that follows the fine cut through P never needs more off-rank
    columns of P than the even split does by more than a few rows, stays balanced, and degenerates to the even
    split where the coarse numbering does not follow the fine one"""
    import scipy.sparse as sp
    from pyamg_amd.distributed import coarse_bounds, split_rows
    g = golden_io.load_hier("sa_gs_3d")
    P = g["levels"][0]["P"]
    n, nc = P.shape
    for world in (2, 3, 4):
        fb = split_rows(n, world)
        cb = coarse_bounds(P, fb)
        assert cb[0] == 0 and cb[-1] == nc and np.all(np.diff(cb) >= 0)
        even = split_rows(nc, world)
        assert np.all(np.abs(np.diff(cb) - np.diff(even)) <= 0.2 * np.diff(even) + 1)
    # a prolongator whose columns run AGAINST the rows: the cut falls back to the even split
    Prev = sp.csr_matrix((np.ones(12), (np.arange(12), 5 - np.arange(12) // 2)), shape=(12, 6))
    assert np.array_equal(coarse_bounds(Prev, split_rows(12, 3)), split_rows(6, 3))
    # aggregates in row order: the cut is exactly the aggregate boundary of the first row of each rank
    Pfwd = sp.csr_matrix((np.ones(12), (np.arange(12), np.arange(12) // 2)), shape=(12, 6))
    assert np.array_equal(coarse_bounds(Pfwd, np.array([0, 4, 8, 12])), np.array([0, 2, 4, 6]))
