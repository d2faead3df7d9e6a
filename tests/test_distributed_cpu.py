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


def _worker(rank, world, port, case, cycle, out_dir):
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
        S = DistributedSolver(levels, coarse, OracleBackend(), rank, world)
        n = g["levels"][0]["A"].shape[0]
        bnd = split_rows(n, world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x0 = g["x0"][lo:hi] if np.any(g["x0"]) else None
        x, res = S.solve(g["b"][lo:hi], x0, tol=g["meta"]["tol"], maxiter=g["meta"]["maxiter"], cycle=cycle)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array(res))
            np.save(os.path.join(out_dir, "halo.npy"), np.array([lv.n_halo for lv in S.lv]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("sa_jacobi_2d", 2), ("sa_cheb2_3d", 2), ("sa_cheb2_3d", 3),
                                        ("sa_mixed_W_2d", None)])
def test_partitioned_cycle_equals_single_process(case, world, tmp_path):
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
    mp.spawn(_worker, args=(world, port, case, m["cycle"], str(tmp_path)), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    res = np.load(tmp_path / "res.npy")
    H = oracle_lib.Hierarchy(g["levels"], g["coarse_pinv"])
    xo, reso = H.solve(g["b"], x0=g["x0"], tol=m["tol"], maxiter=m["maxiter"], cycle=m["cycle"])
    assert len(res) == len(reso) == len(g["residuals"])
    assert np.array_equal(x, xo), np.abs(x - xo).max()
    tol = golden_io.history_tolerance(g["levels"][0]["A"], g["x"], g["b"], g["residuals"])
    assert np.all(np.abs(res - g["residuals"]) <= tol)
    assert np.load(tmp_path / "halo.npy")[0] > 0          # the ranks really exchanged halos


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
