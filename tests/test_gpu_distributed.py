"""GPU: the row-partitioned cycle on real HIP kernels.  Two or three ranks share the one GPU of the box and the
gathered iterates must equal the single-GPU resident solve bit for bit, for every transport:
  peer    the C++ cycle with IPC-mapped arenas: GPU-to-GPU pushes + flag kernels on the hierarchy's stream
  rccl    the C++ cycle with grouped ncclSend/ncclRecv + ncclAllReduce (RCCL refuses two ranks on one device, so
          each rank poses as its own host -- NCCL_HOSTID -- and RCCL takes its socket transport over loopback:
          the C++/RCCL call path is the one an 8-GPU node runs, only the wire differs)
  python  the cycle driven from pyamg_amd/distributed.py with torch.distributed collectives (gloo here)"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_io

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _transport_env(rank, transport):
    os.environ["AMG_DIST_TRANSPORT"] = transport
    if transport == "rccl":
        os.environ.update(NCCL_HOSTID="amgtest-rank%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                          NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1", NCCL_DEBUG="ERROR")


def _worker(rank, world, port, case, out_dir, transport):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _transport_env(rank, transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = None
    try:
        from pyamg_amd.distributed import DistributedSolver, HipBackend, split_rows
        g = golden_io.load_hier(case)
        S = DistributedSolver(g["levels"], g["coarse_pinv"], HipBackend(0), rank, world,
                              replicate_below=(0 if case == "sa_jacobi_2d" else 600))
        n = g["levels"][0]["A"].shape[0]
        bnd = split_rows(n, world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(g["b"][lo:hi], None, tol=g["meta"]["tol"], maxiter=g["meta"]["maxiter"], cycle="V")
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        # fixed-count variant (what bench.py times) from the same start
        x2, res2 = S.solve(g["b"][lo:hi], None, tol=0.0, maxiter=5, cycle="V", fixed=True)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array(res))
            np.save(os.path.join(out_dir, "res_fixed.npy"), np.array(res2))
            np.save(os.path.join(out_dir, "native.npy"), np.array([1.0 if S.native is not None else 0.0]))
    finally:
        if S is not None:
            S.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["peer", "rccl", "python"])
@pytest.mark.parametrize("case", ["sa_jacobi_2d", "sa_cheb2_3d"])
def test_two_ranks_equal_single_gpu(case, transport, tmp_path):
    g = golden_io.load_hier(case)
    ml = golden_io.build_ml(g)
    res1 = []
    x1 = ml.solve(g["b"], tol=g["meta"]["tol"], maxiter=g["meta"]["maxiter"], residuals=res1)
    mp.spawn(_worker, args=(2, _free_port(), case, str(tmp_path), transport), nprocs=2, join=True)
    assert np.load(tmp_path / "native.npy")[0] == (0.0 if transport == "python" else 1.0)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(2)])
    res = np.load(tmp_path / "res.npy")
    assert len(res) == len(res1)
    assert np.array_equal(x, x1), np.abs(x - x1).max()
    assert np.allclose(res, res1, rtol=1e-12, atol=1e-13 * res1[0])
    rf = np.load(tmp_path / "res_fixed.npy")
    assert len(rf) == 6 and np.allclose(rf, res1[:6], rtol=1e-12, atol=1e-13 * res1[0])


def _own_hierarchy(grid):
    from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
    A = poisson(grid)
    np.random.seed(0)
    sm = ("chebyshev", {"degree": 2})
    ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
    b = np.random.RandomState(7).rand(A.shape[0])
    return ml, b


def _worker_stencil(rank, world, port, grid, out_dir, transport):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _transport_env(rank, transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = None
    try:
        from pyamg_amd.distributed import DistributedSolver, HipBackend, split_rows, levels_from_ml
        ml, b = _own_hierarchy(grid)
        levels, coarse = levels_from_ml(ml)
        S = DistributedSolver(levels, coarse, HipBackend(0), rank, world, replicate_below=600)
        assert S.operator_form(0) == 2, "level 0 of the slab should be in stencil form"
        bnd = split_rows(len(b), world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(b[lo:hi], None, tol=0.0, maxiter=5, cycle="V", fixed=True)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array(res))
    finally:
        if S is not None:
            S.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(2, "peer"), (3, "peer"), (3, "rccl"), (2, "python")])
def test_rank_local_stencil_form_with_halos(world, transport, tmp_path):
    """Level 0 of a rank's slab is large enough (>= 1024 rows) for the stencil form: its halo columns
    enter the union stencil as extra offsets, the lower-halo one FIRST in stored order although its
    local index is the largest (slot order is topological, not increasing).  3 ranks: the middle one
    has both halos; the slab boundaries cut through grid planes.  Also covers the residual of the
    convergence test being handed to the next pre-smoother in the partitioned driver."""
    grid = (25, 24, 23)
    ml, b = _own_hierarchy(grid)
    res1 = []
    x1 = ml.solve(b, tol=0.0, maxiter=5, residuals=res1)
    mp.spawn(_worker_stencil, args=(world, _free_port(), grid, str(tmp_path), transport), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    res = np.load(tmp_path / "res.npy")
    assert np.array_equal(x, x1), np.abs(x - x1).max()
    assert np.allclose(res, res1, rtol=1e-12, atol=1e-13 * res1[0])


def _worker_sliced(rank, world, port, grid, out_dir, transport):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _transport_env(rank, transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = None
    try:
        from pyamg_amd.distributed import DistributedSolver, HipBackend, split_rows, levels_from_ml
        ml, b = _own_hierarchy(grid)
        levels, coarse = levels_from_ml(ml)
        S = DistributedSolver(levels, coarse, HipBackend(0), rank, world, replicate_below=600)
        assert S.operator_form(1) == 3, "A_1 of the slab should have its interior rows in the sliced form"
        bnd = split_rows(len(b), world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(b[lo:hi], None, tol=0.0, maxiter=4, cycle="V", fixed=True)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
        if rank == 0:
            np.save(os.path.join(out_dir, "res.npy"), np.array(res))
    finally:
        if S is not None:
            S.close()
        dist.destroy_process_group()


def test_partitioned_level_applies_interior_rows_from_the_sliced_form(tmp_path):
    """A partitioned level whose interior rows (the ones that run beside the halo exchange) number 2^16 and more gets
    the sliced form for exactly that row range; boundary rows, restriction and prolongation as before.  Two ranks on
    the GPU: gathered iterate bit-identical to the single-GPU solve."""
    grid = (128, 126, 122)
    world = 2
    ml, b = _own_hierarchy(grid)
    res1 = []
    x1 = ml.solve(b, tol=0.0, maxiter=4, residuals=res1)
    mp.spawn(_worker_sliced, args=(world, _free_port(), grid, str(tmp_path), "peer"), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    res = np.load(tmp_path / "res.npy")
    assert np.array_equal(x, x1), np.abs(x - x1).max()
    assert np.allclose(res, res1, rtol=1e-12, atol=1e-13 * res1[0])


def _worker_hybrid(rank, world, port, case, out_dir, rep=0, transport="peer"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _transport_env(rank, transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = None
    try:
        from pyamg_amd.distributed import DistributedSolver, HipBackend, split_rows
        g = golden_io.load_hier(case)
        S = DistributedSolver(g["levels"], g["coarse_pinv"], HipBackend(0), rank, world, replicate_below=rep)
        n = g["levels"][0]["A"].shape[0]
        bnd = split_rows(n, world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(g["b"][lo:hi], None, tol=0.0, maxiter=3, cycle="V", fixed=True)
        np.save(os.path.join(out_dir, "x_%d.npy" % rank), x)
    finally:
        if S is not None:
            S.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("case,rep,transport", [("sa_gs_3d", 0, "peer"), ("rs_gs_2d", 450, "peer"), ("rs_gs_2d", 450, "python"),
                                                ("sa_gs_3d", 0, "peer-flow")])
def test_hybrid_gauss_seidel_on_gpu_matches_partition_emulation(case, rep, transport, tmp_path, monkeypatch):
    """C4's smoother: GS inside a rank (level-scheduled HIP kernels), Jacobi across ranks; oracle =
    the partition-emulating CPU run (tests/test_distributed_cpu.py)."""
    import oracle_lib
    from pyamg_amd.distributed import split_rows
    from test_distributed_cpu import _hybrid_cycle, hybrid_bounds
    world = 2
    if transport == "peer-flow":
        # the in-rank sweeps as dataflow launches with the halo as frozen operands: what a node with one rank per GPU runs
        # (forced here although the two ranks share the device: the grids are a few dozen waves each)
        monkeypatch.setenv("AMG_DIST_FLOW", "1")
        monkeypatch.setenv("AMG_GS_FLOW", "2")
        transport = "peer"
    g = golden_io.load_hier(case)
    mp.spawn(_worker_hybrid, args=(world, _free_port(), case, str(tmp_path), rep, transport), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    lib = oracle_lib.load()
    bounds = hybrid_bounds(g, world, rep)
    xe = np.zeros_like(g["b"])
    for _ in range(3):
        _hybrid_cycle(lib, g["levels"], g["coarse_pinv"], bounds, 0, xe, np.ascontiguousarray(g["b"]))
    assert np.array_equal(x, xe), np.abs(x - xe).max()


def _worker_one_way(rank, world, port, out_dir):
    """a channel on which data travels in ONE direction only (rank 0 -> rank 1 -> rank 2, nothing back), driven for many
    back-to-back exchanges: the double-buffered staging slots must not be overwritten before they were unpacked"""
    import ctypes as C
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyamg_amd import _lib
    L = _lib.lib()
    comm = None
    try:
        torch.cuda.set_device(0)
        N = 40000
        comm = L.amg_comm_create(rank, world, 0, 0)
        counts = np.zeros((world, world), dtype=np.intc)           # counts[dst][src]
        for src in range(world - 1):
            counts[src + 1, src] = N
        ch = L.amg_comm_add_channel(comm, _lib.ip(counts))
        assert ch >= 0
        handle = np.zeros(64, dtype=np.uint8)
        _lib.check(L.amg_comm_commit(comm, handle.ctypes.data))
        out = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(out, torch.from_numpy(handle.copy()))
        handles = np.ascontiguousarray(np.concatenate([t.numpy() for t in out]), dtype=np.uint8)
        _lib.check(L.amg_comm_connect(comm, handles.ctypes.data))
        dist.barrier()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        v = torch.zeros(N, dtype=torch.float64, device="cuda")
        halo = torch.zeros(max(N, 1), dtype=torch.float64, device="cuda")
        bad = torch.zeros(1, dtype=torch.int64, device="cuda")
        base = torch.arange(N, dtype=torch.float64, device="cuda")
        for k in range(300):
            v.copy_(base + (1000.0 * rank + k))
            if rank == 0 and k % 50 == 0:
                torch.cuda._sleep(2000000)                           # the producer falls behind now and then ...
            if rank == world - 1 and k % 37 == 0:
                torch.cuda._sleep(3000000)                           # ... and so does the last consumer
            _lib.check(L.amg_comm_exchange(comm, ch, C.c_void_p(v.data_ptr()), None, C.c_void_p(halo.data_ptr()), st))
            if rank > 0:
                bad += (halo != base + (1000.0 * (rank - 1) + k)).sum()
        torch.cuda.synchronize()
        _lib.check(L.amg_comm_check(comm))
        np.save(os.path.join(out_dir, "bad_%d.npy" % rank), bad.cpu().numpy())
        dist.barrier()
    finally:
        if comm:
            L.amg_comm_destroy(comm)
        dist.destroy_process_group()


@pytest.mark.parametrize("fused", ["0", "1"])
def test_one_directional_channel_keeps_its_staging_slots(fused, tmp_path, monkeypatch):
    """a rank that only RECEIVES on a channel still acknowledges every exchange (flags travel
    between partners in either direction), so a producer cannot run two exchanges ahead and overwrite a staging slot
    that is still being unpacked.  300 back-to-back one-way exchanges over a chain of 3 ranks with stalls injected on both
    ends; every received entry checked."""
    world = 3
    monkeypatch.setenv("AMG_COMM_FUSED", fused)        # four launches per hand-off (default) / two (push + signal, wait + unpack)
    mp.spawn(_worker_one_way, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert int(np.load(tmp_path / ("bad_%d.npy" % r))[0]) == 0, r


def _worker_index_sets_gpu(rank, world, port, path, case, seed):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _transport_env(rank, "peer")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = None
    try:
        from pyamg_amd.distributed import DistributedSolver, HipBackend, owners_by_aggregate, split_rows
        from test_distributed_cpu import _permuted_hierarchy
        g, levels, q = _permuted_hierarchy(case, seed)
        n = levels[0]["A"].shape[0]
        owner0 = np.searchsorted(split_rows(n, world), q[0], side="right") - 1
        owners = owners_by_aggregate(levels, owner0, world)
        S = DistributedSolver(levels, g["coarse_pinv"], HipBackend(0), rank, world, replicate_below=40, owners=owners)
        assert S.native is not None
        mine = S.owned(0)
        b = np.asarray(g["b"])[q[0]]
        x, res = S.solve(b[mine], None, tol=0.0, maxiter=4, cycle="V", fixed=True)
        np.save(os.path.join(path, "x_%d.npy" % rank), x)
        np.save(os.path.join(path, "i_%d.npy" % rank), mine)
    finally:
        if S is not None:
            S.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["sa_jacobi_2d", "sa_cheb2_3d"])
def test_ownership_by_index_set_on_gpu(case, tmp_path):
    """a randomly renumbered hierarchy partitioned by INDEX SET over 3 ranks sharing the device (C++ engine, peer
    transport): gathered iterates bit-identical to the single-process oracle solve of the renumbered hierarchy"""
    import oracle_lib
    from test_distributed_cpu import _permuted_hierarchy
    world, seed = 3, 11
    mp.spawn(_worker_index_sets_gpu, args=(world, _free_port(), str(tmp_path), case, seed), nprocs=world, join=True)
    g, levels, q = _permuted_hierarchy(case, seed)
    x = np.zeros(levels[0]["A"].shape[0])
    for r in range(world):
        x[np.load(tmp_path / ("i_%d.npy" % r))] = np.load(tmp_path / ("x_%d.npy" % r))
    b = np.asarray(g["b"])[q[0]]
    xs, _ = oracle_lib.Hierarchy(levels, g["coarse_pinv"]).solve(b, tol=0.0, maxiter=4)
    assert np.array_equal(x, xs), np.abs(x - xs).max()


def _worker_multicolour_gpu(rank, world, port, path, rep, transport):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _transport_env(rank, transport)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = None
    try:
        from pyamg_amd.distributed import DistributedSolver, HipBackend, load_levels, split_rows
        levels, coarse = load_levels(path)
        b = np.load(os.path.join(path, "b.npy"))
        S = DistributedSolver(levels, coarse, HipBackend(0), rank, world, replicate_below=rep)
        bnd = split_rows(len(b), world)
        lo, hi = int(bnd[rank]), int(bnd[rank + 1])
        x, res = S.solve(b[lo:hi], None, tol=0.0, maxiter=3, cycle="V", fixed=True)
        np.save(os.path.join(path, "x_%d.npy" % rank), x)
    finally:
        if S is not None:
            S.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(2, "peer"), (3, "peer"), (2, "python"), (3, "peer-fused")])
def test_hybrid_multicolour_gauss_seidel_on_gpu_matches_partition_emulation(world, transport, tmp_path, monkeypatch):
    """BASELINE configuration C4 as SURVEY 8(e) specifies it -- multicolour Gauss-Seidel (gauss_seidel_indexed,
    relaxation.h:395-430) inside a rank, Jacobi across ranks -- with 2 and 3 ranks sharing the device: gathered
    iterates bit-identical to the partition-emulating oracle (every partition relaxes its part of the index list with
    the reference kernel on a frozen copy)."""
    from test_distributed_cpu import multicolour_emulation, multicolour_hierarchy
    if transport == "peer-fused":                     # the two-launch hand-off (opt-in)
        monkeypatch.setenv("AMG_COMM_FUSED", "1")
        transport = "peer"
    levels, coarse, b = multicolour_hierarchy(str(tmp_path), grid=(18, 16, 15))
    mp.spawn(_worker_multicolour_gpu, args=(world, _free_port(), str(tmp_path), 100, transport), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x_%d.npy" % r)) for r in range(world)])
    xe = multicolour_emulation(levels, coarse, b, world, 100)
    assert np.array_equal(x, xe), np.abs(x - xe).max()


def test_bench_config_c4_runs_partitioned_hybrid_gauss_seidel(tmp_path):
    """`python bench.py --config C4 --gpus 2`: the hybrid (multicolour) Gauss-Seidel configuration row-partitioned over two
    ranks (sharing the one GPU here), and `--smoother hybrid_gs_lex` on one rank"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AMG_DIST_TRANSPORT")}
    for extra in (["--gpus", "2"], ["--smoother", "hybrid_gs_lex"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "C4", "--grid", "40", "--steps", "3",
                            "--warmup", "2"] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["steps"] == 3 and line["roofline"]["frac"] > 0
        if extra[0] == "--gpus":
            assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "hybrid_gs" in line["config"]["workload"]
        else:
            assert line["cpu_baseline"]["value"] > 0
            assert line["cpu_baseline"]["first_step_iterate_bit_identical_to_gpu"] is True
            assert "cpu_model" in line["cpu_baseline"]


def test_bench_gpus_2_runs_two_ranks(tmp_path):
    """`python bench.py --gpus 2` starts two rank processes itself (here they share the one GPU) and the line it
    prints is the partitioned run's: n_gpus 2, strong scaling, the C++ engine's transport, roofline and cpu_baseline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AMG_DIST_TRANSPORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--grid", "48", "--steps", "3",
                        "--warmup", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3
    assert line["config"]["transport"] == "peer", line["config"]["parallelism"]
    assert line["roofline"]["frac"] > 0 and "traffic" in line["roofline"]
    assert line["cpu_baseline"] is None            # the CPU oracle is timed at N = 1 only
    r = line["config"]["residuals"]            # (Chebyshev(2) cycles raise the residual before it decays, as the reference's do)
    assert all(np.isfinite(v) and v > 0 for v in r)
