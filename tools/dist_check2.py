"""2 ranks on one GPU (gloo) vs single-GPU resident solve on an own-setup hierarchy"""
import sys, os, socket, shutil, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port, shared, iters):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyamg_amd.distributed import DistributedSolver, HipBackend, load_levels, split_rows
    levels, coarse = load_levels(shared)
    S = DistributedSolver(levels, coarse, HipBackend(0), rank, world)
    n = levels[0]["A"].shape[0]; bnd = split_rows(n, world); lo, hi = int(bnd[rank]), int(bnd[rank + 1])
    b = np.load(os.path.join(shared, "b.npy"))[lo:hi]
    x, res = S.solve(b, None, tol=0.0, maxiter=iters)
    np.save(os.path.join(shared, "x_%d.npy" % rank), x)
    if rank == 0: np.save(os.path.join(shared, "res.npy"), np.array(res))
    # also dump level-1 coarse b after first restriction for debugging
    dist.destroy_process_group()

if __name__ == "__main__":
    from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
    from pyamg_amd.distributed import levels_from_ml, save_levels
    n = int(sys.argv[1]); world = int(sys.argv[2]) if len(sys.argv) > 2 else 2; iters = 5
    A = poisson((n, n, n)); np.random.seed(0)
    spec = ("chebyshev", {"degree": 2})
    ml = smoothed_aggregation_solver(A, presmoother=spec, postsmoother=spec)
    np.random.seed(0); b = np.random.rand(A.shape[0])
    res = []; x = ml.solve(b, tol=0.0, maxiter=iters, residuals=res)
    levels, coarse = levels_from_ml(ml)
    shared = "/dev/shm/amg_check"; shutil.rmtree(shared, ignore_errors=True)
    save_levels(shared, levels, coarse); np.save(os.path.join(shared, "b.npy"), b)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port, shared, iters), nprocs=world, join=True)
    xd = np.concatenate([np.load(os.path.join(shared, "x_%d.npy" % r)) for r in range(world)])
    rd = np.load(os.path.join(shared, "res.npy"))
    print("levels", [(L["A"].shape[0]) for L in levels])
    print("resident   ", ["%.10e" % r for r in res])
    print("partitioned", ["%.10e" % r for r in rd])
    print("x equal:", np.array_equal(x, xd), np.abs(x - xd).max(), "first diff idx", (np.nonzero(x != xd)[0][:5]))
    shutil.rmtree(shared, ignore_errors=True)
