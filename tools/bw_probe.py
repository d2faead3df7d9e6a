#!/usr/bin/env python3
"""What streaming kernels reach on this GPU (the practical ceiling the SpMV kernels are compared with):
read-only, copy and 2-read-1-write passes over 4 GB fp64 vectors, timed with CUDA events."""
import torch, sys
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
dev = "cuda:0"
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.rand(n, dtype=torch.float64, device=dev)
z = torch.empty_like(x)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for name, fn, nbytes in [("read   (sum)", lambda: x.sum(), 8 * n),
                         ("copy   (z=x)", lambda: z.copy_(x), 16 * n),
                         ("triad  (z=x+y)", lambda: torch.add(x, y, out=z), 24 * n),
                         ("scale  (z=2x)", lambda: torch.mul(x, 2.0, out=z), 16 * n),
                         ("dot    (x.y)", lambda: torch.dot(x, y), 16 * n)]:
    ms = timeit(fn)
    print("%-16s %8.3f ms  %7.1f GB/s" % (name, ms, nbytes / ms / 1e6), flush=True)
