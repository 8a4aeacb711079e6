"""Step time of the cfg2 workload with the data-parallel engine active (fake 2-rank process group with
RCCL-like stream semantics, tests/test_parallel_gpu.py:FakeDist) vs the plain single-process step:
shows what the gradient hooks cost / whether the weight-gradient overlap survives under DP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from test_parallel_gpu import FakeDist

dev = torch.device("cuda")
for tag, dist in (("single process", None), ("DP engine (fake 2 ranks)", FakeDist())):
    model, step = bench.build_step(sys.argv[1] if len(sys.argv) > 1 else "cfg2", dev, dist=dist)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("%-28s %.2f ms/step" % (tag, (time.perf_counter() - t0) / n * 1e3))
    del model, step
