"""Is the GEMM main loop itself efficient?  Large dense problems, every tile config, both loops."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agents_amd import ops
from agents_amd.utils import graph
dev = torch.device("cuda", 0)

def timeit(fn, reps=5):
    fn()
    c = graph._Captured()
    c.capture(lambda: [fn() for _ in range(reps)] and None)
    c.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); c.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for (M, N, K) in [(4096, 4096, 4096), (20736, 64, 512), (16384, 64, 4096)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev); out = torch.empty(M, N, device=dev)
    for nodma in (False, True):
        ops.FORCE_NO_DMA = nodma
        for cfg in (1, 3, 4, 6):
            try:
                t = timeit(lambda: ops.dense_forward(x, w, None, None, out, force_cfg=cfg, force_splits=1))
                print(f"M{M} N{N} K{K} dma={not nodma} cfg{cfg}: {t:9.1f} us {2.0*M*N*K/t/1e6:7.1f} TFLOP/s")
            except Exception as e:
                print("fail", cfg, e)
ops.FORCE_NO_DMA = False
