"""conv1 weight gradient (uint8 frames): GPU time of a few (tile, splits) plans, graph-timed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agents_amd import ops
from agents_amd.utils import graph
dev = torch.device("cuda", 0)
S = 256
g = torch.Generator().manual_seed(0)
obs = torch.randint(0, 256, (S, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
dz = torch.randn(S * 400, 32, generator=g).to(dev)
gk = torch.empty(8, 8, 4, 32, device=dev)
gb = torch.empty(32, device=dev)

def timeit(fn, reps=20):
    fn()
    c = graph._Captured()
    c.capture(lambda: [fn() for _ in range(reps)] and None)
    c.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); c.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for cfg, splits in [(0, 0), (8, 128), (8, 192), (8, 256), (8, 384), (8, 512), (8, 768), (7, 64), (2, 128), (2, 256)]:
    try:
        t = timeit(lambda: ops.conv_dw(obs, dz, (8, 8, 4, 32), 4, gk, a_div=255.0, force_cfg=cfg,
                                       force_splits=splits, bias_grad=gb))
        print(f"cfg {cfg} splits {splits:4d}: {t:7.1f} us  ({2*256*32*102400/t/1e6:6.1f} TFLOP/s)")
    except Exception as e:
        print(f"cfg {cfg} splits {splits}: {e}")
