"""Run only the grouped-FFN entry points at a given shape (for rocprofv3 passes)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eve_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--G", type=int, default=32)
ap.add_argument("--E", type=int, default=32)
ap.add_argument("--R", type=int, default=948)
ap.add_argument("--M", type=int, default=768)
ap.add_argument("--I", type=int, default=3072)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--fill", type=float, default=0.98)
ap.add_argument("--act", default="quick_gelu")
ap.add_argument("--shared", action="store_true", help="flags = EVE_FFN_SHARED_GPU: the launch policy of the pipelined EP path")
a = ap.parse_args()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
x = torch.randn(a.G, a.R, a.M, generator=g).to(bf).to(dev)
w1 = (torch.randn(a.E, a.I, a.M, generator=g) * 0.02).to(bf).to(dev)
b1 = torch.zeros(a.E, a.I, dtype=bf, device=dev)
w2 = (torch.randn(a.E, a.M, a.I, generator=g) * 0.02).to(bf).to(dev)
b2 = torch.zeros(a.E, a.M, dtype=bf, device=dev)
dy = torch.randn(a.G, a.R, a.M, generator=g).to(bf).to(dev)
rows = torch.full((a.G,), int(a.R * a.fill), dtype=torch.int32, device=dev)
act = ops.ACTS[a.act]
for it in range(a.iters):
    if it == a.iters - 1:
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
    z, aa, y = ops.ffn_fwd(x, w1, b1, w2, b2, rows, act, shared_gpu=a.shared)
    if it == a.iters - 1:
        e1.record()
    out = ops.ffn_bwd(dy, x, z, aa, w1, w2, rows, act, shared_gpu=a.shared)
    if it == a.iters - 1:
        e2.record()
torch.cuda.synchronize()
kept = int(rows.sum())
fl = 2.0 * a.M * a.I * kept
print(f"fwd {e0.elapsed_time(e1):.3f} ms ({2*fl/e0.elapsed_time(e1)/1e9:.0f} TF)  bwd {e1.elapsed_time(e2):.3f} ms ({4*fl/e1.elapsed_time(e2)/1e9:.0f} TF)")
