"""Is the bench step launch-bound?  Compares the host time to ENQUEUE K steps with the time until they finish."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

class A: pass
dev = torch.device("cuda", 0)
M, I, E, n_vis, T, B = (1024, 4096, 32, 197, 237, 32) if "--large" in sys.argv else (768, 3072, 32, 197, 237, 64)
moe = bench.build_layer(M, I, E, 1, 0, n_vis, torch.bfloat16, dev)
g = torch.Generator().manual_seed(1234)
h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(torch.bfloat16).to(dev)
dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(dev)
params = list(moe.parameters())
def step():
    for p in params: p.grad = None
    x = h.detach().requires_grad_(True)
    out, l_aux, _ = moe(x, mode="vl")
    torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])
for _ in range(5): step()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.3f} ms/step   complete {1e3*(t2-t0)/K:.3f} ms/step")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K): step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
