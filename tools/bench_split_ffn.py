"""Tokens/s of the hard modality-split FFN fwd+bwd at the config-2 layer shape on one MI355X:
eve_amd.ModalitySplitFFN (eager and HIP-graph replay) beside the reference's formulation issued with stock
PyTorch-ROCm ops (slice -> LayerNorm -> Linear/quick-GELU/Linear per modality -> cat) on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eve_amd
from oracle import moe_oracle as orc  # parameters only

dev = torch.device("cuda", 0)
B, T, M, I, NT = 64, 237, 768, 3072, 40
p = {k: v.to(torch.bfloat16).to(dev) for k, v in orc.make_split_params(M, I, seed=1).items()}
m = eve_amd.ModalitySplitFFN("quick_gelu", M, I, num_text_tokens=NT)
with torch.no_grad():
    for i, (norm, mlp) in enumerate(((m.norm2_imag, m.mlp_imag), (m.norm2_text, m.mlp_text))):
        norm.weight.copy_(p["ln_w"][i]); norm.bias.copy_(p["ln_b"][i])
        mlp.fc1.weight.copy_(p["w1"][i]); mlp.fc1.bias.copy_(p["b1"][i])
        mlp.fc2.weight.copy_(p["w2"][i]); mlp.fc2.bias.copy_(p["b2"][i])
m = m.to(torch.bfloat16).to(dev)
g = torch.Generator().manual_seed(0)
h = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(dev)
dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(dev)
leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}


def ours():
    for q in m.parameters():
        q.grad = None
    x = h.detach().requires_grad_(True)
    m(x, mode="vl").backward(dout)


def stock():
    for q in leaves.values():
        q.grad = None
    x = h.detach().requires_grad_(True)
    F = torch.nn.functional

    def branch(t, i):
        t = F.layer_norm(t, (M,), leaves["ln_w"][i], leaves["ln_b"][i], 1e-6)
        t = F.linear(t, leaves["w1"][i], leaves["b1"][i])
        t = t * torch.sigmoid(1.702 * t)
        return F.linear(t, leaves["w2"][i], leaves["b2"][i])

    torch.cat([branch(x[:, :-NT], 0), branch(x[:, -NT:], 1)], dim=1).backward(dout)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


class Wrap(torch.nn.Module):
    def __init__(self, mod):
        super().__init__()
        self.mod = mod

    def forward(self, x, mode="vl"):
        return self.mod(x, mode=mode), x.new_zeros((), dtype=torch.float32, requires_grad=True) + 0.0


flops = 12.0 * M * I * B * T
arms = (("eve_amd eager", ours),) if "--ours-only" in sys.argv else (("eve_amd eager", ours), ("stock torch ops (rocBLAS)", stock))
for name, fn in arms:
    dt = timeit(fn)
    print(f"{name:28s} {dt*1e3:7.3f} ms/step  {B*T/dt/1e6:6.2f} M tokens/s  {flops/dt/1e12:6.0f} TFLOP/s")
if "--ours-only" in sys.argv:
    sys.exit(0)
step = eve_amd.GraphedTrainStep(Wrap(m), h, dout, mode="vl", aux_weight=0.0)
dt = timeit(lambda: step(h, dout))
print(f"{'eve_amd HIP-graph replay':28s} {dt*1e3:7.3f} ms/step  {B*T/dt/1e6:6.2f} M tokens/s  {flops/dt/1e12:6.0f} TFLOP/s")
