"""EVE-Base encoder (12 layers, MoE in layers 10-11, hard modality split elsewhere) forward + backward at the
BASELINE configs[1] shape: batch 64 x (197 vision + 40 text) tokens, bf16, one MI355X.

    python tools/bench_encoder.py            # eve_amd FFN paths (HIP kernels)
    python tools/bench_encoder.py --stock    # the reference's formulation of the FFN branches, issued with stock
                                             # PyTorch-ROCm ops on the same GPU (slice / cat, per-expert nn.Linear loop,
                                             # dense one-hot [S,E,C] dispatch / combine einsums)

Attention, LayerNorm-1, layer scale and DropPath are the same stock ops in both arms.  Not the driver's bench
(bench.py measures the MoE layer, BASELINE.json's metric); this is the SURVEY 8f N2 measurement.
"""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import eve_amd
from eve_amd.encoder import MultiwayTransformerEnocderLayer

ap = argparse.ArgumentParser()
ap.add_argument("--stock", action="store_true")
ap.add_argument("--stock-branch", action="store_true", help="eve_amd FFN paths, residual + drop_path(x * gamma) as stock ops")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--layers", type=int, default=12)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def qgelu(x):
    return x * torch.sigmoid(1.702 * x)


def mlp(m, x):
    return F.linear(qgelu(F.linear(x, m.fc1.weight, m.fc1.bias)), m.fc2.weight, m.fc2.bias)


def stock_split(l, h, mode):  # reference multiway_transformer.py:365-370, :385-387, :401-403
    ln = lambda n, x: F.layer_norm(x, (x.shape[-1],), n.weight, n.bias, n.eps)
    if mode == "vision":
        return mlp(l.mlp_imag, ln(l.norm2_imag, h))
    if mode == "text":
        return mlp(l.mlp_text, ln(l.norm2_text, h))
    nt = l.num_text_tokens
    return torch.cat([mlp(l.mlp_imag, ln(l.norm2_imag, h[:, :-nt])), mlp(l.mlp_text, ln(l.norm2_text, h[:, -nt:]))], dim=1)


def stock_moe(l, h, mode):  # reference sharded_moe.py:172-235, :348-381 and experts.py:18-28, dense formulation
    moe = l.mlp_moe.moe
    E = moe.gate.wg.weight.shape[0]
    x = F.layer_norm(h, (h.shape[-1],), l.norm2_vl.weight, l.norm2_vl.bias, l.norm2_vl.eps).clone()
    nv = moe.num_vision_tokens
    x[:, :nv] += moe.vision_embed.to(x.dtype)
    x[:, nv:] += moe.text_embed.to(x.dtype)
    xg = x.reshape(-1, x.shape[-1])
    S = xg.shape[0]
    logits = F.linear(xg.float(), moe.gate.wg.weight.float())
    gates = F.softmax(logits, dim=1)
    C = max(math.ceil(S / E * 2.0), 4)
    i1 = torch.argmax(gates, dim=1)
    mask1 = F.one_hot(i1, E)
    noise = -torch.log(-torch.log(torch.rand_like(logits).clamp_(1e-20, 1 - 1e-7)))
    i2 = torch.argmax((logits + noise).masked_fill(mask1.bool(), float("-inf")), dim=1)
    mask2 = F.one_hot(i2, E)
    loc1 = torch.cumsum(mask1, 0) - 1
    loc2 = torch.cumsum(mask2, 0) - 1 + mask1.sum(0, keepdim=True)
    l_aux = torch.mean(gates.mean(0) * mask1.float().mean(0)) * E * E
    mask1 = mask1 * (loc1 < C)
    mask2 = mask2 * (loc2 < C)
    l1s, l2s = (loc1 * mask1).sum(1), (loc2 * mask2).sum(1)
    g1 = (gates * mask1).sum(1)
    g2 = (gates * mask2).sum(1)
    den = (g1 + g2).clamp(min=torch.finfo(gates.dtype).eps)
    g1, g2 = g1 / den, g2 / den
    comb = (torch.einsum("se,sc->sec", g1[:, None] * mask1, F.one_hot(l1s, C).float())
            + torch.einsum("se,sc->sec", g2[:, None] * mask2, F.one_hot(l2s, C).float()))
    disp = comb.bool()
    d = torch.einsum("sec,sm->ecm", disp.to(xg.dtype), xg)
    y = torch.stack([mlp(ex, d[e]) for e, ex in enumerate(moe.experts.experts)])
    out = torch.einsum("sec,ecm->sm", comb.to(xg.dtype), y).reshape(h.shape)
    l.aux_loss_vl = l.aux_loss_vl + l_aux
    l.aux_loss_vl_i += 1
    return out


class StockLayer(MultiwayTransformerEnocderLayer):
    def _ffn(self, h, mode):
        if self.use_ffn and self.ffn_type == "moe":
            return stock_moe(self, h, mode)
        return stock_split(self, h, mode)

    def _branch(self, residual, x, gamma):  # reference :341, :408 as three stock ops
        return residual + self.drop_path(x * gamma)


class StockBranchLayer(MultiwayTransformerEnocderLayer):  # A/B of the fused residual branch alone
    _branch = StockLayer._branch


enc = eve_amd.eve_base_encoder()
if a.layers != 12:  # shorter stacks for quick runs: keep the last two (MoE) layers
    enc.layers = torch.nn.ModuleList(list(enc.layers)[12 - a.layers:])
enc = enc.to(torch.bfloat16).to(dev).train()
if a.stock or a.stock_branch:
    for l in enc.layers:
        l.__class__ = StockLayer if a.stock else StockBranchLayer
B, T, M = a.batch, 237, 768
g = torch.Generator().manual_seed(1)
x0 = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(dev)
dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(dev)
mask = torch.ones(B, T, dtype=torch.long, device=dev)
params = list(enc.parameters())


def step():
    for p in params:
        p.grad = None
    x = x0.detach().requires_grad_(True)
    out = enc(x, mask, mode="vl")
    gl = enc.gate_loss(0.001)
    torch.autograd.backward([out, gl], [dout, torch.ones_like(gl)])


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
print(f"{'stock-ops reference formulation' if a.stock else 'eve_amd'}: {len(enc.layers)} layers, B={B}, T={T}, bf16: "
      f"{ms:.2f} ms/step fwd+bwd = {B * T / ms * 1e3 / 1e6:.3f} M tokens/s; peak memory "
      f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
