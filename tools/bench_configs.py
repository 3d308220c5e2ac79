"""MoE-layer throughput at the other BASELINE.json layer shapes on ONE MI355X (bench.py measures configs[1]):

  cfg3  EVE-Large, 224 px: B=32/GPU x 237 tokens, M=1024, I=4096            fwd+bwd
  cfg4  EVE-Large, 480 px (VQA shape): B=16/GPU x (901 + 40) tokens           fwd+bwd
  cfg5  EVE-Large, 384 px rerank inference: B=128 x (577 + 40) tokens, eval   forward only, no grad
        (S = 78 976 tokens per forward; the reference's dense one-hot [S,E,C] would need ~50 GB)

Single-GPU numbers with all 32 experts local (the multi-GPU runs shard the experts; the driver launches those).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

CFG = {"cfg2": (768, 3072, 64, 197, True), "cfg3": (1024, 4096, 32, 197, True), "cfg4": (1024, 4096, 16, 901, True),
       "cfg5": (1024, 4096, 128, 577, False)}
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="cfg3,cfg4,cfg5")
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
for name in a.configs.split(","):
    M, I, B, n_vis, train = CFG[name]
    T = n_vis + 40
    moe = bench.build_layer(M, I, 32, 1, 0, n_vis, torch.bfloat16, dev)
    g = torch.Generator().manual_seed(1)
    h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(torch.bfloat16).to(dev)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(dev)
    params = list(moe.parameters())
    if train:
        moe.train()
        def step():
            for p in params:
                p.grad = None
            x = h.detach().requires_grad_(True)
            out, l_aux, _ = moe(x, mode="vl")
            torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])
    else:
        moe.eval()
        def step():
            with torch.no_grad():
                moe(h, mode="vl")
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    kept = int((moe.moe.last_slot >= 0).sum())
    flops = (12 if train else 4) * M * I * kept
    print(f"{name}: S={B * T} tokens (B={B}, T={T}), M={M}, I={I}, {'fwd+bwd' if train else 'forward, eval, no grad'}: "
          f"{ms:.3f} ms/step = {B * T / ms * 1e3 / 1e6:.2f} M tokens/s; expert GEMMs {flops / ms / 1e9:.0f} TFLOP/s of the whole step "
          f"({kept} kept assignments)", flush=True)
    del moe, h, dout, params
    torch.cuda.empty_cache()
