"""Per-kernel timing of the gate / permute kernels at a bench shape (HIP events around each C-ABI call).
   python tools/side_bench.py [--large]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eve_amd import _lib, ops

dev = torch.device("cuda", 0)
large = "--large" in sys.argv
M, I, E, n_vis, T, B = (1024, 4096, 32, 197, 237, 32) if large else (768, 3072, 32, 197, 237, 64)
S = B * T
bf = dict(dtype=torch.bfloat16, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
h = torch.randn(S, M, generator=g, **bf)
ve, te = torch.randn(M, generator=g, **bf) * 0.02, torch.randn(M, generator=g, **bf) * 0.02
wg = torch.randn(E, M, generator=g, device=dev) * 0.02
noise = torch.rand(S, E, generator=g, device=dev)
C = ops.capacity(S, E, 1.0, 4)
mode = _lib.MODE_VL


def timeit(name, f, alg_bytes, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"{name:22s} {us:7.1f} us   {alg_bytes / us / 1e3:7.0f} GB/s algorithmic = {alg_bytes / us / 1e3 / 80:5.1f}% of 8 TB/s", flush=True)


logits = ops.gate_logits(h, ve, te, wg, T, n_vis, mode)
r = ops.gate_top2(logits, noise, C, True)
K = int((r.slot >= 0).sum())
print(f"S={S} M={M} E={E} C={C} kept={K}")
b = 2
xd = ops.dispatch_fwd(h, ve, te, r.slot_token, T, n_vis, mode)
y = torch.randn(E * C, M, generator=g, **bf)
dout = torch.randn(S, M, generator=g, **bf)
dy, dw = ops.combine_bwd(dout, y, r.slot, r.slot_token, r.slot_weight)
dl = ops.gate_top2_bwd(r, dw, torch.ones(1, device=dev))
timeit("gate_logits", lambda: ops.gate_logits(h, ve, te, wg, T, n_vis, mode), S * M * b + 4 * S * E)
timeit("gate_top2", lambda: ops.gate_top2(logits, noise, C, True), 12 * E * S + 40 * S)
timeit("dispatch_fwd", lambda: ops.dispatch_fwd(h, ve, te, r.slot_token, T, n_vis, mode, slot=r.slot), (S + K) * M * b)
timeit("dispatch_fwd slotmajor", lambda: ops.dispatch_fwd(h, ve, te, r.slot_token, T, n_vis, mode), (S + K) * M * b)
timeit("combine_fwd", lambda: ops.combine_fwd(y, r.slot, r.weight), (S + K) * M * b)
timeit("combine_bwd", lambda: ops.combine_bwd(dout, y, r.slot, r.slot_token, r.slot_weight), (S + 2 * K) * M * b)
timeit("gate_top2_bwd", lambda: ops.gate_top2_bwd(r, dw, torch.ones(1, device=dev)), 12 * E * S)
timeit("route_bwd", lambda: ops.route_bwd(dy, r.slot, dl, wg, S, T, n_vis, mode, True, True, torch.bfloat16), (S + K) * M * b)
timeit("gate_wgrad", lambda: ops.gate_wgrad(h, ve, te, dl, T, n_vis, mode), S * M * b + 4 * S * E)
if "--pair" in sys.argv:  # route_bwd and gate_wgrad (independent of each other) on two streams: what running them side by side buys
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ops.route_bwd(dy, r.slot, dl, wg, S, T, n_vis, mode, True, True, torch.bfloat16)
        with torch.cuda.stream(s2):
            ops.gate_wgrad(h, ve, te, dl, T, n_vis, mode)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def serial():
        ops.route_bwd(dy, r.slot, dl, wg, S, T, n_vis, mode, True, True, torch.bfloat16)
        ops.gate_wgrad(h, ve, te, dl, T, n_vis, mode)

    timeit("route_bwd ; gate_wgrad", serial, (2 * S + K) * M * b)
    timeit("route_bwd || gate_wgrad", both, (2 * S + K) * M * b)
