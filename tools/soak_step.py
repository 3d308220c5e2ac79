"""Diagnostic: replay the bench step (cfg-2 layer, HIP graph, fresh device noise every step) N times and check, every
CHECK steps, that outputs and gradients are finite, that every token got a first choice, that the kept-assignment
count stays in its usual band, and that the generator offset advanced by exactly one draw per step.
    python tools/soak_step.py [N=20000] [CHECK=1000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, eve_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
M, I, E, n_vis, B, T = 768, 3072, 32, 197, 64, 237
S = B * T
moe = bench.build_layer(M, I, E, 1, 0, n_vis, torch.bfloat16, dev)
h, dout = bench.synth_batch(B, T, M, torch.bfloat16, dev, seed=1234)
step = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
off0 = int(moe.moe.device_noise.state[1])
bad = 0
for i in range(1, N + 1):
    out, l_aux = step()
    if i % CHECK == 0:
        torch.cuda.synchronize()
        cnt = moe.moe.exp_counts
        kept = int((moe.moe.last_slot >= 0).sum())
        fin = bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(l_aux)) and all(
            bool(torch.isfinite(p.grad.float()).all()) for p in moe.parameters() if p.grad is not None) and bool(
            torch.isfinite(step.dx.float()).all())
        off = int(moe.moe.device_noise.state[1])
        ok = fin and int(cnt.sum()) == S and 0.95 * 2 * S < kept <= 2 * S and off == off0 + i * (S * E // 4)
        bad += 0 if ok else 1
        print(f"step {i}: finite={fin} first-choices={int(cnt.sum())} kept={kept} l_aux={float(l_aux):.4f} offset_ok={off == off0 + i * (S * E // 4)}", flush=True)
# determinism under load: the same generator state replayed twice must give the same bits everywhere (a race in
# a kernel would show as a difference); 50 pairs
ndiff = 0
for k in range(50):
    snap = moe.moe.device_noise.snapshot()
    step()
    torch.cuda.synchronize()
    ref = [step.out.clone(), step.dx.clone(), step.l_aux.clone()] + [p.grad.clone() for p in moe.parameters() if p.grad is not None]
    moe.moe.device_noise.restore(snap)
    step()
    torch.cuda.synchronize()
    now = [step.out, step.dx, step.l_aux] + [p.grad for p in moe.parameters() if p.grad is not None]
    ndiff += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, now))
print(f"determinism: {ndiff} tensors differed over 50 replayed pairs")
bad += 1 if ndiff else 0
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad} checks)")
sys.exit(0 if bad == 0 else 1)
