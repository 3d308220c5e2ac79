"""A/B timing of the grouped-GEMM kernel variants (EVE_FFN_KERNEL / EVE_FFN_PERSIST / EVE_FFN_TILE_ROWS knobs), one child
process per variant on the same device (the library reads the knobs once per process)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eve_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--G", type=int, default=32)
ap.add_argument("--E", type=int, default=32)
ap.add_argument("--R", type=int, default=948)
ap.add_argument("--M", type=int, default=768)
ap.add_argument("--I", type=int, default=3072)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--fill", type=float, default=1.0)
ap.add_argument("--variants", default="lockstep,pingpong")
a = ap.parse_args()
if "," in a.variants:  # the library reads its debug knobs once per process: one child process per variant
    import subprocess
    for v in a.variants.split(","):
        rc = subprocess.run([sys.executable, os.path.abspath(__file__)] + [x for x in sys.argv[1:] if not x.startswith("--variants")
                                                                         and x != a.variants] + ["--variants", v]).returncode
        if rc:
            sys.exit(rc)
    sys.exit(0)
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
x = torch.randn(a.G, a.R, a.M, generator=g).to(bf).to(dev)
w1 = (torch.randn(a.E, a.I, a.M, generator=g) * 0.02).to(bf).to(dev)
b1 = (torch.randn(a.E, a.I, generator=g) * 0.02).to(bf).to(dev)
w2 = (torch.randn(a.E, a.M, a.I, generator=g) * 0.02).to(bf).to(dev)
b2 = (torch.randn(a.E, a.M, generator=g) * 0.02).to(bf).to(dev)
dy = torch.randn(a.G, a.R, a.M, generator=g).to(bf).to(dev)
rows = torch.full((a.G,), int(a.R * a.fill), dtype=torch.int32, device=dev)
if a.fill < 1.0:  # ragged row counts
    rows = (torch.rand(a.G, generator=g) * a.R * a.fill).to(torch.int32).to(dev)
act = ops.ACTS["quick_gelu"]
kept = int(rows.sum())
fl = 2.0 * a.M * a.I * kept


def run(variant, iters):
    os.environ["EVE_FFN_KERNEL"] = "lockstep" if variant == "lockstep" else "pingpong"
    os.environ.pop("EVE_FFN_TILE_ROWS", None)
    os.environ["EVE_FFN_PERSIST"] = "0" if variant == "onetile" else "1"  # onetile: one tile per workgroup
    if variant in ("pp256", "pp192", "pp128"):  # force the block-tile height (128: the 4-wave lock-step kernel)
        os.environ["EVE_FFN_TILE_ROWS"] = variant[2:]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(iters):
        e[0].record()
        z, aa, y = ops.ffn_fwd(x, w1, b1, w2, b2, rows, act)
        e[1].record()
        out = ops.ffn_bwd(dy, x, z, aa, w1, w2, rows, act)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / iters, tb / iters, (z, aa, y) + tuple(out)


variants = a.variants.split(",")
ref = None
for v in variants:
    _, _, outs = run(v, 2)
    if ref is None:
        ref = outs
    else:
        for i, (p, q) in enumerate(zip(ref, outs)):
            if p is None:
                continue
            r = int(rows.min())
            pp, qq = (p, q) if p.shape[0] != a.G or p.dim() < 3 else (p[:, :r], q[:, :r])
            same = torch.equal(pp, qq)
            err = (pp.float() - qq.float()).abs().max().item()
            print(f"  output {i}: {'bit-identical' if same else f'DIFFERENT max abs err {err:.3e}'}")
for rd in range(a.rounds):
    for v in variants:
        tf, tb, _ = run(v, a.iters)
        print(f"round {rd} {v:9s} fwd {tf:.3f} ms ({2*fl/tf/1e9:.0f} TF)  bwd {tb:.3f} ms ({4*fl/tb/1e9:.0f} TF)", flush=True)
