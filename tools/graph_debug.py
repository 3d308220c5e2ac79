import subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
 "eager_then_clear_counts": "M,I,E,B,T=256,512,8,4,37; prior=True; clear='counts'",
 "eager_then_clear_grads": "M,I,E,B,T=256,512,8,4,37; prior=True; clear='grads'",
 "eager_then_clear_x": "M,I,E,B,T=256,512,8,4,37; prior=True; clear='x'",
}
OLD = {
 "small_noeager": "M,I,E,B,T=256,512,8,4,37; prior=False",
 "small_eager":   "M,I,E,B,T=256,512,8,4,37; prior=True",
 "mid_noeager":   "M,I,E,B,T=256,512,8,16,237; prior=False",
 "E32_small":     "M,I,E,B,T=256,512,32,4,37; prior=False",
 "big_M768":      "M,I,E,B,T=768,3072,32,8,237; prior=False",
}
BODY = '''
import sys; sys.path.insert(0, %r)
import torch, eve_amd
from oracle import moe_oracle as orc
from tests.helpers import build_moe
%s
p = orc.make_params(M, I, E, seed=3)
moe = build_moe(p, M, I, E, 5, device="cuda:0").to(torch.bfloat16)
g = torch.Generator().manual_seed(5)
h = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to("cuda:0")
dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to("cuda:0")
if prior:
    x = h.clone().requires_grad_(True)
    out, l_aux, _ = moe(x, mode="vl")
    torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])
import gc
clear = globals().get("clear")
if clear == "counts":
    moe.moe.exp_counts = None; del _
elif clear == "grads":
    for q in moe.parameters(): q.grad = None
elif clear == "x":
    del x, out, l_aux
gc.collect(); torch.cuda.synchronize()
step = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
step(); torch.cuda.synchronize(); print("OK")
'''
for name, cfg in CASES.items():
    r = subprocess.run([sys.executable, "-c", BODY % (REPO, cfg)], capture_output=True, text=True, timeout=120)
    print(name, "rc", r.returncode, r.stdout.strip()[-20:], flush=True)
