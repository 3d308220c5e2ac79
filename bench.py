"""bench.py -- MoE-layer forward+backward tokens/s on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without WORLD_SIZE in the environment: starts N worker processes
        itself (one per GPU, RCCL, rendezvous on 127.0.0.1) and prints rank 0's line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W        (the same workers under torchrun)

A step is one forward + backward of ONE EVE-Base modality-aware MoE layer over one synthetic
image+text token batch (BASELINE.json configs[1] layer shape: B=64 per GPU, 197 vision + 40
text tokens, M=768, I=3072, 32 experts, top-2, bf16), inputs resident in HBM.  With N > 1 the
32 experts are sharded over the N ranks (expert parallel, RCCL all-to-all) and every rank
brings its own batch (weak scaling).  Rank 0 prints ONE JSON line.

On one GPU the line also carries `other_configs` (the other BASELINE.json layer shapes and the full 12-layer
EVE-Base encoder step, a few timed steps each) and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU")
    ap.add_argument("--vision-tokens", type=int, default=197)
    ap.add_argument("--text-tokens", type=int, default=40)
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg 3/4/5 and 12-layer encoder entries")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying a HIP graph")
    ap.add_argument("--torch-noise", action="store_true",
                    help="draw the gate's Gumbel noise from torch's device generator (a torch.rand kernel per step, and "
                         "torch's generator bookkeeping before every graph replay) instead of inside the gate kernel "
                         "from a device-resident Philox state (eve_amd.DeviceNoise)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--allow-foreign-lib", action="store_true",
                    help="run although EVE_MOE_LIB selects another library than eve_amd/libeve_moe.so (kernel A/B "
                         "experiments); the line then names it.  Without the flag such a run is refused.")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: after the timed region, check one expert-parallel step of every rank against the "
                         "single-rank layer (all experts local) on the same batch and noise; adds `verify` to the line")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ N > 1 launcher
def self_launch(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE: start the N ranks as child processes.  Nothing in this
    parent has touched the GPU (importing torch and counting devices do not initialise HIP), the children are
    fresh interpreters, and the parent only waits and passes rank 0's output through."""
    n = args.gpus
    rehearsal = os.environ.get("EVE_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if not rehearsal and have < n:
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) visible (EVE_BENCH_REHEARSAL=1 runs all ranks on "
                         "cuda:0 over gloo as a functional rehearsal, not a measurement)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1) // 2)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    deadline = time.time() + 1500
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:  # one rank failed: the others would wait in a collective forever
                    q.terminate()
        if time.time() > deadline:
            for q in alive:
                q.kill()
            return 124
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------ workload
DEVICE_NOISE = True  # main(): False under --torch-noise


def build_layer(M, I, E, ep, rank, n_vis, dtype, dev):
    """One modality-aware MoE layer with independent experts, initialised on the GPU."""
    import eve_amd
    torch.manual_seed(1234)  # replicated parameters (gate, modality embeddings) identical on all ranks
    moe = eve_amd.MoE(M, eve_amd.MultiwayTransformerMLP("quick_gelu", M, I), num_experts=E, ep_size=ep, k=2,
                      num_tokens=n_vis)
    with torch.no_grad():
        torch.nn.init.trunc_normal_(moe.moe.gate.wg.weight, std=0.02, a=-0.02, b=0.02)
    moe = moe.to(dtype).to(dev)
    with torch.no_grad():
        e_local = E // ep
        for j, ex in enumerate(moe.moe.experts.experts):
            g = torch.Generator(device=dev).manual_seed(1000 + rank * e_local + j)
            for lin in (ex.fc1, ex.fc2):
                w = torch.empty(lin.weight.shape, dtype=torch.float32, device=dev)
                torch.nn.init.trunc_normal_(w, std=0.02, a=-0.02, b=0.02, generator=g)
                lin.weight.copy_(w)
                lin.bias.zero_()
    # exp_counts as an asynchronous pinned copy (opt-in; the default is the reference's blocking .to('cpu')):
    # a timed step must not contain a host synchronisation per layer
    moe.moe.blocking_exp_counts = False
    if DEVICE_NOISE:  # fresh Gumbel noise every step either way; here the gate kernel draws it (eve_amd.DeviceNoise)
        moe.set_noise_source("device", seed=777 + rank)
    return moe


def synth_batch(B, T, M, dtype, dev, seed):
    g = torch.Generator().manual_seed(seed)
    h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(dtype).to(dev)
    dout = torch.randn(B, T, M, generator=g).to(dtype).to(dev)
    return h, dout


def cpu_baseline(args, M, I, E, n_vis, T):
    """The oracle's dense one-hot port of the reference algorithm (what models/sharded_moe.py +
    models/experts.py execute), fp32 PyTorch on the host cores, fwd+bwd, on a bounded sample.  The experts'
    parameters are separate leaf tensors, as the reference's ModuleList of nn.Linear (models/experts.py:12-16):
    with stacked leaves the backward of every `w1[e]` select materialises a full [E,I,M] zero tensor."""
    from oracle import moe_oracle as orc  # CPU baseline leg only
    B = 8
    cores = torch.get_num_threads()
    raw = orc.make_params(M, I, E, seed=1234)
    p = {k: ([t.clone().requires_grad_(True) for t in v] if k in ("w1", "b1", "w2", "b2")
             else v.clone().requires_grad_(True)) for k, v in raw.items()}
    leaves = [t for v in p.values() for t in (v if isinstance(v, list) else [v])]
    h = orc.synth_hidden(B, T, M, seed=7)
    g = torch.Generator().manual_seed(8)
    noise = orc.gumbel_noise((B * T, E), generator=g)
    fwd_s = bwd_s = 0.0

    def step(timed):
        nonlocal fwd_s, bwd_s
        for v in leaves:
            v.grad = None
        hh = h.clone().requires_grad_(True)
        t0 = time.perf_counter()
        out, l_aux, _ = orc.dense_moe_forward(hh, p, noise, mode="vl", n_vis=n_vis)
        t1 = time.perf_counter()
        (out.sum() + l_aux).backward()
        t2 = time.perf_counter()
        if timed:
            fwd_s += t1 - t0
            bwd_s += t2 - t1

    step(False)
    step(False)
    n, t0 = 0, time.perf_counter()
    while True:
        step(True)
        n += 1
        el = time.perf_counter() - t0
        if (n >= 5 and el >= args.cpu_baseline_seconds) or n >= 200:
            break
    return {"value": round(B * T * n / el, 1), "unit": "tokens/s", "cores": cores, "kind": "port",
            "fwd_ms": round(fwd_s / n * 1e3, 1), "bwd_ms": round(bwd_s / n * 1e3, 1),
            "sample": f"B={B} of the {args.batch} sequences (S={B * T} tokens, EVE-{args.model} layer dims), fp32 dense "
                      f"one-hot/einsum port of the reference algorithm with per-expert leaf parameters, fwd+bwd, "
                      f"2 warm-up + {n} timed iterations"}


def verify_against_single_rank(moe, h, dout, M, I, E, n_vis, dtype, dev, rank, world):
    """One more step of the expert-parallel layer with a pinned Gumbel draw, and the same step on a single-rank layer
    that holds all E experts (same parameters: build_layer seeds experts by their global index): a rank's own
    tokens go through the same kernels row by row in both layouts, so output and input gradient must be bit-equal.
    The flags are AND-ed over the ranks."""
    from eve_amd import ops
    S = h.shape[0] * h.shape[1]
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    noise = ops.gumbel_from_uniform(torch.rand(S, E, generator=g, device=dev))

    def run(layer):
        for p in layer.parameters():
            p.grad = None
        x = h.detach().requires_grad_(True)
        layer.moe.noise_override = noise
        out, l_aux, _ = layer(x, mode="vl")
        torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])
        torch.cuda.synchronize()
        return out.detach(), x.grad, layer.moe.last_slot.clone()

    out, dh, slot = run(moe)
    one = build_layer(M, I, E, 1, 0, n_vis, dtype, dev)
    out1, dh1, slot1 = run(one)
    d_out = float((out.float() - out1.float()).abs().max())
    d_dh = float((dh.float() - dh1.float()).abs().max())
    flags = torch.tensor([float(torch.equal(out, out1)), float(torch.equal(dh, dh1)), float(torch.equal(slot, slot1)),
                          -d_out, -d_dh], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    del one
    torch.cuda.empty_cache()
    f = flags.tolist()
    return {"against": "single-rank layer (ep_size=1, all experts local), same batch and Gumbel draw, every rank",
            "out_bit_equal": bool(f[0]), "dh_bit_equal": bool(f[1]), "routing_equal": bool(f[2]),
            "max_abs_out_diff": -f[3], "max_abs_dh_diff": -f[4]}


def hbm_rooflines(prof, steps, S, K, M, E, b):
    """Gate / permute stages against the HBM roof: ALGORITHMIC bytes per call (`b` bytes per activation
    element, K kept assignments) / the HIP-event time of the call on the launch stream / 8 TB/s.  A call is one to
    three kernels and its event pair also spans their launch gaps, so these fractions are lower bounds on the
    kernels' own (profiles/r03_kernel_roofline.txt has the rocprofv3 per-kernel durations and the PMC traffic)."""
    alg = {
        "eve_gate_logits_fwd": S * M * b + 4 * S * E,
        "eve_gate_top2_fwd": (12 * E + 24) * S,
        "eve_dispatch_fwd": (S + K) * M * b,
        "eve_combine_fwd": (S + K) * M * b,
        "eve_combine_bwd": (S + 2 * K) * M * b,
        "eve_gate_top2_bwd": 12 * E * S,
        "eve_route_bwd": (S + K) * M * b,
        "eve_gate_wgrad": S * M * b + 4 * S * E,
    }
    out = {}
    for name, nbytes in alg.items():
        if name in prof and prof[name][1]:
            us = prof[name][0] / prof[name][1] * 1e3
            gbs = nbytes / (us * 1e-6) / 1e9
            out[name] = {"us_per_call": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 2),
                         "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def gemm_frac(prof, steps, M, I, kept, peak=MFMA_BF16_PEAK_TFLOPS):
    """Per-launch rate of the grouped expert GEMMs from a per-call profile: (TFLOP/s, fraction of the MFMA peak, ms)."""
    ffn_ms = prof.get("eve_ffn_fwd", (0.0, 0))[0] + prof.get("eve_ffn_bwd", (0.0, 0))[0]
    launches = 2 * prof.get("eve_ffn_fwd", (0.0, 0))[1] + 3 * prof.get("eve_ffn_bwd", (0.0, 0))[1]
    if ffn_ms <= 0 or launches == 0:
        return None
    gemms = 2 * prof.get("eve_ffn_fwd", (0.0, 0))[1] + 4 * prof.get("eve_ffn_bwd", (0.0, 0))[1]
    tflops = gemms * 2.0 * M * I * kept / (ffn_ms * 1e-3) / 1e12
    return {"tflops": round(tflops, 1), "frac": round(tflops / peak, 4), "avg_launch_ms": round(ffn_ms / launches, 4)}


def time_layer(moe, h, dout, steps, warmup, train=True):
    """ms per eager step of `moe` on (h, dout); train: fwd+bwd, else eval forward under no_grad."""
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
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def torch_noise_companion(args, M, I, E, n_vis, dtype, dev, h, dout):
    """The headline step once more with the Gumbel noise drawn by torch.rand from torch's device generator (the
    reference's own source, sharded_moe.py:70-77), same layer parameters, same inputs, same timing protocol: what
    the in-kernel draw of `value` is worth (the work per step is the same: one fresh [S,E] draw)."""
    global DEVICE_NOISE
    import eve_amd
    DEVICE_NOISE = False
    try:
        moe = build_layer(M, I, E, 1, 0, n_vis, dtype, dev)
    finally:
        DEVICE_NOISE = True
    moe.moe.track_last_slot = False
    graphed = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
    for _ in range(args.warmup):
        graphed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graphed()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    S = h.shape[0] * h.shape[1]
    del graphed, moe
    torch.cuda.empty_cache()
    return {"ms_per_step": round(ms, 4), "tokens_per_s": round(S / ms * 1e3, 1),
            "gumbel_noise": "torch.rand on torch's device generator, inside the replayed graph"}


def other_configs(dev, steps=10, warmup=4):
    """The other BASELINE.json layer shapes on ONE MI355X, all 32 experts local, eager issue, a few steps each
    (the headline entry above is configs[1]'s layer; configs[0] is the CPU case = cpu_baseline)."""
    M, I, E = 1024, 4096, 32
    out = []
    moe = build_layer(M, I, E, 1, 0, 197, torch.bfloat16, dev)
    cases = [("cfg3", "EVE-Large MoE layer, 224 px, B=32/GPU x 237 tokens (the per-rank shape of configs[2]), fwd+bwd", 32, 197, True),
             ("cfg4", "EVE-Large MoE layer, 480 px VQA shape, B=16/GPU x (901 + 40) tokens (configs[3]), fwd+bwd", 16, 901, True),
             ("cfg5", "EVE-Large MoE layer, 384 px rerank inference, B=128 x (577 + 40) tokens = 78 976 tokens per forward "
                      "(configs[4]), eval, forward only, no grad", 128, 577, False)]
    for name, what, B, n_vis, train in cases:
        T = n_vis + 40
        moe.moe.num_vision_tokens = n_vis
        h, dout = synth_batch(B, T, M, torch.bfloat16, dev, seed=1)
        ms = time_layer(moe, h, dout, steps, warmup, train)
        kept = int((moe.moe.last_slot >= 0).sum())
        flops = (12 if train else 4) * M * I * kept
        from eve_amd import ops
        ops.PROFILE.enable()  # the same steps once more, one C call per stage with HIP events around each
        time_layer(moe, h, dout, 4, 0, train)
        prof = ops.PROFILE.disable()
        out.append({"name": name, "workload": what, "tokens": B * T, "ms_per_step": round(ms, 4),
                    "tokens_per_s": round(B * T / ms * 1e3, 1), "kept_assignments": kept,
                    "expert_gemm_tflops_over_whole_step": round(flops / ms / 1e9, 1),
                    "expert_gemm_per_launch": gemm_frac(prof, 4, M, I, kept),
                    "roofline_hbm": hbm_rooflines(prof, 4, B * T, kept, M, E, 2), "launch": "eager", "steps": steps})
        del h, dout
    del moe
    torch.cuda.empty_cache()
    return out


def encoder12(dev, steps=8, warmup=4):
    """BASELINE.json configs[1] as a whole: EVE-Base 12-layer encoder (MoE in layers 10-11, hard modality split in the
    others, stock fused attention), batch 64 x 237 tokens, bf16, fwd+bwd incl. the gate loss."""
    import eve_amd
    torch.manual_seed(0)
    enc = eve_amd.eve_base_encoder().to(torch.bfloat16).to(dev).train()
    for layer in enc.layers:
        moe = getattr(layer, "mlp_moe", None)
        if moe is not None:
            moe.moe.blocking_exp_counts = False
    B, T, M = 64, 237, 768
    x0, dout = synth_batch(B, T, M, torch.bfloat16, dev, seed=1)
    mask = torch.ones(B, T, dtype=torch.long, device=dev)
    params = list(enc.parameters())

    def step():
        for p in params:
            p.grad = None
        x = x0.detach().requires_grad_(True)
        out = enc(x, mask, mode="vl")
        gl = enc.gate_loss(0.001)
        torch.autograd.backward([out, gl], [dout, torch.ones_like(gl)])

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    enqueue_ms = (time.perf_counter() - t0) / steps * 1e3  # host time to issue a step: close to ms_per_step = host-bound
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # composition of the step: HIP events around every call into this library (one call per stage) over a few more
    # steps; the remainder is stock PyTorch-ROCm (attention, norm1 / norm2_vl, the optimizer-free glue)
    from eve_amd import ops
    nprof = 3
    ops.PROFILE.enable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nprof):
        step()
    prof = ops.PROFILE.disable()
    prof_ms = (time.perf_counter() - t0) / nprof * 1e3
    lib_ms = sum(v[0] for v in prof.values()) / nprof
    tagged = ops.PROFILE.tagged
    split_ms = (tagged.get("eve_ffn_fwd[G=2]", (0.0, 0))[0] + tagged.get("eve_ffn_bwd[G=2]", (0.0, 0))[0]) / nprof
    n_split = tagged.get("eve_ffn_fwd[G=2]", (0.0, 0))[1] // nprof
    q = torch.randn(B, 12, T, 64, dtype=torch.bfloat16, device=dev, requires_grad=True)

    def attn():
        q.grad = None
        torch.nn.functional.scaled_dot_product_attention(q, q, q).sum().backward()

    for _ in range(3):
        attn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        attn()
    torch.cuda.synchronize()
    attn_ms = (time.perf_counter() - t0) / 10 * 1e3 * 12
    res = {"workload": "EVE-Base encoder, 12 layers, B=64 x (197 + 40) tokens, bf16, fwd+bwd (BASELINE configs[1]); attention "
                       "is PyTorch-ROCm's fused SDPA, the FFN branches are this library", "ms_per_step": round(ms, 3),
           "tokens_per_s": round(B * T / ms * 1e3, 1), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "launch": "eager", "host_enqueue_ms_per_step": round(enqueue_ms, 3), "steps": steps,
           "composition": {
               "ms_per_step_while_profiled": round(prof_ms, 3),
               "library_calls_ms": round(lib_ms, 3), "library_share": round(lib_ms / prof_ms, 3),
               "stock_fused_attention_ms_standalone": round(attn_ms, 3),
               "stock_other_ms": round(max(prof_ms - lib_ms - attn_ms, 0.0), 3),
               "split_ffn_layers": n_split,
               "split_ffn_gemm_ms": round(split_ms, 3),
               "split_ffn_gemm_tflops": round(n_split * 12.0 * M * 3072 * B * T / max(split_ms, 1e-9) / 1e9, 1),
               "split_ffn_gemm_frac_of_mfma_peak": round(n_split * 12.0 * M * 3072 * B * T / max(split_ms, 1e-9) / 1e9
                                                         / MFMA_BF16_PEAK_TFLOPS, 4),
               "by_call_ms": {k: round(v[0] / nprof, 3) for k, v in sorted(prof.items())},
               "note": "HIP events on the launch stream around each call of this library (eager issue); attention timed "
                       "alone at the layer's shape (12 heads x 64) x 12 layers"}}
    del q
    del enc, params
    torch.cuda.empty_cache()
    return res


def gemm_traffic(world, args, B):
    """HBM bytes per GEMM launch from separate rocprofv3 --pmc passes of THIS workload (tools/collect_profiles.sh:
    FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE); counters cannot be read from inside the
    process.  Only trusted when the file was collected with the library that is loaded now (source digest)."""
    from eve_amd import _lib
    if not (world == 1 and args.dtype == "bf16" and args.model == "base" and B == 64):
        return None, None
    for name in ("r03_gemm_traffic.json", "r02_gemm_traffic.json", "r01_gemm_traffic.json"):
        path = os.path.join(REPO, "profiles", name)
        if not os.path.exists(path):
            continue
        rec = json.load(open(path))
        if rec.get("source_hash") == _lib.source_hash():
            return rec["avg_hbm_bytes_per_launch"], f"profiles/{name}"
    return None, "no rocprofv3 --pmc collection matches the loaded library's source digest"


def main():
    args = parse()
    foreign = os.environ.get("EVE_MOE_LIB")
    if foreign and not args.allow_foreign_lib:
        raise SystemExit(f"bench.py: EVE_MOE_LIB={foreign} selects a library that is exempt from the source-digest check; "
                         "unset it, or pass --allow-foreign-lib for an A/B experiment (the line then says so)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    global DEVICE_NOISE
    DEVICE_NOISE = not args.torch_noise
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    # EVE_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo (the all-to-all goes through host memory), to run the
    # N > 1 code path on a one-GPU box.  Its numbers mean nothing and the line says so.
    rehearsal = os.environ.get("EVE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from eve_amd import ops

    M, I = (768, 3072) if args.model == "base" else (1024, 4096)
    E, n_vis = 32, args.vision_tokens
    T = n_vis + args.text_tokens
    B = args.batch
    S = B * T
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    moe = build_layer(M, I, E, world, rank, n_vis, dtype, dev)
    h, dout = synth_batch(B, T, M, dtype, dev, seed=1234 + rank)
    params = [p for p in moe.parameters()]

    def eager_step():
        for p in params:
            p.grad = None
        x = h.detach().requires_grad_(True)
        out, l_aux, _ = moe(x, mode="vl")
        torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])

    # One step is ~25 launches; issued from Python the host is slower than the GPU, so on one GPU the
    # step (noise draw, forward, full backward) is captured once into a HIP graph and replayed.  With
    # expert parallelism the step contains RCCL all-to-alls and is issued eagerly.
    use_graph = world == 1 and not args.no_graph
    if use_graph:
        import eve_amd
        moe.moe.track_last_slot = False  # the diagnostic copy of the routing map is not part of a training step
        graphed = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
        moe.moe.track_last_slot = True   # (the eager profile steps below read it for the kept-row count)
        # h / dout are resident in the graph's input buffers (graphed.static_h / static_dout), as when the
        # producing layer writes them in place; the eager path reads them in place as well
        step = lambda: graphed()
    else:
        step = eager_step

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    enqueue_s = time.perf_counter() - t0  # host time to issue the K steps (close to `elapsed` = host-bound)
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # Per-kernel HIP-event timing (on the launch stream) of the same K steps, issued eagerly right after
    # the timed region: events cannot be recorded inside a graph replay.  Kernel durations are the
    # same in both modes (profiles/ holds the rocprofv3 trace of this command).
    ops.PROFILE.enable()
    for _ in range(args.steps):
        eager_step()
    sync()
    prof = ops.PROFILE.disable()

    # kept assignments of the last step (exact K for the algorithmic flop count)
    kept = int((moe.moe.last_slot >= 0).sum().item())
    C = ops.capacity(S, E, 1.0, 4)
    ffn_ms = prof.get("eve_ffn_fwd", (0.0, 0))[0] + prof.get("eve_ffn_bwd", (0.0, 0))[0]
    # Per step: 6 grouped GEMMs of 2*M*I*K flops each; an eve_ffn_fwd call is 2 launches, an eve_ffn_bwd call 3 (the
    # two weight-gradient GEMMs share one): 5 launches per step on one GPU, 5 per expert chunk with expert parallelism.
    # Algorithmic flops count KEPT rows only.  With expert parallelism the owner does not know its peers' fill
    # counts and processes every capacity row it received (E*C per rank): `achieved` stays algorithmic (rank 0's own
    # kept count stands in for the rows its experts received: about as many) and `processed_tflops` counts E*C rows.
    launches = 2 * prof.get("eve_ffn_fwd", (0.0, 0))[1] + 3 * prof.get("eve_ffn_bwd", (0.0, 0))[1]
    flops_per_launch = 6.0 * 2.0 * M * I * kept * args.steps / max(launches, 1)
    peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
    roofline = None
    if ffn_ms > 0:
        avg_ms = ffn_ms / launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = gemm_traffic(world, args, B)
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": f"grouped expert-FFN GEMM (eve::grouped_gemm_pp*, {launches // max(args.steps, 1)} launches/step computing 6 GEMMs)",
                    "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": flops_per_launch,
                    "timing": "HIP events on the launch stream around eve_ffn_fwd / eve_ffn_bwd over the K steps"}
        if world > 1:
            roofline["processed_tflops"] = round(achieved * (E * C) / max(kept, 1), 2)
            roofline["note"] = ("expert parallel: every capacity row received is processed (E*C rows per rank, empty ones "
                                "included); `achieved` counts kept rows only, `processed_tflops` all processed rows")

    verify = None
    if args.verify and world > 1:
        verify = verify_against_single_rank(moe, h, dout, M, I, E, n_vis, dtype, dev, rank, world)

    if rank == 0:
        from eve_amd import _lib
        line = {
            "metric": "MoE-layer fwd+bwd tokens/sec", "value": round(S * world * args.steps / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EVE-{args.model} modality-aware MoE layer fwd+bwd (BASELINE configs[1] layer shape): "
                                   f"B={B}/GPU x ({n_vis} vision + {args.text_tokens} text) tokens, M={M}, I={I}, E={E} top-2, "
                                   f"capacity_factor 1.0, mode vl", "tokens_per_gpu": S, "kept_assignments": kept,
                       "parallelism": f"ep{world}" if world > 1 else "single",
                       "launch": "hipgraph replay of the whole step" if use_graph else "eager, one launch per kernel",
                       "gumbel_noise": ("fresh every step, drawn inside the gate kernel from a device-resident Philox4x32-10 state"
                                        if DEVICE_NOISE else "fresh every step, torch.rand on torch's device generator"),
                       "library_source_hash": _lib.source_hash()},
            "roofline": roofline,
            "roofline_hbm": hbm_rooflines(prof, args.steps, S, kept, M, E, 2 if args.dtype == "bf16" else 4),
            "phases_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
        }
        if foreign:
            line["foreign_library"] = foreign  # NOT the shipped library: an A/B experiment, not a result
        if verify is not None:
            line["verify"] = verify
        if rehearsal:
            line["rehearsal"] = "all ranks on cuda:0 over gloo with a host-staged exchange: NOT a measurement"
        if world == 1:
            if use_graph and DEVICE_NOISE:
                line["same_step_with_torch_generator_noise"] = torch_noise_companion(args, M, I, E, n_vis, dtype, dev, h, dout)
            del moe, params, h, dout
            if use_graph:
                del graphed, step
            torch.cuda.empty_cache()
            if not args.no_other_configs:
                line["other_configs"] = other_configs(dev)
                line["encoder12"] = encoder12(dev)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args, M, I, E, n_vis, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
