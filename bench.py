"""bench.py -- MoE-layer forward+backward tokens/s on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one forward + backward of ONE EVE-Base modality-aware MoE layer over one synthetic
image+text token batch (BASELINE.json configs[1] layer shape: B=64 per GPU, 197 vision + 40
text tokens, M=768, I=3072, 32 experts, top-2, bf16), inputs resident in HBM.  With N > 1 the
32 experts are sharded over the N ranks (expert parallel, RCCL all-to-all) and every rank
brings its own batch (weak scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
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
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying a HIP graph")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    return ap.parse_args()


def build_layer(M, I, E, ep, rank, n_vis, dtype, dev):
    import eve_amd
    torch.manual_seed(1234)  # replicated parameters (gate, modality embeddings) identical on all ranks
    moe = eve_amd.MoE(M, eve_amd.MultiwayTransformerMLP("quick_gelu", M, I), num_experts=E, ep_size=ep, k=2,
                      num_tokens=n_vis)
    with torch.no_grad():
        torch.nn.init.trunc_normal_(moe.moe.gate.wg.weight, std=0.02, a=-0.02, b=0.02)
        e_local = E // ep
        for j, ex in enumerate(moe.moe.experts.experts):  # independent experts
            g = torch.Generator().manual_seed(1000 + rank * e_local + j)
            for lin in (ex.fc1, ex.fc2):
                torch.nn.init.trunc_normal_(lin.weight, std=0.02, a=-0.02, b=0.02, generator=g)
                lin.bias.zero_()
    return moe.to(dtype).to(dev)


def cpu_baseline(args, M, I, E, n_vis, T):
    """The oracle's dense one-hot port of the reference algorithm (what models/sharded_moe.py +
    models/experts.py execute), fp32 PyTorch on the host cores, fwd+bwd, on a bounded sample."""
    from oracle import moe_oracle as orc  # CPU baseline leg only
    B = 8
    cores = torch.get_num_threads()
    p = {k: v.requires_grad_(True) for k, v in orc.make_params(M, I, E, seed=1234).items()}
    h = orc.synth_hidden(B, T, M, seed=7)
    g = torch.Generator().manual_seed(8)
    noise = orc.gumbel_noise((B * T, E), generator=g)

    def step():
        for v in p.values():
            v.grad = None
        hh = h.clone().requires_grad_(True)
        out, l_aux, _ = orc.dense_moe_forward(hh, p, noise, mode="vl", n_vis=n_vis)
        (out.sum() + l_aux).backward()

    step()
    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if (n >= 5 and el >= args.cpu_baseline_seconds) or n >= 200:
            break
    return {"value": round(B * T * n / el, 1), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"B={B} of the {args.batch} sequences (S={B * T} tokens, EVE-{args.model} layer dims), fp32 dense "
                      f"one-hot/einsum port of the reference algorithm, fwd+bwd, 2 warm-up + {n} timed iterations"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
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

    g = torch.Generator().manual_seed(1234 + rank)
    h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(dtype).to(dev)
    dout = torch.randn(B, T, M, generator=g).to(dtype).to(dev)
    params = [p for p in moe.parameters()]

    def eager_step():
        for p in params:
            p.grad = None
        x = h.detach().requires_grad_(True)
        out, l_aux, _ = moe(x, mode="vl")
        torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])

    # One step is ~50 launches; issued from Python the host is slower than the GPU, so on one GPU the
    # step (noise draw, forward, full backward) is captured once into a HIP graph and replayed.  With
    # expert parallelism the step contains RCCL all-to-alls and is issued eagerly.
    use_graph = world == 1 and not args.no_graph
    if use_graph:
        import eve_amd
        graphed = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
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
    ffn_ms = prof.get("eve_ffn_fwd", (0.0, 0))[0] + prof.get("eve_ffn_bwd", (0.0, 0))[0]
    # Per step: 6 grouped GEMMs of 2*M*I*K flops each; an eve_ffn_fwd call is 2 launches, an eve_ffn_bwd call 3 (the
    # two weight-gradient GEMMs share one): 5 launches per step on one GPU, 5 per expert chunk with expert parallelism
    # (there K counts rank 0's own kept assignments; the rows its experts receive from all ranks are about as many).
    launches = 2 * prof.get("eve_ffn_fwd", (0.0, 0))[1] + 3 * prof.get("eve_ffn_bwd", (0.0, 0))[1]
    flops_per_launch = 6.0 * 2.0 * M * I * kept * args.steps / max(launches, 1)
    peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
    roofline = None
    if ffn_ms > 0:
        avg_ms = ffn_ms / launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        tpath = os.path.join(REPO, "profiles", "r01_gemm_traffic.json")
        if world == 1 and args.dtype == "bf16" and args.model == "base" and B == 64 and os.path.exists(tpath):
            # HBM bytes per launch from separate rocprofv3 --pmc passes of this workload (FETCH_SIZE doubled as the
            # gfx950 guide prescribes, WRITE_SIZE); counters cannot be read live from inside the process
            traffic = json.load(open(tpath))["avg_hbm_bytes_per_launch"]
            traffic_src = "profiles/r01_gemm_traffic.json"
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": f"grouped expert-FFN GEMM (eve::grouped_gemm_pp, {launches // max(args.steps, 1)} launches/step computing 6 GEMMs)",
                    "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": flops_per_launch,
                    "timing": "HIP events on the launch stream around eve_ffn_fwd / eve_ffn_bwd over the K steps"}

    if rank == 0:
        line = {
            "metric": "MoE-layer fwd+bwd tokens/sec", "value": round(S * world * args.steps / elapsed, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EVE-{args.model} modality-aware MoE layer fwd+bwd (BASELINE configs[1] layer shape): "
                                   f"B={B}/GPU x ({n_vis} vision + {args.text_tokens} text) tokens, M={M}, I={I}, E={E} top-2, "
                                   f"capacity_factor 1.0, mode vl", "tokens_per_gpu": S, "kept_assignments": kept,
                       "parallelism": f"ep{world}" if world > 1 else "single",
                       "launch": "hipgraph replay of the whole step" if use_graph else "eager, one launch per kernel"},
            "roofline": roofline,
            "phases_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
        }
        if rehearsal:
            line["rehearsal"] = "all ranks on cuda:0 over gloo with a host-staged exchange: NOT a measurement"
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, M, I, E, n_vis, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
