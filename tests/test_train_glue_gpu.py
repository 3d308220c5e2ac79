"""Training-step glue around the layer, two ranks on one GPU with gloo:
DistributedDataParallel(find_unused_parameters=True) as the reference's drivers wrap the model (VQA.py:239-244),
gate-loss aggregation (model_vqa_beit.py:75-83), clip_grad_norm_, AdamW with the reference's decay / no-decay
parameter groups (optim.py:12-52), and a state-dict round trip (utils/checkpointer.py:26-52) through the per-expert
keys while the kernels read the packed expert storage."""
import copy
import io
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_DECAY = {"bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(ep_size=1):
    from eve_amd.encoder import MultiwayTransformerEncoder
    torch.manual_seed(3)
    enc = MultiwayTransformerEncoder(hidden_size=64, hidden_act="quick_gelu", num_attention_heads=4, intermediate_size=128,
                                     num_hidden_layers=2, num_tokens=(5, 4), drop_path_rate=0.0, ffn_type="moe",
                                     ffn_layers=[1], ffn_param={"num_experts": 8, "topk": 2, "type_route": True,
                                                                "ep_size": ep_size},
                                     single_ffn={"type": "vl"})
    with torch.no_grad():
        for e in enc.layers[1].mlp_moe.moe.experts.experts:  # deep copies of one template: make them differ
            for p in e.parameters():
                p.add_(torch.randn_like(p) * 0.02)
    return enc


def _data(rank):
    g = torch.Generator().manual_seed(50 + rank)
    h = torch.randn(3, 9, 64, generator=g)
    dout = torch.randn(3, 9, 64, generator=g)
    u = torch.rand(27, 8, generator=g).clamp_(1e-6, 1 - 1e-6)
    return h, dout, -torch.log(-torch.log(u))


def _loss(enc, h, dout, noise, dev):
    core = enc.module if hasattr(enc, "module") else enc
    core.layers[1].mlp_moe.moe.noise_override = noise.to(dev)
    out = enc(h.to(dev), torch.ones(3, 9, dtype=torch.long, device=dev), mode="vl")
    return (out * dout.to(dev)).sum() + core.gate_loss(0.1)


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        enc = _model().to(dev).train()
        ref = copy.deepcopy(enc)  # single-process arm: gradients of every rank's batch, averaged by hand
        want = None
        for r in range(world):
            ref.zero_grad(set_to_none=True)
            _loss(ref, *_data(r), dev).backward()
            gr = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ref.named_parameters()}
            want = gr if want is None else {n: want[n] + gr[n] for n in gr}
        want = {n: v / world for n, v in want.items()}

        ddp = torch.nn.parallel.DistributedDataParallel(enc, find_unused_parameters=True)
        _loss(ddp, *_data(rank), dev).backward()
        for n, p in enc.named_parameters():
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            torch.testing.assert_close(got, want[n], rtol=1e-4, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")

        groups = [{"params": [], "weight_decay": 0.05}, {"params": [], "weight_decay": 0.0}]
        for n, p in enc.named_parameters():
            groups[1 if any(nd in n for nd in NO_DECAY) else 0]["params"].append(p)
        assert groups[0]["params"] and groups[1]["params"]
        opt = torch.optim.AdamW(groups, lr=1e-3, eps=1e-8, betas=(0.9, 0.98))
        torch.nn.utils.clip_grad_norm_(enc.parameters(), 1.0)
        before = enc.layers[1].mlp_moe.moe.experts.experts[2].fc1.weight.detach().clone()
        opt.step()
        ex = enc.layers[1].mlp_moe.moe.experts
        assert not torch.equal(before, ex.experts[2].fc1.weight.detach())
        # the optimizer updated the per-expert Parameters in place: they still alias the packed buffers
        bank = ex._bank
        assert bank._aliased(bank.w1, [m.fc1.weight for m in ex.experts]) and bank.w1.data_ptr() == ex.experts[0].fc1.weight.data_ptr()
        # replicas stay identical
        flat = torch.cat([p.detach().flatten() for p in enc.parameters()]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])
        # a second step runs on the updated weights
        opt.zero_grad(set_to_none=True)
        _loss(ddp, *_data(rank), dev).backward()

        # checkpoint round trip through the flat state dict with per-expert keys
        buf = io.BytesIO()
        torch.save(enc.state_dict(), buf)
        buf.seek(0)
        sd = torch.load(buf, map_location="cpu", weights_only=True)
        assert "layers.1.mlp_moe.moe.experts.experts.7.fc2.weight" in sd and "layers.1.mlp_moe.moe.gate.wg.weight" in sd
        fresh = _model()
        msg = fresh.load_state_dict(sd, strict=False)  # the reference loads with strict=False (model_vqa_beit.py:59)
        assert not msg.missing_keys and not msg.unexpected_keys
        fresh = fresh.to(dev).eval()
        enc.eval()
        h, _, noise = _data(rank)
        with torch.no_grad():
            enc.layers[1].mlp_moe.moe.noise_override = noise.to(dev)
            a = enc(h.to(dev), None, mode="vl")
            fresh.layers[1].mlp_moe.moe.noise_override = noise.to(dev)
            b = fresh(h.to(dev), None, mode="vl")
        assert torch.equal(a, b)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]))


def test_ddp_clip_adamw_checkpoint_two_ranks_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _ep_worker(rank, world, port, q):
    """Expert parallelism INSIDE data parallelism, as the reference's drivers would run it (DDP around the whole
    model, VQA.py:239-244; experts sharded, models/layer.py:54-58): MoE(ep_size=2) in the encoder, the expert
    parameters on DDP's ignore list.  Expected from the single-process model (all 8 experts, same kernels):
    replicated parameters get the ranks' AVERAGE (DDP), expert parameters on their owner the SUM over the ranks'
    tokens; one AdamW step leaves the replicas' shared parameters identical."""
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ["EVE_EP_CHUNKS"] = "1"
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        ref = _model().to(dev).train()  # all 8 experts
        E, e_local = 8, 8 // world
        lo = rank * e_local
        enc = _model(ep_size=world)     # this rank's 4 experts, local names experts.0..3
        sd = ref.state_dict()
        mine = {}
        for k, v in sd.items():
            if ".experts.experts." in k:
                head, tail = k.split(".experts.experts.")
                e, rest = tail.split(".", 1)
                if lo <= int(e) < lo + e_local:
                    mine[f"{head}.experts.experts.{int(e) - lo}.{rest}"] = v
            else:
                mine[k] = v
        msg = enc.load_state_dict(mine, strict=True)
        assert not msg.missing_keys and not msg.unexpected_keys
        enc = enc.to(dev).train()
        expert_names = [n for n, _ in enc.named_parameters() if ".experts.experts." in n]
        assert len(expert_names) == 4 * e_local

        per_rank = []
        for r in range(world):
            ref.zero_grad(set_to_none=True)
            _loss(ref, *_data(r), dev).backward()
            per_rank.append({n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ref.named_parameters()})

        torch.nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(enc, expert_names)
        ddp = torch.nn.parallel.DistributedDataParallel(enc, find_unused_parameters=True)
        _loss(ddp, *_data(rank), dev).backward()
        for n, p in enc.named_parameters():
            assert p.grad is not None, n
            if ".experts.experts." in n:
                head, tail = n.split(".experts.experts.")
                e, rest = tail.split(".", 1)
                gname = f"{head}.experts.experts.{int(e) + lo}.{rest}"
                want = sum(per_rank[r][gname] for r in range(world))          # SUM over the ranks' tokens
            else:
                want = sum(per_rank[r][n] for r in range(world)) / world       # DDP's average
            torch.testing.assert_close(p.grad, want, rtol=1e-4, atol=2e-6, msg=lambda m, n=n: f"{n}: {m}")

        opt = torch.optim.AdamW(enc.parameters(), lr=1e-3, weight_decay=0.05)
        opt.step()
        shared = torch.cat([p.detach().flatten() for n, p in enc.named_parameters() if ".experts.experts." not in n]).cpu()
        both = [torch.empty_like(shared) for _ in range(world)]
        dist.all_gather(both, shared)
        assert torch.equal(both[0], both[1]), "replicated parameters diverged between the ranks"
        own = torch.cat([p.detach().flatten() for n, p in enc.named_parameters() if ".experts.experts." in n]).cpu()
        both = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(both, own)
        assert not torch.equal(both[0], both[1])  # different experts live on different ranks
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]))


def test_expert_parallel_inside_ddp_two_ranks_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
