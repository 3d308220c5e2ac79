"""HIP permute (dispatch / combine) and grouped-FFN kernels vs plain torch CPU arithmetic."""
import pytest
import torch

from oracle import moe_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: dict(rtol=1e-5, atol=1e-5), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}


def _routing(S, E, seed, cf=1.0):
    from eve_amd import ops
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(S, E, generator=g)
    noise = orc.gumbel_noise((S, E), generator=g)
    C = ops.capacity(S, E, cf, 4)
    return ops.gate_top2(logits.to(DEV), noise.to(DEV), C), C


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,n_vis,M,E,mode", [(4, 9, 5, 64, 8, "vl"), (8, 237, 197, 768, 32, "vl"),
                                               (3, 40, 7, 1024, 32, "text"), (6, 11, 4, 96, 4, "none")])
def test_dispatch_and_combine_roundtrip(dtype, B, T, n_vis, M, E, mode):
    from eve_amd import ops
    S = B * T
    r, C = _routing(S, E, seed=S + M, cf=0.75)
    g = torch.Generator().manual_seed(1)
    h = torch.randn(B, T, M, generator=g).to(dtype)
    p = {"vision_embed": (torch.randn(M, generator=g) * 0.1).to(dtype), "text_embed": (torch.randn(M, generator=g) * 0.1).to(dtype)}
    x = orc.modality_input(h, p, mode, n_vis if mode != "none" else None)          # [S,M] oracle
    ve = p["vision_embed"].to(DEV) if mode != "none" else None
    te = p["text_embed"].to(DEV) if mode != "none" else None
    xd = ops.dispatch_fwd(h.reshape(S, M).to(DEV), ve, te, r.slot_token, T, n_vis, ops.MODES[mode]).cpu()
    st = r.slot_token.cpu().long()
    ref = torch.zeros(E * C, M, dtype=dtype)
    ref[st >= 0] = x[st[st >= 0]]
    assert torch.equal(xd, ref), "dispatch must be an exact row gather"
    # with the forward map the rows move token-major (each token row read once for both slots): same result
    xd2 = ops.dispatch_fwd(h.reshape(S, M).to(DEV), ve, te, r.slot_token, T, n_vis, ops.MODES[mode], slot=r.slot).cpu()
    assert torch.equal(xd2, ref), "token-major dispatch must be the same exact row gather"

    # combine forward / backward against dense arithmetic in fp64
    y = torch.randn(E * C, M, generator=g).to(dtype)
    slot, w = r.slot.cpu().long(), r.weight.cpu()
    out = ops.combine_fwd(y.to(DEV), r.slot, r.weight).cpu()
    yd = y.double()
    ref_out = torch.zeros(S, M, dtype=torch.float64)
    for k in range(2):
        m = slot[:, k] >= 0
        ref_out[m] += w[m, k, None].double() * yd[slot[m, k]]
    torch.testing.assert_close(out.double(), ref_out, **TOL[dtype])

    dout = torch.randn(S, M, generator=g).to(dtype)
    dy, dw = ops.combine_bwd(dout.to(DEV), y.to(DEV), r.slot, r.slot_token, r.slot_weight)
    ref_dy = torch.zeros(E * C, M, dtype=torch.float64)
    ref_dw = torch.zeros(S, 2, dtype=torch.float64)
    for k in range(2):
        m = slot[:, k] >= 0
        ref_dy[slot[m, k]] = w[m, k, None].double() * dout[m].double()
        ref_dw[m, k] = (dout[m].double() * yd[slot[m, k]]).sum(1)
    torch.testing.assert_close(dy.cpu().double(), ref_dy, **TOL[dtype])
    torch.testing.assert_close(dw.cpu().double(), ref_dw, rtol=1e-4, atol=1e-3 if dtype == torch.bfloat16 else 1e-4)

    # dispatch backward: gather-add of slot rows + dlogits @ wg
    dxd = torch.randn(E * C, M, generator=g).to(dtype)
    dlog = torch.randn(S, E, generator=g)
    wg = torch.randn(E, M, generator=g) * 0.05
    dh = ops.dispatch_bwd(dxd.to(DEV), r.slot, dlog.to(DEV), wg.to(DEV), S).cpu()
    ref_dh = dlog.double() @ wg.double()
    for k in range(2):
        m = slot[:, k] >= 0
        ref_dh[m] += dxd.double()[slot[m, k]]
    torch.testing.assert_close(dh.double(), ref_dh, **TOL[dtype])

    if mode != "none":
        dv, dt = ops.embed_grad(dh.to(DEV), T, n_vis, ops.MODES[mode])
        d3 = dh.double().view(B, T, M)
        if mode == "vl":
            rv, rt = d3[:, :n_vis].sum((0, 1)), d3[:, n_vis:].sum((0, 1))
        elif mode == "vision":
            rv, rt = d3.sum((0, 1)), torch.zeros(M, dtype=torch.float64)
        else:
            rv, rt = torch.zeros(M, dtype=torch.float64), d3.sum((0, 1))
        torch.testing.assert_close(dv.cpu().double(), rv, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(dt.cpu().double(), rt, rtol=1e-4, atol=1e-3)


_ACT_REF = {"gelu": torch.nn.functional.gelu, "relu": torch.relu, "identity": lambda t: t}


def _ffn_ref(x, w1, b1, w2, b2, rows, E_l, act="quick_gelu"):
    G, R, M = x.shape
    ys = []
    for g in range(G):
        e = g % E_l
        if act == "quick_gelu":
            ys.append(orc.expert_ffn(x[g], w1[e], b1[e], w2[e], b2[e]))
        else:  # the other ACT2FN choices of models/multiway_transformer.py:47, straight from torch
            ys.append(_ACT_REF[act](x[g] @ w1[e].T + b1[e]) @ w2[e].T + b2[e])
    return torch.stack(ys)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("G,E_l,R,M,I,use_rows", [(4, 4, 7, 32, 64, False), (8, 4, 33, 64, 128, True),
                                                  (32, 32, 119, 768, 3072, True), (6, 2, 130, 256, 512, True),
                                                  (1, 1, 300, 256, 512, False), (4, 2, 700, 256, 256, True),
                                                  (2, 2, 1024, 512, 256, False), (8, 4, 474, 256, 768, True)])
def test_grouped_ffn_fwd_bwd(dtype, G, E_l, R, M, I, use_rows, act="quick_gelu"):
    from eve_amd import ops
    g = torch.Generator().manual_seed(G * R + M)
    x = torch.randn(G, R, M, generator=g)
    w1 = torch.randn(E_l, I, M, generator=g) * (M ** -0.5)
    b1 = torch.randn(E_l, I, generator=g) * 0.1
    w2 = torch.randn(E_l, M, I, generator=g) * (I ** -0.5)
    b2 = torch.randn(E_l, M, generator=g) * 0.1
    dy = torch.randn(G, R, M, generator=g)
    rows = torch.randint(0, R + 1, (G,), generator=g, dtype=torch.int32) if use_rows else None
    if rows is not None:
        rows[0] = R
        rows[-1] = 0
    cast = lambda t: t.to(dtype)
    xs, w1s, b1s, w2s, b2s, dys = map(cast, (x, w1, b1, w2, b2, dy))
    # reference in fp64 on the (possibly bf16-rounded) operands; rows beyond rows[g] masked out
    mask = torch.ones(G, R, 1, dtype=torch.float64)
    if rows is not None:
        mask = (torch.arange(R)[None, :, None] < rows[:, None, None]).double()
    leaves = [t.double().requires_grad_(True) for t in (xs, w1s, b1s, w2s, b2s)]
    yref = _ffn_ref(*leaves, rows, E_l, act)
    (yref * dys.double() * mask).sum().backward()

    d = lambda t: t.to(DEV) if t is not None else None
    z, a, y = ops.ffn_fwd(d(xs), d(w1s), d(b1s), d(w2s), d(b2s), d(rows), ops.ACTS[act])
    dx, dw1, db1, dw2, db2 = ops.ffn_bwd(d(dys), d(xs), z, a, d(w1s), d(w2s), d(rows), ops.ACTS[act])
    torch.cuda.synchronize()
    m3 = mask.bool().expand(G, R, M)
    tol = TOL[dtype]
    torch.testing.assert_close(y.cpu().double()[m3], yref.detach()[m3], **tol)
    torch.testing.assert_close(dx.cpu().double()[m3], leaves[0].grad[m3], **tol)
    for got, ref, name in ((dw1, leaves[1].grad, "dw1"), (db1, leaves[2].grad, "db1"), (dw2, leaves[3].grad, "dw2"),
                           (db2, leaves[4].grad, "db2")):
        scale = float(ref.abs().max().clamp(min=1e-6))
        err = float((got.cpu().double() - ref).abs().max()) / scale
        assert err < (2e-2 if dtype == torch.bfloat16 else 2e-5), (name, err)
        if dtype == torch.bfloat16:  # and element by element: rtol 2e-2 + 4 % of the tensor's rms (see tests/test_layer_gpu.py)
            from tests.helpers import assert_close_scaled
            assert_close_scaled(got, ref, 2e-2, 4e-2, name)


@pytest.mark.parametrize("act", ["gelu", "relu", "identity"])
@pytest.mark.parametrize("dtype,G,E_l,R,M,I", [(torch.float32, 4, 2, 33, 64, 128), (torch.bfloat16, 6, 2, 300, 256, 512),
                                               (torch.bfloat16, 3, 3, 130, 512, 256)])
def test_grouped_ffn_other_activations(act, dtype, G, E_l, R, M, I):
    """ACT2FN's other choices (models/multiway_transformer.py:47; EVE's configs use quick_gelu): the bf16 MFMA path
    saves act'(z) in the forward epilogue for every activation, not only for the one the benchmarks run."""
    test_grouped_ffn_fwd_bwd(dtype, G, E_l, R, M, I, True, act)


@pytest.mark.parametrize("G,E_l,R,M,I,use_rows", [
    # EVE-Base dims with few local experts: the weight gradients are split along K (E_local 1: 4 slices, 2: 2
    # slices, 4: none), over whole segments (G > E_local: expert parallelism) or over row slices of one segment
    (1, 1, 948, 768, 3072, False), (4, 1, 300, 768, 3072, True), (2, 2, 500, 768, 3072, True),
    (8, 2, 237, 768, 3072, True), (4, 4, 474, 768, 3072, True), (8, 4, 119, 768, 3072, True),
    # EVE-Large dims (M = 1024, I = 4096): every gradient against fp64
    (4, 4, 300, 1024, 4096, True), (2, 1, 474, 1024, 4096, True), (4, 2, 256, 1024, 4096, False),
    # few groups, one segment per expert, ragged rows[] (the modality split's shape): compact tile enumeration in
    # the forward / dgrad launches (two groups), weight gradients over dynamic K slices (round 3)
    (2, 2, 1900, 768, 3072, True), (3, 3, 700, 768, 3072, True), (2, 2, 900, 1024, 4096, True)])
def test_grouped_ffn_bf16_model_dims_and_split_k(G, E_l, R, M, I, use_rows):
    from eve_amd import _lib
    want_split = {(768, 1): 4, (768, 2): 2, (768, 3): 2, (768, 4): 1, (1024, 1): 2, (1024, 2): 1, (1024, 4): 1}[(M, E_l)]
    nbytes = _lib.load().eve_ffn_bwd_workspace_bytes(G, E_l, R, M, I, _lib.BF16)
    colsum = -(-(G * (-(-R // 96) + 1) * I * 4) // 256) * 256   # fp32 partials of db1, one slot per wave row block
    part = 0 if want_split == 1 else want_split * E_l * 2 * M * I * 4
    if G == E_l and 2 <= G <= 8:  # room for the dynamic slices: about two rounds of tiles, at most kslots + E_l slices
        slots = min(16, max(E_l, 2 * 256 // (2 * (M // 256) * (I // 256)) - (E_l - 1)))
        part = max(part, (slots + E_l) * 2 * M * I * 4)
    assert nbytes == colsum + part
    test_grouped_ffn_fwd_bwd(torch.bfloat16, G, E_l, R, M, I, use_rows)


def test_ffn_shared_gpu_launch_policy_gives_identical_results():
    """flags = EVE_FFN_SHARED_GPU: one workgroup per tile instead of persistent workgroups (the pipelined
    expert-parallel path runs RCCL beside the GEMMs) -- same arithmetic per tile, bit-equal results.  A per-call
    argument, so calls with and without it may be interleaved freely."""
    from eve_amd import ops, _lib
    G, E_l, R, M, I = 32, 16, 300, 768, 3072  # 768 / 192 forward tiles: more than the 256 CUs, so the default is persistent
    g = torch.Generator().manual_seed(5)
    mk = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(torch.bfloat16).to(DEV)
    x, dy = mk(G, R, M), mk(G, R, M)
    w1, b1, w2, b2 = mk(E_l, I, M, s=M ** -0.5), mk(E_l, I, s=0.1), mk(E_l, M, I, s=I ** -0.5), mk(E_l, M, s=0.1)
    rows = torch.randint(100, R + 1, (G,), generator=g, dtype=torch.int32).to(DEV)

    def run(flag):
        z, a, y = ops.ffn_fwd(x, w1, b1, w2, b2, rows, ops.ACTS["quick_gelu"], shared_gpu=flag)
        return (y, a) + tuple(ops.ffn_bwd(dy, x, z, a, w1, w2, rows, ops.ACTS["quick_gelu"], shared_gpu=flag))

    base = run(False)
    shared = run(True)
    again = run(False)
    assert all(torch.equal(p, q) for p, q in zip(base[3:], again[3:]))  # parameter gradients (all rows written)
    valid = (torch.arange(R, device=DEV)[None, :, None] < rows[:, None, None])
    for k, (p, q) in enumerate(zip(base, shared)):
        if p.dim() == 3 and p.shape[0] == G:  # rows beyond rows[g] are left unwritten
            p, q = torch.where(valid, p, 0), torch.where(valid, q, 0)
        assert torch.equal(p, q), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,n_vis,M,E,mode", [(3, 37, 5, 64, 8, "vl"), (4, 237, 197, 768, 32, "vl"),
                                                  (2, 33, 0, 256, 32, "text"), (2, 50, 7, 128, 5, "vision"),
                                                  (1, 31, 0, 72, 3, "none")])
def test_route_bwd_matches_two_pass(dtype, B, T, n_vis, M, E, mode):
    """eve_route_bwd (fused, fp32-MFMA gate part) vs eve_dispatch_bwd + eve_embed_grad, and vs fp64 torch."""
    from eve_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    S, C = B * T, max(4, (2 * B * T) // E)
    slot = torch.full((S, 2), -1, dtype=torch.int32)
    perm = torch.randperm(E * C, generator=g)[: 2 * S].to(torch.int32)
    n_fill = min(2 * S, E * C)
    flat = slot.view(-1)
    pos = torch.randperm(2 * S, generator=g)[: int(n_fill * 0.9)]
    flat[pos] = perm[: pos.numel()]
    dxd = torch.randn(E * C, M, generator=g).to(dtype)
    dlogits = torch.randn(S, E, generator=g) * 0.1
    wg = torch.randn(E, M, generator=g) * 0.05
    mcode = ops.MODES[mode]
    args = [t.to(dev) for t in (dxd, slot, dlogits, wg)]
    dh, dv, dt = ops.route_bwd(*args, S, T, n_vis, mcode, True, True)
    dh_ref = ops.dispatch_bwd(*args, S)
    torch.cuda.synchronize()
    # fp64 reference of dh
    want = dlogits.double() @ wg.double()
    for k in range(2):
        sl = slot[:, k].long()
        want += torch.where((sl >= 0)[:, None], dxd.double()[sl.clamp(min=0)], torch.zeros(1, dtype=torch.float64))
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dh.cpu().double(), want, **tol)
    torch.testing.assert_close(dh.cpu().double(), dh_ref.cpu().double(), **tol)
    if mode == "none":
        assert dv is None and dt is None
        return
    # embedding gradients = column sums of the ROUNDED dh by modality (what eve_embed_grad computes from dh)
    dv_ref, dt_ref = ops.embed_grad(dh, T, n_vis, mcode, True, True)
    torch.cuda.synchronize()
    torch.testing.assert_close(dv, dv_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dt, dt_ref, rtol=1e-5, atol=1e-5)
    # bf16 embedding gradients: the same fp32 sums, rounded once
    _, dv16, dt16 = ops.route_bwd(*args, S, T, n_vis, mcode, True, True, torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(dv16, dv.to(torch.bfloat16)) and torch.equal(dt16, dt.to(torch.bfloat16))
    is_text = torch.tensor([orc_is_text(s, T, n_vis, mode) for s in range(S)])
    d64 = dh.cpu().double()
    torch.testing.assert_close(dv.cpu().double(), d64[~is_text].sum(0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dt.cpu().double(), d64[is_text].sum(0), rtol=1e-4, atol=1e-4)


def orc_is_text(s, T, n_vis, mode):
    if mode == "vision":
        return False
    if mode == "text":
        return True
    return (s % T) >= n_vis
