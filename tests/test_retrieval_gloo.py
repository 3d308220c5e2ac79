"""Rank-sliced rerank loop + score-matrix merge (eve_amd/retrieval.py) on CPU over gloo.

Expected values: the reference's own procedure restated inline (Retrieval.py:146-196) -- every rank fills ITS rows
of a full matrix pre-filled with -100 and the matrices are combined with a SUM all-reduce -- run in the same
processes with real collectives.  World 3 makes the row split uneven (10 rows -> 4, 3, 3)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _score(offset):
    """A stand-in for backbone + ITM head: a deterministic score per (row, candidate)."""
    def fn(row, idx):
        return torch.sin(idx.double() * 0.37 + row * 1.3 + offset).float() * 3.0
    return fn


def _reference(sims, k, score, world, rank):
    """Retrieval.py:149-170 / :172-191 for one direction, then :193-196."""
    full = torch.full(sims.shape, -100.0)
    step, rest = sims.size(0) // world, sims.size(0) % world
    steps = [step + 1 if i < rest else step for i in range(world)]
    start = sum(steps[:rank])
    end = start + steps[rank]
    for i, row in enumerate(sims[start:end]):
        _, topk_idx = row.topk(k=k, dim=0)
        full[start + i, topk_idx] = score(start + i, topk_idx)
    dist.barrier()
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(1)
        from eve_amd import retrieval
        g = torch.Generator().manual_seed(5)
        sims = torch.randn(10, 23, generator=g)  # [images, texts], identical on every rank
        k = 6
        want_i2t = _reference(sims, k, _score(0.0), world, rank)
        want_t2i = _reference(sims.t().contiguous(), k, _score(0.5), world, rank)
        got_i2t, got_t2i = retrieval.rerank_evaluation(sims, k, _score(0.0), _score(0.5))
        torch.testing.assert_close(got_i2t, want_i2t, rtol=1e-6, atol=1e-4)
        torch.testing.assert_close(got_t2i, want_t2i, rtol=1e-6, atol=1e-4)
        # rankings (what itm_eval reads) are the reference's exactly
        assert torch.equal(got_i2t.argsort(dim=1, descending=True, stable=True),
                           want_i2t.argsort(dim=1, descending=True, stable=True))
        # main-process-only variant: rank 1 gets the matrices, the others nothing
        one = retrieval.rerank_evaluation(sims, k, _score(0.0), _score(0.5), dst=1)
        if rank == 1:
            torch.testing.assert_close(one[0], want_i2t, rtol=1e-6, atol=1e-4)
            torch.testing.assert_close(one[1], want_t2i, rtol=1e-6, atol=1e-4)
        else:
            assert one == (None, None)
        # un-shifted scores: the owner's block as it is
        raw, _ = retrieval.rerank_evaluation(sims, k, _score(0.0), _score(0.5), reference_offset=False)
        torch.testing.assert_close(raw, want_i2t + 100.0 * (world - 1), rtol=1e-6, atol=1e-4)
        for n in (10, 7, 3, 1):
            spans = [retrieval.rank_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_rerank_and_score_matrix_merge_match_the_reference_procedure(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(m == "ok" for _, m in res), res


def test_single_process_is_the_plain_loop():
    sys.path.insert(0, REPO)
    from eve_amd import retrieval
    sims = torch.randn(5, 9, generator=torch.Generator().manual_seed(1))
    i2t, t2i = retrieval.rerank_evaluation(sims, 4, _score(0.0), _score(0.5))
    assert i2t.shape == (5, 9) and t2i.shape == (9, 5)
    assert int((i2t != -100.0).sum()) == 5 * 4 and int((t2i != -100.0).sum()) == 9 * 4
    assert retrieval.rank_rows(10, 3, 0) == (0, 4) and retrieval.rank_rows(10, 3, 2) == (7, 10)
