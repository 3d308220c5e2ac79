"""Rank-sliced rerank loop and score-matrix merge of the retrieval evaluation.

Reference: `Retrieval.py:146-196`.  Every rank re-scores the top-k candidates of ITS slice of the rows of the
similarity matrix with the cross-modal encoder (the config-5 workload: k_test x (577 + 40) tokens per forward through
the MoE layers), writes the scores into a full `[rows, cols]` matrix pre-filled with -100 and the ranks' matrices are
combined with two ring all-reduces (SUM) of the whole matrix -- about 2 x 500 MB for COCO's 5 000 x 25 000 scores,
of which every rank contributed 1/world.

Here a rank keeps only its own row block; the blocks are exchanged once (all-gather, or a gather to one rank when only
the main process evaluates, as `Retrieval.py:330` does): (world-1)/world of the matrix arrives per rank instead of
2 (world-1)/world, and over point-to-point xGMI links every peer's block travels its own link.  The result is the
reference's: a SUM all-reduce of matrices that hold -100 wherever a rank did not score adds -100 (world - 1) to every
entry of a row (the owner's score, or its -100 fill) -- a per-row constant that the ranking metrics of `itm_eval`
(`Retrieval.py:208-246`) do not see, reproduced here arithmetically (`reference_offset=True`) so that the numbers
match the reference's output files.

The scoring callable (backbone + ITM head, `Retrieval.py:165-168, :186-189`) is the caller's: stock modules around
`eve_amd.MoE`; nothing here computes on the CPU on its behalf.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

FILL = -100.0  # Retrieval.py:149, :170


def rank_rows(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, end) of the rows rank `rank` re-scores: `n // world` each, the first `n % world` ranks one more
    (Retrieval.py:151-158)."""
    step, rest = divmod(n_rows, world)
    steps = [step + 1 if i < rest else step for i in range(world)]
    start = sum(steps[:rank])
    return start, start + steps[rank]


@torch.no_grad()
def rerank_block(sims: Tensor, start: int, end: int, k: int,
                 score_row: Callable[[int, Tensor], Tensor]) -> Tensor:
    """This rank's row block `[end - start, cols]` of the score matrix: FILL everywhere except, in row i, the
    columns of the k largest similarities, which hold `score_row(start + i, topk_idx)` ([k] scores; the caller runs
    the cross-modal encoder and the ITM head on the k candidates, Retrieval.py:160-169 / :181-190)."""
    cols = sims.shape[1]
    block = torch.full((end - start, cols), FILL, dtype=torch.float32, device=sims.device)
    for i in range(end - start):
        _, topk_idx = sims[start + i].topk(k=min(k, cols), dim=0)
        block[i, topk_idx] = score_row(start + i, topk_idx).to(torch.float32)
    return block


def _gather_blocks(block: Tensor, counts: List[int], group, dst: Optional[int]) -> Optional[Tensor]:
    """Row blocks of unequal height -> `[sum(counts), cols]` (on every rank, or on `dst` only)."""
    world = len(counts)
    rank = dist.get_rank(group)
    tall = max(counts)
    cols = block.shape[1]
    padded = block
    if block.shape[0] != tall:  # the ranks behind the first `rest` hold one row fewer
        padded = torch.zeros(tall, cols, dtype=block.dtype, device=block.device)
        padded[: block.shape[0]] = block
    padded = padded.contiguous()
    backend = str(dist.get_backend(group)).lower()
    stage = padded.is_cuda and backend == "gloo"  # rehearsal on one GPU: gloo moves host memory
    src = padded.cpu() if stage else padded
    if dst is None:
        out = torch.empty(world * tall, cols, dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        out = torch.empty(world * tall, cols, dtype=src.dtype, device=src.device) if rank == dst else None
        parts = list(out.view(world, tall, cols).unbind(0)) if rank == dst else None
        dist.gather(src, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
    if stage:
        out = out.to(block.device)
    if all(c == tall for c in counts):
        return out
    return torch.cat([out[r * tall: r * tall + c] for r, c in enumerate(counts)], dim=0)


def merge_score_matrix(block: Tensor, n_rows: int, group=None, dst: Optional[int] = None,
                       reference_offset: bool = True) -> Optional[Tensor]:
    """The full `[n_rows, cols]` score matrix from the ranks' row blocks (`rerank_block`), on every rank (`dst=None`,
    what `evaluation` returns, Retrieval.py:193-205) or on rank `dst` of the group only (others get None).

    `reference_offset`: add the `FILL * (world - 1)` that the reference's SUM all-reduce of FILL-initialised
    matrices leaves on every entry (see the module docstring); False gives the un-shifted scores.  The shifted
    matrix agrees with the reference's to fp32 ROUNDING, not bit for bit: the reference adds -100 once per peer in ring
    order, here the whole constant is one fp32 add, and at magnitude 100 * (world - 1) (ulp 6e-5 at 700) the two can
    round differently for world > 2 -- near-tied scores may then rank differently.  The constant is the same for every
    entry of a row, so it changes no ranking metric: prefer `reference_offset=False` (scores at full precision) for
    itm_eval-style evaluation and keep True only where the reference's absolute values are wanted."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert block.shape[0] == n_rows
        return block
    world = dist.get_world_size(group)
    counts = [rank_rows(n_rows, world, r)[1] - rank_rows(n_rows, world, r)[0] for r in range(world)]
    assert block.shape[0] == counts[dist.get_rank(group)], "block height must be this rank's share of the rows"
    full = _gather_blocks(block, counts, group, dst)
    if full is not None and reference_offset:
        full = full + FILL * (world - 1)
    return full


@torch.no_grad()
def rerank_evaluation(sims_i2t: Tensor, k: int, score_i2t: Callable[[int, Tensor], Tensor],
                      score_t2i: Callable[[int, Tensor], Tensor], group=None, dst: Optional[int] = None,
                      reference_offset: bool = True):
    """Both directions of Retrieval.py:146-196: image->text over the rows of `sims_i2t` [images, texts], text->image
    over the rows of its transpose.  Returns (score_matrix_i2t, score_matrix_t2i)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    out = []
    for sims, fn in ((sims_i2t, score_i2t), (sims_i2t.t(), score_t2i)):
        start, end = rank_rows(sims.shape[0], world, rank)
        block = rerank_block(sims, start, end, k, fn)
        out.append(merge_score_matrix(block, sims.shape[0], group, dst, reference_offset))
    return tuple(out)
