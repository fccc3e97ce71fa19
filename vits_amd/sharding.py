"""Utterance-level data parallelism: deal whole batches to ranks.  No collective is needed on the data path --
utterances are independent and every GPU holds the full (198 MB) weight set -- so a rank only needs to know which
batches are its own.  Batch composition is fixed before dealing (padding changes numerics in the reference, see
models.py:756-765), so results do not depend on the number of ranks."""
from __future__ import annotations

from typing import List, Sequence, Tuple

Batch = Tuple[Sequence[int], Sequence[int]]      # (phoneme counts, frame counts)


def batch_cost(batch: Batch) -> int:
    """Algorithmic FLOPs of one batch (2*MAC; per frame 837 861 376, per phoneme 25 069 056, plus attention)."""
    ps, ts = batch
    return sum(837861376 * t + 3072 * t * t + 25069056 * p + 7680 * p * p for p, t in zip(ps, ts))


def deal_batches(batches: List[Batch], world: int, rank: int) -> List[int]:
    """Indices of the batches rank `rank` processes: longest-processing-time-first greedy assignment, which every
    rank computes identically from the batch list alone."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    order = sorted(range(len(batches)), key=lambda i: (-batch_cost(batches[i]), i))
    load = [0] * world
    mine: List[int] = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += batch_cost(batches[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def merge_reports(reports: List[dict]) -> dict:
    """Host-side merge of per-rank {audio_seconds, seconds, utterances}: aggregate = sum(audio) / max(seconds)."""
    audio = sum(r["audio_seconds"] for r in reports)
    wall = max(r["seconds"] for r in reports)
    return dict(audio_seconds=audio, seconds=wall, utterances=sum(r["utterances"] for r in reports),
                audio_s_per_s=audio / wall if wall > 0 else float("nan"))
