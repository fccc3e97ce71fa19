"""The N > 1 path on CPU: two gloo ranks deal a synthetic corpus, 'process' their shards, and rank 0 merges.
(The GPU benchmark uses the same dealing and the same barrier + max-over-ranks timing with backend nccl = RCCL.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vits_amd import sharding, synthetic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batches = synthetic.corpus_c4(n_utts=300, max_batch=16)
    mine = sharding.deal_batches(batches, world, rank)
    dist.barrier()
    frames = sum(sum(batches[i][1]) for i in mine)
    seconds = torch.tensor([0.5 + rank], dtype=torch.float64)          # pretend rank 1 is slower
    dist.all_reduce(seconds, op=dist.ReduceOp.MAX)
    rep = dict(audio_seconds=frames * 512 / 44100.0, seconds=0.5 + rank, utterances=sum(len(batches[i][0]) for i in mine),
               indices=mine, max_seconds=float(seconds))
    gathered = [None] * world
    dist.all_gather_object(gathered, rep)
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dealing_and_merge(tmp_path):
    out = str(tmp_path / "reports.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    reports = torch.load(out, weights_only=False)
    batches = synthetic.corpus_c4(n_utts=300, max_batch=16)
    seen = sorted(i for r in reports for i in r["indices"])
    assert seen == list(range(len(batches)))                      # every batch exactly once
    assert all(r["max_seconds"] == 1.5 for r in reports)           # max over ranks reached everyone
    merged = sharding.merge_reports(reports)
    assert merged["utterances"] == 300 and merged["seconds"] == 1.5
    total_audio = sum(sum(b[1]) for b in batches) * 512 / 44100.0
    assert abs(merged["audio_seconds"] - total_audio) < 1e-6
    loads = [sum(sharding.batch_cost(batches[i]) for i in r["indices"]) for r in reports]
    assert abs(loads[0] - loads[1]) / max(loads) < 0.1             # balanced within 10 %


def test_dealing_is_rank_count_independent_in_composition():
    batches = synthetic.corpus_c4(n_utts=200, max_batch=16)
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in sharding.deal_batches(batches, world, r))
        assert got == list(range(len(batches)))
