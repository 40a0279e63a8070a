"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): contiguous batch sharding with no
data-path collective, barrier-bracketed timing, max over ranks.  The per-rank transform is the
CPU oracle standing in for the GPU engine (this file is a test, so it may)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from ntt_amd.sharding import job_throughput, max_over_ranks, shard_range  # noqa: E402

Q61 = 0x1FFFFFFFFFE00001
N = 1 << 10
BATCH = 7  # ragged on purpose: ranks get 4 and 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle as orc

    psi = pow(0x0B4F9CB5C38B40B1, (1 << 20) // N, Q61)
    table = orc.Table.from_prime_and_root(Q61, psi, N)
    data = orc.splitmix_residues(Q61, BATCH * N, 42)          # every rank derives the same global batch
    first, count = shard_range(BATCH, world, rank)
    mine = data[first * N:(first + count) * N].copy()
    dist.barrier()
    for b in range(count):
        table.forward_inplace(mine[b * N:(b + 1) * N])
    dist.barrier()
    elapsed = max_over_ranks(0.5 + rank, dist)                 # fake clocks: rank 1 is the slowest
    np.save(os.path.join(out_dir, "shard%d.npy" % rank), mine)
    if rank == 0:
        np.save(os.path.join(out_dir, "elapsed.npy"), np.array([elapsed]))
    dist.destroy_process_group()


def test_shard_range_covers_batch_exactly():
    for batch in (0, 1, 7, 8, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == batch
            pos = 0
            for first, count in spans:
                assert first == pos or count == 0
                pos += count
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_sharded_forward_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import _oracle as orc

    psi = pow(0x0B4F9CB5C38B40B1, (1 << 20) // N, Q61)
    table = orc.Table.from_prime_and_root(Q61, psi, N)
    want = orc.splitmix_residues(Q61, BATCH * N, 42)
    for b in range(BATCH):
        table.forward_inplace(want[b * N:(b + 1) * N])
    got = np.concatenate([np.load(str(tmp_path / ("shard%d.npy" % r))) for r in range(world)])
    assert np.array_equal(got, want)
    assert float(np.load(str(tmp_path / "elapsed.npy"))[0]) == 1.5       # max over ranks
    assert job_throughput(4096, 2, 10, 2.0) == 40960.0
