"""One transform split over ranks, rehearsed on CPU with gloo (world_size 2 and 4).

ntt_amd.distributed.ShardedTransform does the layouts and the all-to-all; the local butterfly
stages, which on a GPU box are libnttx launches, are done here by a small numpy engine (this
file is a test, so a CPU stand-in is allowed).  The result must equal the CPU oracle's transform
of the whole vector, slab by slab.  The modulus is the reference's 32-bit-compatible prime
(ntt.rs:66-68) so that products fit numpy's uint64.
"""
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

Q32 = 4293918721
PSI32_2_17 = 2004365341  # the reference's 2^17-th root for Q32 (ntt.rs:66-68)


def brev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


class NumpyEngine:
    """Plain-residue butterflies of the reference (ntt.rs:129-171 forward, :185-235 backward) on one
    shard's buffer, in the layouts of include/nttx.h."""

    def __init__(self, q, psi, n):
        self.q, self.n = np.uint64(q), n
        self.logn = n.bit_length() - 1
        psi_inv = pow(psi, q - 2, q)
        self.w = np.array([pow(psi, brev(i, self.logn), q) for i in range(n)], dtype=np.uint64)
        self.winv = np.array([pow(psi_inv, brev(i, self.logn), q) for i in range(n)], dtype=np.uint64)
        self.n_inv = np.uint64(pow(n, q - 2, q))

    def geometry(self, world):
        from ntt_amd.distributed import split_geometry
        return split_geometry(self.n, world)

    def _stage(self, a, u, first_block, forward):
        """Stage u on array a of shape [blocks, 2, half, cols]; block b uses twiddle 2^(u-1) + first_block + b."""
        q = self.q
        nb = a.shape[0]
        idx = (1 << (u - 1)) + first_block + np.arange(nb)
        s = (self.w if forward else self.winv)[idx].reshape(nb, 1, 1)
        U, V = a[:, 0].copy(), a[:, 1].copy()
        if forward:
            V = V * s % q
            a[:, 0] = (U + V) % q
            a[:, 1] = (U + q - V) % q
        else:
            a[:, 0] = (U + V) % q
            a[:, 1] = (U + q - V) % q * s % q

    def phase(self, buf, direction, phase, shard, world):
        local, rows, cols, _ = self.geometry(world)
        f = rows.bit_length() - 1
        a = buf.numpy().view(np.uint64)
        forward = direction == 0
        if phase == 0:   # column layout [rows][cols]: stages 1..f pair rows
            us = range(1, f + 1) if forward else range(f, 0, -1)
            for u in us:
                nb = 1 << (u - 1)
                self._stage(a.reshape(nb, 2, rows // (2 * nb), cols), u, 0, forward)
            if not forward:
                a[:] = a * self.n_inv % self.q
        else:            # block layout: stages f+1..log2 n inside this shard's contiguous slab
            us = range(f + 1, self.logn + 1) if forward else range(self.logn, f, -1)
            for u in us:
                nb_global = 1 << (u - 1)
                nb = nb_global // world
                self._stage(a.reshape(nb, 2, local // (2 * nb), 1), u, shard * nb, forward)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, logn, out_dir, max_elems_per_call=0, direct_rows=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle as orc
    from ntt_amd.distributed import ShardedTransform, column_slab

    if max_elems_per_call:   # force the exchange into several collective calls
        ShardedTransform.MAX_ELEMS_PER_CALL = max_elems_per_call
    # the grouped point-to-point exchange (rows land in place) or one collective plus a reordering copy
    ShardedTransform.MAX_ROWS_DIRECT = direct_rows

    n = 1 << logn
    psi = pow(PSI32_2_17, (1 << 16) // n, Q32)
    x = orc.splitmix_residues(Q32, n, 77)             # every rank derives the same whole vector
    st = ShardedTransform(None, engine=NumpyEngine(Q32, psi, n))
    col = torch.from_numpy(column_slab(x, rank, world, st.rows).view(np.int64).copy())
    blk = st.forward(col.clone())
    back = st.backward(blk.clone())
    np.save(os.path.join(out_dir, "fwd%d.npy" % rank), blk.numpy().view(np.uint64))
    np.save(os.path.join(out_dir, "col%d.npy" % rank), back.numpy().view(np.uint64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,logn,max_elems_per_call,direct_rows",
                         [(2, 14, 0, 0), (2, 16, 0, 0), (4, 14, 0, 0), (4, 16, 0, 0), (2, 14, 1024, 0), (4, 16, 1000, 0),
                          (2, 14, 0, 256), (4, 14, 0, 256), (4, 16, 0, 256), (2, 16, 0, 256)])
def test_split_transform_matches_oracle(tmp_path, world, logn, max_elems_per_call, direct_rows):
    import _oracle as orc
    from ntt_amd.distributed import column_slab, split_geometry

    n = 1 << logn
    psi = pow(PSI32_2_17, (1 << 16) // n, Q32)
    x = orc.splitmix_residues(Q32, n, 77)
    want = x.copy()
    orc.Table.from_prime_and_root(Q32, psi, n).forward_inplace(want)
    mp.spawn(_worker, args=(world, _free_port(), logn, str(tmp_path), max_elems_per_call, direct_rows), nprocs=world,
             join=True)
    local, rows, cols, rpp = split_geometry(n, world)
    assert local * world == n and rows * cols * world == n and rpp * world == rows
    for r in range(world):
        got = np.load(tmp_path / ("fwd%d.npy" % r))
        assert np.array_equal(got, want[r * local:(r + 1) * local]), "forward slab of rank %d" % r
        back = np.load(tmp_path / ("col%d.npy" % r))
        assert np.array_equal(back, column_slab(x, r, world, rows)), "round trip slab of rank %d" % r


def test_split_geometry_limits():
    from ntt_amd.distributed import split_geometry

    assert split_geometry(1 << 28, 8) == (1 << 25, 16, 1 << 21, 2)      # the N = 2^28 / 8-GPU case: f = 4
    assert split_geometry(1 << 16, 8) == (1 << 13, 256, 32, 32)         # f = 8
    assert split_geometry(1 << 20, 1) == (1 << 20, 16, 1 << 16, 16)
    for n, world in ((1 << 20, 32), (1 << 12, 2), (1 << 15, 2), (1 << 16, 3), (1 << 8, 1)):
        with pytest.raises(ValueError):
            split_geometry(n, world)
