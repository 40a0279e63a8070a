"""Rank body of test_gpu_distributed.test_one_process_per_shard (launched by torch.distributed.run).

Every rank derives the same whole vector, transforms its own shard on GPU 0 with
ntt_amd.distributed.ShardedTransform and compares its slab with the CPU oracle's transform of the
whole vector.  argv: kind log2(n) out_dir.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

Q61 = 0x1FFFFFFFFFE00001
PSI61_2_21 = 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001


def main():
    kind, logn, out_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    import _oracle as orc
    import ntt_amd
    from ntt_amd.distributed import ShardedTransform, column_slab

    n = 1 << logn
    if kind == "mont":
        q, psi = Q61, pow(PSI61_2_21, (1 << 20) // n, Q61)
        gpu, ref = ntt_amd.Table.from_prime_and_root(q, psi, n), orc.Table.from_prime_and_root(q, psi, n)
    else:
        q, psi = QGL, pow(7, (QGL - 1) // (2 * n), QGL)
        gpu, ref = ntt_amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n)
    x = orc.splitmix_residues(q, n, 900 + logn)
    want = x.copy()
    ref.forward_inplace(want)

    st = ShardedTransform(gpu)
    mine = column_slab(x, rank, world, st.rows)
    col = torch.from_numpy(mine.view(np.int64).copy()).cuda()
    blk = st.forward(col)
    torch.cuda.synchronize()
    got = blk.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want[rank * st.local:(rank + 1) * st.local]), "forward slab of rank %d" % rank
    back = st.backward(blk)
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy().view(np.uint64), mine), "round trip slab of rank %d" % rank
    dist.barrier()
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
