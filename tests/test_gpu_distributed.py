"""GPU parity of ONE transform split over several shards (include/nttx.h, "distributed").

The reference computes a transform of any length in one loop nest (ntt.rs:122-176, :178-236;
ntt_goldilock.rs:81-180); the split must give the same bits.  On the one-GPU test box all shards
share device 0: nttx_forward_distributed places shard i on plan device i % n_devices, so the
kernels, the shard twiddle offsets and the exchange run exactly as they do with one GPU per
shard; only the copies stay on one device.  The one-process-per-GPU flavour (ShardedTransform
over torch.distributed) is run with 2 and 4 ranks sharing the GPU and gloo carrying the pieces.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
Q61 = 0x1FFFFFFFFFE00001
PSI61_2_21 = 0x0B4F9CB5C38B40B1
QGL = 0xFFFFFFFF00000001


def tables(amd, kind, n):
    if kind == "mont":
        psi = pow(PSI61_2_21, (1 << 20) // n, Q61)
        return (amd.Table.from_prime_and_root(Q61, psi, n), orc.Table.from_prime_and_root(Q61, psi, n), Q61)
    psi = pow(7, (QGL - 1) // (2 * n), QGL)
    return amd.TableGoldilock.from_root(psi, n), orc.TableGoldilock.from_root(psi, n), QGL


@pytest.fixture(scope="module")
def amd():
    import ntt_amd
    return ntt_amd


@pytest.mark.parametrize("kind", ["mont", "goldilocks"])
@pytest.mark.parametrize("logn", [16, 18, 20])
def test_virtual_shards_match_oracle(amd, kind, logn):
    n = 1 << logn
    gpu, ref, q = tables(amd, kind, n)
    x = orc.splitmix_residues(q, n, 500 + logn)
    want = x.copy()
    ref.forward_inplace(want)
    f = logn % 8 or 8
    for shards in (1, 2, 4, 8, 16):
        if shards > (1 << f) or logn < f + (shards.bit_length() - 1) + 4:
            continue
        assert gpu.shard_geometry(shards) == amd.distributed.split_geometry(n, shards)
        a = x.copy()
        gpu.forward_inplace_distributed(a, shards)
        assert np.array_equal(a, want), "forward, %d shards" % shards
        gpu.backward_inplace_distributed(a, shards)
        assert np.array_equal(a, x), "backward, %d shards" % shards


def test_goldilocks_out_of_range_input_split(amd):
    """TableGoldilock takes any u64 (no input sweep); the split must follow the same rules."""
    n = 1 << 16
    gpu, ref, _ = tables(amd, "goldilocks", n)
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 2**64, n, dtype=np.uint64)
    raw[:4] = [2**64 - 1, QGL, QGL + 1, QGL - 1]
    for fn in ("forward_inplace", "backward_inplace"):
        a, w = raw.copy(), raw.copy()
        getattr(gpu, fn + "_distributed")(a, 8)
        getattr(ref, fn)(w)
        assert np.array_equal(a, w), fn


def test_split_limits(amd):
    gpu, _, _ = tables(amd, "goldilocks", 1 << 20)            # first pass has 4 stages
    a = np.zeros(1 << 20, dtype=np.uint64)
    for bad in (3, 32, 128, -1):
        with pytest.raises(amd.Panic):
            gpu.forward_inplace_distributed(a, bad)
    small, _, _ = tables(amd, "goldilocks", 1 << 8)           # single-pass transform: nothing to split
    with pytest.raises(amd.Panic):
        small.forward_inplace_distributed(np.zeros(1 << 8, dtype=np.uint64), 2)
    with pytest.raises(amd.Panic):
        gpu.forward_inplace_distributed(np.zeros(1 << 16, dtype=np.uint64), 2)   # wrong length


def test_two_to_the_24_eight_shards(amd):
    """First pass with 8 stages (32 rows per peer), three passes in all."""
    n = 1 << 24
    gpu, ref, q = tables(amd, "goldilocks", n)
    x = orc.splitmix_residues(q, n, 524)
    want = x.copy()
    ref.forward_inplace(want)
    a = x.copy()
    gpu.forward_inplace_distributed(a, 8)
    assert np.array_equal(a, want)
    gpu.backward_inplace_distributed(a, 8)
    assert np.array_equal(a, x)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,kind,logn", [(2, "mont", 16), (4, "goldilocks", 20), (8, "goldilocks", 20), (8, "mont", 16)])
def test_one_process_per_shard(tmp_path, world, kind, logn):
    """ShardedTransform over torch.distributed; the ranks share GPU 0, so gloo carries the pieces.
    Eight ranks at 2^20 have BASELINE configs[4]'s exchange shape (first pass of 4 stages, two rows per
    peer, the grouped point-to-point form); eight at 2^16 have 32 rows per peer (the all-to-all form)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_gpu_worker.py"), kind, str(logn), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    for r in range(world):
        assert (tmp_path / ("ok%d" % r)).exists(), res.stdout[-2000:] + res.stderr[-2000:]


def test_two_to_the_28_goldilocks(amd):
    """BASELINE configs[4]'s vector: N = 2^28, Goldilocks, psi = 7^((q-1)/2^29), split over 8 shards
    (first pass 4 stages, 2 rows per peer, 16 MiB pieces).  Checked against the oracle's transform
    of the whole vector, against the unsplit GPU transform, and by the round trip."""
    import time
    n = 1 << 28
    psi = pow(7, (QGL - 1) // (2 * n), QGL)
    assert psi == 0xDFA8C93BA46D2666
    t0 = time.time()
    gpu = amd.TableGoldilock.from_root(psi, n)
    assert gpu.n_inv == 0xFFFFFFEF00000011
    assert gpu.shard_geometry(8) == (1 << 25, 16, 1 << 21, 2)
    t1 = time.time()
    x = orc.splitmix_residues(QGL, n, 528)
    split = x.copy()
    gpu.forward_inplace_distributed(split, 8)
    whole = x.copy()
    gpu.forward_inplace(whole)
    assert np.array_equal(split, whole), "split and unsplit GPU transforms differ"
    del whole
    t2 = time.time()
    ref = orc.TableGoldilock.from_root(psi, n)
    want = x.copy()
    ref.forward_inplace(want)
    del ref
    t3 = time.time()
    assert np.array_equal(split, want), "GPU transform differs from the oracle"
    del want
    back = split.copy()
    gpu.backward_inplace(back)   # unsplit: three launches, the last stages fused
    assert np.array_equal(back, x), "unsplit round trip"
    del back
    gpu.backward_inplace_distributed(split, 8)
    assert np.array_equal(split, x), "round trip"
    print("2^28: plan %.1fs, gpu %.1fs, oracle %.1fs, backward %.1fs" % (t1 - t0, t2 - t1, t3 - t2, time.time() - t3))


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_gpus() < 2, reason="needs at least two GPUs")
@pytest.mark.parametrize("peer_access", [True, False])
def test_plan_over_all_devices(peer_access):
    """On a multi-GPU host: one plan over every visible device; a host batch split over them
    (nttx_*_host, one host thread per device), one transform split with one shard per device (peer
    copies between distinct devices), with and without direct peer access (NTTX_NO_PEER_ACCESS is read
    once per process, so each variant runs in a child)."""
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import _oracle as orc, ntt_amd
nd = torch.cuda.device_count()
shards = 1 << (nd.bit_length() - 1)
devs = list(range(shards))
QGL = 0xFFFFFFFF00000001
n = 1 << 20
psi = pow(7, (QGL - 1) // (2 * n), QGL)
gpu, ref = ntt_amd.TableGoldilock.from_root(psi, n, devices=devs), orc.TableGoldilock.from_root(psi, n)
x = orc.splitmix_residues(QGL, n, 77)
want = x.copy(); ref.forward_inplace(want)
a = x.copy(); gpu.forward_inplace_distributed(a, 0)
assert np.array_equal(a, want), "split over devices"
gpu.backward_inplace_distributed(a, 0)
assert np.array_equal(a, x), "split round trip"
Q61, R61 = 0x1FFFFFFFFFE00001, 0x0B4F9CB5C38B40B1
n = 1 << 16
t, r = ntt_amd.Table.from_prime_and_root(Q61, pow(R61, 16, Q61), n, devices=devs), orc.Table.new()
batch = 8 * shards + 3
b = orc.splitmix_residues(Q61, n * batch, 78)
w = b.copy()
for i in range(batch): r.forward_inplace(w[i * n:(i + 1) * n])
t.forward_inplace_batch(b, batch)
assert np.array_equal(b, w), "host batch over devices"
print("ok", shards)
""" % (os.path.dirname(HERE), HERE)
    env = dict(os.environ)
    if not peer_access:
        env["NTTX_NO_PEER_ACCESS"] = "1"
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
