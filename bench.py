#!/usr/bin/env python3
"""Headline benchmark: batched N=2^16 forward+inverse NTT pairs per second on MI355X.

One "step" = forward_inplace followed by backward_inplace (the reference's README loop,
src/main.rs:28-31) over one device-resident batch of 4096 vectors of 2^16 residues modulo
0x1fffffffffe00001 per GPU (BASELINE.json configs[1]).  Inputs are synthetic uniform residues
generated on the device before the timed region.  With --gpus N (launched by torch.distributed.run,
one rank per GPU) every rank transforms its own batch; no data-path collective exists.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against HBM bandwidth
using algorithmic bytes; `cpu_baseline` times the CPU oracle (the C restatement of the reference)
on this host's cores for a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

Q61 = 0x1FFFFFFFFFE00001
PSI61 = 0x15EB043C7AA2B01F
LOGN = 16
N = 1 << LOGN
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, GB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="transforms per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=10, help="event-timed launches per kernel for the roofline")
    args = ap.parse_args()

    import numpy as np
    import torch

    import ntt_amd
    from ntt_amd._lib import lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    batch = args.batch
    table = ntt_amd.Table.from_prime_and_root(Q61, PSI61, N, devices=[local_rank])

    # synthetic uniform residues in [0, q), generated on the device
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0000 + rank)
    data = torch.randint(0, Q61, (batch * N,), dtype=torch.int64, device=dev, generator=gen)
    original = data.clone()

    # correctness outside the timed region: sampled transforms against the oracle, full round trip
    import _oracle as orc
    oracle = orc.Table.new()
    table.forward_inplace(data)
    torch.cuda.synchronize()
    for b in sorted({0, batch // 2, batch - 1}):
        want = original[b * N:(b + 1) * N].cpu().numpy().view(np.uint64).copy()
        oracle.forward_inplace(want)
        got = data[b * N:(b + 1) * N].cpu().numpy().view(np.uint64)
        if not np.array_equal(got, want):
            raise SystemExit("parity failure on transform %d" % b)
    table.backward_inplace(data)
    torch.cuda.synchronize()
    if not torch.equal(data, original):
        raise SystemExit("round trip failure")

    def step():
        table.forward_inplace(data)
        table.backward_inplace(data)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not torch.equal(data, original):
        raise SystemExit("round trip failure after the timed region")

    pairs = batch * world * args.steps
    value = pairs / elapsed
    frac_job = value * 32 * N / (world * HBM_PEAK_GBS * 1e9)

    # ---- roofline of the dominant kernel: HIP events on the launch stream, per pass -----------
    stream = torch.cuda.current_stream(dev)
    npass = lib.nttx_pass_count(table._plan)
    kernels = []
    for direction in (0, 1):
        for pi in range(npass):
            buf = C.create_string_buffer(96)
            lib.nttx_pass_describe(table._plan, direction, pi, buf, 96)
            evs = []
            for r in range(args.kernel_reps + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                rc = lib.nttx_run_pass_device(table._plan, local_rank, data.data_ptr(), batch, direction, pi,
                                              stream.cuda_stream)
                e1.record(stream)
                if rc != 0:
                    raise SystemExit("nttx_run_pass_device failed")
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in evs[2:]]
            kernels.append({"name": buf.value.decode(), "avg_ms": sum(ms) / len(ms)})
    dom = max(kernels, key=lambda k: k["avg_ms"])
    stages_per_pass = LOGN // npass
    # algorithmic bytes: 16*N per transform (8N read + 8N written once), apportioned to a launch
    # by the share of the log2(N) stages it executes
    alg_bytes = 16 * N * batch * stages_per_pass / LOGN
    achieved = alg_bytes / (dom["avg_ms"] * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom["avg_ms"], 4),
        "whole_job_frac": round(frac_job, 4),
        "kernels": [{"name": k["name"], "avg_ms": round(k["avg_ms"], 4)} for k in kernels],
    }

    # ---- CPU baseline: the oracle on this host's cores, bounded sample ------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
        per_thread = 150
        secs = orc.bench_pairs(0, oracle, N, per_thread, cores)
        single = {}
        for kind, name, tab in ((0, "base", oracle), (1, "avx2", orc.TableAVX2.new()),
                                (2, "goldilock", orc.TableGoldilock.new())):
            single[name + "_ms_30_pairs_1_thread"] = round(orc.bench_pairs(kind, tab, N, 30, 1) * 1e3, 1)
        cpu = {
            "value": round(cores * per_thread / secs, 1), "unit": "fwd+inv NTT pairs/s", "cores": cores,
            "kind": "port",
            "sample": "%d threads x %d pairs, N=2^16, q=0x1fffffffffe00001, scalar Montgomery Table (C oracle)"
                      % (cores, per_thread),
            "readme_shape": single,
        }

    if rank == 0:
        out = {
            "metric": "batched N=2^16 fwd+inv NTTs/sec per node; % of HBM roofline",
            "value": round(value, 1), "unit": "fwd+inv NTT pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "batch %d x N=2^16 Montgomery 61-bit fwd+inv per GPU (BASELINE configs[1])" % batch,
                       "batch_per_gpu": batch, "n": N, "modulus": hex(Q61), "sharding": "batch split, no collective"},
            "hbm_roofline_frac": round(frac_job, 4),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
