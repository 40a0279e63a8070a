#!/usr/bin/env python3
"""Headline benchmark: batched N=2^16 forward+inverse NTT pairs per second on MI355X.

One "step" = forward_inplace followed by backward_inplace (the reference's README loop,
src/main.rs:28-31) over one device-resident batch of 4096 vectors of 2^16 residues modulo
0x1fffffffffe00001 per GPU (BASELINE.json configs[1]).  Inputs are synthetic uniform residues
generated on the device before the timed region.  With --gpus N every rank (one process per GPU)
transforms its own batch; no data-path collective exists.  Started without RANK in the environment
and N > 1, this process launches the N ranks itself (torch.distributed.run as a child, before
anything here touches a GPU) and relays rank 0's JSON line.  --total-batch B fixes the job's batch
and splits it over the ranks instead (strong scaling).

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
QGL = 0xFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, GB/s

# name -> (kind, log2 n, default batch per GPU, description)
WORKLOADS = {
    "mont16": ("mont", 16, 4096, "batch %d x N=2^16 Montgomery 61-bit fwd+inv per GPU (BASELINE configs[1])"),
    "goldilocks20": ("goldilocks", 20, 256, "batch %d x N=2^20 Goldilocks fwd+inv per GPU (BASELINE configs[2])"),
    "goldilocks16": ("goldilocks", 16, 4096, "batch %d x N=2^16 Goldilocks fwd+inv per GPU"),
    "goldilocks28": ("goldilocks", 28, 1, "%d x N=2^28 Goldilocks fwd+inv on one GPU (BASELINE configs[4]'s vector, unsplit)"),
    # ONE vector split over all ranks with one all-to-all per direction (strong scaling); needs
    # torch.distributed.run, one rank is allowed.  --logn picks another even size for rehearsals.
    "goldilocks28split": ("goldilocks", 28, 1, "%d x N=2^28 Goldilocks fwd+inv split over the ranks (BASELINE configs[4])"),
}


def launch_ranks(argv, gpus):
    """Starts `gpus` ranks of this script under torch.distributed.run and relays their stdout (rank 0's
    JSON line).  Runs before any GPU call in this process: a process that has initialised the GPU
    must not be replaced, so the ranks are children and this process only waits for them."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def rank_device(local_rank, allow_shared):
    """Device of a local rank, whether ranks share devices, and the number of distinct devices the job
    runs on.  More ranks than visible GPUs is refused unless --allow-shared-devices is given (a
    rehearsal of the multi-rank path on a smaller box: RCCL refuses two ranks on one device, so
    those runs line ranks up over gloo, and the record reports the distinct devices as n_gpus).
    torch.cuda.device_count() does not initialise the GPU."""
    import torch

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("no GPU visible")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > ndev and not allow_shared:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible; ranks would share devices and the "
                         "record would not measure %d GPUs (--allow-shared-devices rehearses the path anyway)"
                         % (world, ndev, world))
    return local_rank % ndev, world > ndev, min(world, ndev)


def cpu_model():
    """Model string of the host CPU with what /proc/cpuinfo adds to it (some hosts report a bare
    "AMD EPYC"): family / model numbers, clock and the number of logical CPUs."""
    fields = {}
    logical = 0
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if ":" not in line:
                    continue
                k, v = (t.strip() for t in line.split(":", 1))
                if k == "processor":
                    logical += 1
                elif k in ("model name", "cpu family", "model", "cpu MHz") and k not in fields:
                    fields[k] = v
    except OSError:
        return "unknown"
    if "model name" not in fields:
        return "unknown"
    return "%s (family %s model %s, %s MHz, %d logical CPUs on the host)" % (
        fields["model name"], fields.get("cpu family", "?"), fields.get("model", "?"), fields.get("cpu MHz", "?"), logical)


def board_power_sensor(torch, index, root="/sys/class/drm"):
    """hwmon files of the card behind torch device `index`: (power in microwatts, its limit), or None.

    amdgpu exposes the board's power as power1_average or power1_input and the limit the firmware holds
    it to as power1_cap; the card is matched by PCI bus number, or taken when it is the only one."""
    import glob
    cards = []
    for hw in sorted(glob.glob(os.path.join(root, "card*/device/hwmon/hwmon*"))):
        power = next((os.path.join(hw, f) for f in ("power1_average", "power1_input")
                      if os.path.exists(os.path.join(hw, f))), None)
        if power:
            pci = os.path.basename(os.path.realpath(os.path.join(hw, "..", "..")))  # 0000:bb:dd.f
            cards.append((pci, power, os.path.join(hw, "power1_cap")))
    if not cards:
        return None
    if len(cards) == 1:
        return cards[0][1:]
    try:
        bus = torch.cuda.get_device_properties(index).pci_bus_id
    except Exception:
        return None
    for pci, power, cap in cards:
        parts = pci.split(":")
        if len(parts) == 3 and int(parts[1], 16) == bus:
            return power, cap
    return None


def read_number(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return None


def kernel_source_stamp():
    """Hash of the kernel sources: PMC summaries under profiles/ carry the stamp of the sources they
    were measured on, and are only quoted while it matches."""
    import hashlib

    h = hashlib.sha256()
    src = os.path.join(ROOT, "ntt_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hpp", ".hip")):
            with open(os.path.join(src, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def recorded_counter(kind, workload, kernel_name):
    """profiles/<kind>_<workload>.json entry for a kernel, or None (with a warning) when the file is
    missing, was measured on other kernel sources, or lacks the kernel."""
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (kind, workload))
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("_stamp") != kernel_source_stamp():
        sys.stderr.write("bench.py: %s was measured on other kernel sources (stamp %s, now %s): not quoted; "
                         "rerun tools/refresh_profiles.sh\n" % (path, rec.get("_stamp"), kernel_source_stamp()))
        return None
    return rec.get(kernel_name)


def run_split(args, json_fd):
    """One transform split over the ranks: forward = first pass on this rank's column slab, one
    all_to_all_single over RCCL, remaining passes on its contiguous slab; backward mirrors it
    (ntt_amd.distributed.ShardedTransform over nttx_shard_phase_device)."""
    import torch
    import torch.distributed as dist

    import ntt_amd
    from ntt_amd.distributed import ShardedTransform
    from ntt_amd.sharding import max_over_ranks

    rank, world, local_rank = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    local_rank, shared, distinct = rank_device(local_rank, args.allow_shared_devices)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if shared:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    logn = args.logn or 28
    n = 1 << logn
    table = ntt_amd.TableGoldilock.from_root(pow(7, (QGL - 1) // (2 * n), QGL), n, devices=[local_rank])
    st = ShardedTransform(table)

    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0000 + rank)
    hi = torch.randint(0, 2**32 - 1, (st.local,), dtype=torch.int64, device=dev, generator=gen)  # < q
    lo = torch.randint(0, 2**32, (st.local,), dtype=torch.int64, device=dev, generator=gen)
    col = (hi << 32) | lo
    del hi, lo
    original = col.clone()

    def barrier():
        torch.cuda.synchronize()
        if shared:
            dist.barrier()
        else:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    def step(x):
        return st.backward(st.forward(x))

    barrier()  # the process's first barrier pays its set-up here, not in front of the timed region
    col = step(col)
    if not torch.equal(col, original):
        raise SystemExit("round trip failure")
    for _ in range(args.warmup):
        col = step(col)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        col = step(col)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, None if shared else dev)
    if not torch.equal(col, original):
        raise SystemExit("round trip failure after the timed region")

    # local phases and the exchange, event-timed on this rank (rank 0 reports its own)
    stream = torch.cuda.current_stream(dev)

    def timed(fn, reps=5):
        out = []
        for _ in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1))
        return sum(out[1:]) / reps

    npass = -(-logn // 8)  # launches of a split transform: the first pass holds log2 n mod 8 (or 8) stages
    scratch = torch.empty_like(col)
    phases = [
        {"name": "fwd phase 0 (first pass, column layout)", "launches": 1,
         "avg_ms": timed(lambda: st.engine.phase(col, 0, 0, rank, world))},
        {"name": "fwd phase 1 (remaining passes, block layout)", "launches": npass - 1,
         "avg_ms": timed(lambda: st.engine.phase(col, 0, 1, rank, world))},
        {"name": "all_to_all_single of this rank's slab", "launches": 0,
         "avg_ms": timed(lambda: st._all_to_all(scratch, col))},
    ]
    dom = max(phases[:2], key=lambda p: p["avg_ms"] / p["launches"])
    f = st.rows.bit_length() - 1
    stages = f if dom is phases[0] else logn - f
    alg_bytes = 16 * st.local * stages / logn          # this rank's share of 16 n bytes, by stages
    achieved = alg_bytes / (dom["avg_ms"] * 1e-3) / 1e9
    value = args.steps / elapsed
    if rank == 0:
        out = {
            "metric": "single N=2^%d fwd+inv NTTs/sec per node; %% of HBM roofline" % logn,
            "value": round(value, 2), "unit": "fwd+inv NTT pairs/s", "n_gpus": distinct, "ranks": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload][3] % 1 if logn == 28 else "1 x N=2^%d Goldilocks, split" % logn,
                       "n": n, "modulus": hex(QGL), "shards": world, "rows_per_peer": st.rows_per_peer,
                       "sharding": "column slabs, one all-to-all per direction, contiguous slabs"
                                   + (", ranks share devices (rehearsal)" if shared else "")},
            "hbm_roofline_frac": round(value * 32 * n / (distinct * HBM_PEAK_GBS * 1e9), 4),
            "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes / dom["launches"],
                         "avg_launch_ms": round(dom["avg_ms"] / dom["launches"], 4),
                         "kernels": [{"name": p["name"], "avg_ms": round(p["avg_ms"], 4)} for p in phases]},
            "cpu_baseline": None,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mont16")
    ap.add_argument("--batch", type=int, default=0, help="transforms per GPU (0 = the workload's default)")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="transforms of the whole job, split contiguously over the ranks (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-shared-devices", action="store_true",
                    help="let --gpus exceed the visible GPUs (ranks share devices, gloo barrier): a rehearsal, "
                         "reported with n_gpus = distinct devices and \"ranks\" = --gpus")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the round-trip checks (timing runs with the NTTX_DEBUG_* profiling switches, whose "
                         "results are meaningless on purpose); the record is marked invalid")
    ap.add_argument("--no-power", action="store_true", help="skip the one-second board-power reading")
    ap.add_argument("--kernel-reps", type=int, default=10, help="event-timed launches per kernel for the roofline")
    ap.add_argument("--logn", type=int, default=0, help="goldilocks28split only: another (even) log2 n for rehearsals")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "RANK" not in os.environ and (args.gpus > 1 or args.workload == "goldilocks28split"):
        import torch  # device_count() does not initialise the GPU

        if args.gpus > torch.cuda.device_count() and not args.allow_shared_devices:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to share devices between ranks "
                             "(--allow-shared-devices rehearses the multi-rank path and reports the distinct "
                             "devices as n_gpus)\n" % (args.gpus, torch.cuda.device_count()))
            sys.exit(2)
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))

    # stdout carries the JSON line and nothing else: libraries that write to file descriptor 1
    # (RCCL prints a version banner there when its first communicator comes up) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.workload == "goldilocks28split":
        return run_split(args, json_fd)

    import numpy as np
    import torch

    import ntt_amd
    from ntt_amd._lib import lib
    from ntt_amd.sharding import max_over_ranks, shard_range

    kind, LOGN, default_batch, describe = WORKLOADS[args.workload]
    N = 1 << LOGN

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    local_rank, shared, distinct = rank_device(local_rank, args.allow_shared_devices)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            if shared:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    strong = args.total_batch > 0
    if strong:
        if args.total_batch < world:
            raise SystemExit("--total-batch must be at least the number of ranks")
        batch = shard_range(args.total_batch, world, rank)[1]
        job_batch = args.total_batch
    else:
        batch = args.batch or default_batch
        job_batch = batch * world
    if kind == "mont":
        q = Q61
        table = ntt_amd.Table.from_prime_and_root(Q61, PSI61, N, devices=[local_rank])
    else:
        q = QGL
        psi = pow(7, (QGL - 1) // (2 * N), QGL)
        table = ntt_amd.TableGoldilock.from_root(psi, N, devices=[local_rank])

    # synthetic uniform residues in [0, q), generated on the device
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0000 + rank)
    if q < 2**63:
        data = torch.randint(0, q, (batch * N,), dtype=torch.int64, device=dev, generator=gen)
    else:  # uniform in [0, q) for a 64-bit q: 64 random bits, the few values >= q folded back
        hi = torch.randint(0, 2**32, (batch * N,), dtype=torch.int64, device=dev, generator=gen)
        lo = torch.randint(0, 2**32, (batch * N,), dtype=torch.int64, device=dev, generator=gen)
        hi = torch.where(hi == 2**32 - 1, hi - 1, hi)  # below 0xffffffff00000000 < q
        data = (hi << 32) | lo
        del hi, lo
    original = data.clone()

    # correctness outside the timed region: the forward transform changes the data, the backward
    # transform restores every element (the oracle comparison is part of the CPU-baseline leg)
    table.forward_inplace(data)
    torch.cuda.synchronize()
    if torch.equal(data, original) and not args.no_verify:
        raise SystemExit("forward transform left the data unchanged")
    forward_sample = {b: data[b * N:(b + 1) * N].cpu().numpy().view(np.uint64).copy()
                      for b in sorted({0, batch // 2, batch - 1})}
    table.backward_inplace(data)
    torch.cuda.synchronize()
    if not torch.equal(data, original) and not args.no_verify:
        raise SystemExit("round trip failure")

    def step():
        table.forward_inplace(data)
        table.backward_inplace(data)

    # The first collective of a process pays the communicator's set-up (tens of milliseconds with the GPU
    # idle); taken here, the barrier in front of the timed region is a warm one (~30 us) and the chip, which
    # runs this step at its power limit, enters the timed region in the state the warm-up left it in.
    barrier()
    for _ in range(5):  # settle clocks and caches before the counted warm-up, whatever W is
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, None if shared else dev)
    if not torch.equal(data, original) and not args.no_verify:
        raise SystemExit("round trip failure after the timed region")

    value = job_batch * args.steps / elapsed  # units all ranks processed / the slowest rank's time
    frac_job = value * 32 * N / (distinct * HBM_PEAK_GBS * 1e9)

    # ---- roofline of the dominant kernel: HIP events on the launch stream, per pass -----------
    # The step is issued again pass by pass (nttx_run_pass_device, the same launches in the same order as
    # the two calls of step() make), back to back with an event between consecutive launches and no
    # synchronisation until the end: every kernel is timed in the sequence and at the power state of the
    # timed region, not in isolation (an isolated launch starts with power headroom and runs 1-9 % faster).
    stream = torch.cuda.current_stream(dev)
    npass = lib.nttx_pass_count(table._plan)
    names = []
    for direction in (0, 1):
        for pi in range(npass):
            buf = C.create_string_buffer(96)
            lib.nttx_pass_describe(table._plan, direction, pi, buf, 96)
            names.append((direction, pi, buf.value.decode()))
    reps = max(args.kernel_reps, 1) + 2
    marks = []
    for r in range(reps):
        row = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        row[0].record(stream)
        for i, (direction, pi, _) in enumerate(names):
            rc = lib.nttx_run_pass_device(table._plan, local_rank, data.data_ptr(), batch, direction, pi, stream.cuda_stream)
            if rc != 0:
                raise SystemExit("nttx_run_pass_device failed")
            row[i + 1].record(stream)
        marks.append(row)
    torch.cuda.synchronize()
    if not torch.equal(data, original) and not args.no_verify:
        raise SystemExit("the passes issued one by one do not add up to the transform pair")
    kernels = []
    for i, (_, _, name) in enumerate(names):
        ms = [row[i].elapsed_time(row[i + 1]) for row in marks[2:]]
        kernels.append({"name": name, "avg_ms": sum(ms) / len(ms)})
    dom = max(kernels, key=lambda k: k["avg_ms"])
    # algorithmic bytes: 16*N per transform (8N read + 8N written once), apportioned to a launch
    # by the share of the log2(N) stages it executes ("... stages a-b" in the pass name)
    a, b = dom["name"].rsplit(" ", 1)[1].split("-")
    alg_bytes = 16 * N * batch * (int(b) - int(a) + 1) / LOGN
    achieved = alg_bytes / (dom["avg_ms"] * 1e-3) / 1e9
    # HBM bytes per launch of that kernel from the PMC passes recorded under profiles/ (FETCH_SIZE
    # doubled as the gfx950 guide prescribes, plus WRITE_SIZE), quoted only while the kernel sources
    # are the ones they were measured on and the batch is the one they were measured at
    at_default = batch == default_batch
    traffic = recorded_counter("traffic", args.workload, dom["name"]) if at_default else None
    # Shader clock while the dominant kernel runs: a one-wave probe on a second stream reads the
    # shader-cycle and the 100 MHz counters around a 2 ms spin while that kernel is launched back to
    # back on the main stream (nttx_clock_probe_device).  The chip is power-limited under this load,
    # so the clock -- not the 2.4 GHz of the data sheet -- turns an instruction count into time.
    clock_mhz = None
    di = kernels.index(dom)
    probe_out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    readings = []
    for _ in range(5):  # median of five 1 ms probes, each in the middle of six back-to-back launches
        if di % npass == 0:
            data.copy_(original)
        torch.cuda.synchronize()
        for r in range(6):
            lib.nttx_run_pass_device(table._plan, local_rank, data.data_ptr(), batch, di // npass, di % npass, stream.cuda_stream)
            if r == 1:
                lib.nttx_clock_probe_device(local_rank, probe_out.data_ptr(), 1000, side.cuda_stream)
        torch.cuda.synchronize()
        cyc, ticks = (int(v) for v in probe_out.cpu())
        if ticks > 0 and cyc > 0:
            readings.append(100.0 * cyc / ticks)
    if readings:
        clock_mhz = sorted(readings)[len(readings) // 2]
    # leave `data` as the timed loop left it: these launches ran out of sequence
    data.copy_(original)

    # Board power under the timed loop's load.  The step is run back to back for about a second
    # while a thread reads the card's power sensor; the second half of the readings (the sensor
    # lags) against the limit the firmware enforces says whether the clock above is a power
    # limit at work: at the limit, time goes with the energy a step costs, not with issue slots.
    power = None
    sensor = None if args.no_power else board_power_sensor(torch, local_rank)
    if sensor:
        import threading
        samples, stop = [], threading.Event()

        def sample():
            while not stop.is_set():
                w = read_number(sensor[0])
                if w is not None:
                    samples.append(w / 1e6)
                time.sleep(0.02)

        th = threading.Thread(target=sample)
        th.start()
        t_end = time.perf_counter() + 1.0
        while time.perf_counter() < t_end:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
        stop.set()
        th.join()
        cap = read_number(sensor[1])
        late = sorted(samples[len(samples) // 2:])
        if late:
            power = {"board_w": round(late[len(late) // 2], 1), "limit_w": round(cap / 1e6, 1) if cap else None,
                     "samples": len(late), "source": "hwmon power1, second half of 1 s of back-to-back steps"}

    # second bound: integer VALU.  Wave-level VALU instructions per launch of that kernel
    # (SQ_INSTS_VALU); a wave64 instruction holds its SIMD for about 4 cycles (the multiplies and
    # 64-bit forms that make up most of a butterfly: tools/ubench_issue.hip) on 1024 SIMDs, priced at
    # the clock measured above.
    valu = None
    insts = recorded_counter("valu", args.workload, dom["name"]) if at_default else None
    if insts:
        ghz = (clock_mhz or 2400.0) / 1e3
        valu = {"wave_instructions_per_launch": insts, "clock_mhz": round(clock_mhz, 1) if clock_mhz else None,
                "cycles_per_instruction": 4,  # mean of the measured issue costs (2.5 for 32-bit adds / logic, 4.3-4.7 otherwise)
                "issue_utilisation": round(insts * 4 / (dom["avg_ms"] * 1e-3 * 1024 * ghz * 1e9), 3),
                "issue_utilisation_at_2400_mhz": round(insts * 4 / (dom["avg_ms"] * 1e-3 * 1024 * 2.4e9), 3)}
    elif clock_mhz:
        valu = {"clock_mhz": round(clock_mhz, 1)}
    # measured floors of the whole step (profiling build, tools/floors.sh): butterflies switched off
    # (memory only) and every workgroup on 64 L2-resident tiles (compute only)
    floors = None
    if at_default:
        mem_ms = recorded_counter("floors", args.workload, "memory_only_ms")
        cmp_ms = recorded_counter("floors", args.workload, "compute_only_ms")
        if mem_ms and cmp_ms:
            floors = {"memory_only_ms": mem_ms, "compute_only_ms": cmp_ms,
                      "shipped_ms_same_run": recorded_counter("floors", args.workload, "shipped_ms")}
    roofline = {
        "bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom["avg_ms"], 4),
        "whole_job_frac": round(frac_job, 4), "valu": valu, "floors": floors, "power": power,
        "kernels": [{"name": k["name"], "avg_ms": round(k["avg_ms"], 4)} for k in kernels],
    }

    # ---- CPU baseline: the oracle on this host's cores, bounded sample ------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_verify and LOGN <= 22:  # beyond that the CPU leg takes minutes
        import _oracle as orc  # the CPU restatement: checker and timed baseline of this leg only
        if kind == "mont":
            oracle = orc.Table.from_prime_and_root(Q61, PSI61, N)
        else:
            oracle = orc.TableGoldilock.from_root(psi, N)
        for b, got in forward_sample.items():
            want = original[b * N:(b + 1) * N].cpu().numpy().view(np.uint64).copy()
            oracle.forward_inplace(want)
            if not np.array_equal(got, want):
                raise SystemExit("parity failure on transform %d" % b)
        # threads: every core this process may run on, unless the box sets a thread budget for jobs
        # (OMP_NUM_THREADS on the shared GPU hosts, whose affinity mask shows the whole machine)
        host_cores = len(os.sched_getaffinity(0))
        budget = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
        cores = max(1, min(host_cores, budget) if budget > 0 else host_cores)
        okind = 0 if kind == "mont" else 2
        # bounded sample: about 1.5 s of transforms per thread
        per_thread = max(2, (600 << 16) >> LOGN if kind == "mont" else (300 << 16) >> LOGN)
        secs = orc.bench_pairs(okind, oracle, N, per_thread, cores)
        single = {}
        for k2, name, tab in ((0, "base", orc.Table.new()), (1, "avx2", orc.TableAVX2.new()),
                              (2, "goldilock", orc.TableGoldilock.new())):
            single[name + "_ms_30_pairs_1_thread"] = round(orc.bench_pairs(k2, tab, 1 << 16, 30, 1) * 1e3, 1)
        cpu = {
            "value": round(cores * per_thread / secs, 1), "unit": "fwd+inv NTT pairs/s", "cores": cores,
            "threads": cores, "host_cores": host_cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": "%d threads x %d pairs, N=2^%d, %s (C oracle)"
                      % (cores, per_thread, LOGN, "scalar Montgomery Table, q=0x1fffffffffe00001" if kind == "mont"
                         else "TableGoldilock"),
            "readme_shape": single,
        }

    if rank == 0:
        out = {
            "metric": "batched N=2^%d fwd+inv NTTs/sec per node; %% of HBM roofline" % LOGN,
            "value": round(value, 1), "unit": "fwd+inv NTT pairs/s",
            "n_gpus": distinct, "ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": (describe % job_batch).replace("per GPU", "in the job, split over %d GPUs" % world)
                       if strong else describe % batch, "batch_per_gpu": batch,
                       "job_batch": job_batch, "n": N, "modulus": hex(q),
                       "sharding": "batch split, no collective" + (", ranks share devices (rehearsal)" if shared else "")},
            "hbm_roofline_frac": round(frac_job, 4),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if args.no_verify:
            out["invalid"] = "--no-verify: results unchecked (profiling switches); not a benchmark record"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
