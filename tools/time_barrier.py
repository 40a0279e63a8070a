"""Cost of the barrier that brackets bench.py's timed region, one rank under torch.distributed.run:
dist.barrier over RCCL (device_ids given) against a barrier over a gloo side group, five calls each."""
import os, time
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
side = dist.new_group(backend="gloo")
x = torch.zeros(1 << 20, device=dev)
for name, fn in (("rccl barrier(device_ids)", lambda: dist.barrier(device_ids=[0])),
                 ("gloo side-group barrier", lambda: dist.barrier(group=side)),
                 ("rccl all_reduce of 1 element + synchronize", lambda: (dist.all_reduce(x[:1]), torch.cuda.synchronize()))):
    ts = []
    for _ in range(6):
        x.add_(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-45s first %.3f ms, then %s ms" % (name, ts[0], " ".join("%.3f" % t for t in ts[1:])), flush=True)
dist.destroy_process_group()
