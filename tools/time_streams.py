"""The bench step (forward + backward of a resident batch) with the batch split over 1, 2 and 4 streams: do
transforms of one part fill the gaps that the dependent launches of another leave at their boundaries?
Usage: python3 tools/time_streams.py [mont16|goldilocks16] [steps]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench, ntt_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "mont16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kind, LOGN, batch, _ = bench.WORKLOADS[wl]
N = 1 << LOGN
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if kind == "mont":
    q = bench.Q61
    table = ntt_amd.Table.from_prime_and_root(bench.Q61, bench.PSI61, N, devices=[0])
else:
    q = bench.QGL
    table = ntt_amd.TableGoldilock.from_root(pow(7, (bench.QGL - 1) // (2 * N), bench.QGL), N, devices=[0])
data = torch.randint(0, min(q, 2**62), (batch * N,), dtype=torch.int64, device=dev)
original = data.clone()
streams = [torch.cuda.Stream() for _ in range(4)]


def run(parts):
    per = batch // parts
    views = [data[i * per * N:(i + 1) * per * N] for i in range(parts)]

    def step():
        for i, v in enumerate(views):
            with torch.cuda.stream(streams[i]):
                table.forward_inplace(v)
                table.backward_inplace(v)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rnd in range(3):
    out = []
    for parts in (1, 2, 4):
        out.append("%d stream%s %.4f ms" % (parts, "" if parts == 1 else "s", run(parts)))
    assert torch.equal(data, original)
    print("%s round %d: %s" % (wl, rnd, ", ".join(out)), flush=True)
