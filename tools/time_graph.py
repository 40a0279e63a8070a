"""One bench step (forward + backward of a resident batch) as plain launches against a captured HIP graph
replayed: what the gaps between the four dependent launches of a step cost.
Usage: python3 tools/time_graph.py [mont16|goldilocks16] [steps]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench, ntt_amd

wl = sys.argv[1] if len(sys.argv) > 1 else "mont16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
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

def step():
    table.forward_inplace(data)
    table.backward_inplace(data)

def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
for rnd in range(3):
    plain = timed(step)
    graph = timed(g.replay)
    assert torch.equal(data, original)
    print("%s round %d: plain %.4f ms/step, graph replay %.4f ms/step (%+.2f %%)" % (wl, rnd, plain, graph, (plain / graph - 1) * 100), flush=True)
