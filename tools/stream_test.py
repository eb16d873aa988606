import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
B = 1024
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
def run(label, stream, graph, n=300):
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        if graph:
            sx, fn = e.capture_train_step(B); sx.copy_(x)
        else:
            fn = lambda: e.train_step(x)
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:32s} {dt/n*1e6:8.1f} us/step", flush=True)
run("null stream, graph", None, True)
run("side stream, graph", torch.cuda.Stream(), True)
run("null stream, eager", None, False)
run("side stream, eager", torch.cuda.Stream(), False)
# host-only cost of the eager call path
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): e.train_step(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
