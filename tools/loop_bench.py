"""C-side loop: eager vs hipGraph, no Python per step."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
e = Engine("gmvae", 784, 64, 10, [H], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
d, ws = e._workspace(B)
for mode in (0, 1, 0, 1):
    us = C.c_float()
    rc = L.lib.gmvae_bench_loop(C.byref(d), e.model, L.ptr(x), L.ptr(e.params), L.ptr(e.m), L.ptr(e.v), L.ptr(e.grads),
                                L.ptr(ws), L.ptr(e.step_dev), 300, mode, C.byref(us), L.current_stream())
    print(f"H={H} B={B} mode={'graph' if mode else 'eager'} rc={rc}: {us.value:.1f} us/step -> {B/us.value:.2f} Msamples/s", flush=True)
