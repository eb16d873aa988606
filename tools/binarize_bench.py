"""gmvae_binarize (the input pipeline's dynamic binarisation) against the HBM roofline: D bytes read + D bytes
written per row.  Times 200 launches with hipEvents (torch.cuda.Event on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.data import binarize
D = 784
pix = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (60000, D), dtype=np.uint8)).cuda()
perm = torch.randperm(60000, device="cuda").to(torch.int32)
for B in (1024, 8192, 60000):
    out = torch.empty(B, D, dtype=torch.uint8, device="cuda")
    rows = perm[:B].contiguous()
    for gather in (False, True):
        for _ in range(20): binarize(pix, rows=rows if gather else None, batch=B, seed=1, step=0, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): binarize(pix, rows=rows if gather else None, batch=B, seed=1, step=i, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 200
        print(f"B={B:6d} {'permuted rows' if gather else 'contiguous   '}: {us:8.2f} us/launch  {2*B*D/us*1e-3:8.1f} GB/s of 8000 (HBM) = {2*B*D/us*1e-3/80:.1f} %")
