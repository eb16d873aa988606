import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gmvae_amd import _lib as L
torch.zeros(1, device="cuda")
for w, n in enumerate(["gemm CfgS", "gemm CfgM", "gemm CfgL", "mega_fwd_bwd", "finalize_adam"]):
    o = C.c_int(); rc = L.lib.gmvae_kernel_occupancy(w, C.byref(o)); print(n, "rc", rc, "blocks/CU", o.value)
