"""Device wall-clock stamps of the skinny schedule's ten launches (GMVAE_SK_STAMPS=1): per launch the gap to the previous
launch's last workgroup end, the in-kernel span, and the phase medians of its workgroups."""
import sys, os, ctypes as C
os.environ["GMVAE_SK_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
L.check(L.lib.gmvae_debug_sk_stamps(None), "arm")
e = Engine("gmvae", 784, 128, 10, [512], random_seed=0)
B, G = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 8
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, 784)) < 0.87).astype(np.uint8)).cuda())
for _ in range(300): replay()
torch.cuda.synchronize()
buf = np.zeros(10 * 256 * 8, np.uint64)
L.check(L.lib.gmvae_debug_sk_stamps(buf.ctypes.data_as(C.c_void_p)), "stamps")
st = buf.reshape(10, 256, 8).astype(np.float64)
names = ["F1 first", "F2 ypath", "F3 qhead", "F4 dechid", "F5 decout", "B1 dhd", "B2 dz", "B3 dhg", "B4 ybwd", "W dw"]
prev_end = None
for i, nm in enumerate(names):
    r = st[i][st[i][:, 0] > 0]
    if not len(r): continue
    start, end = r[:, 0].min(), r[:, 3].max()
    gap = (start - prev_end) * 0.01 if prev_end is not None else float("nan")
    ph = [np.median(r[:, j + 1] - r[:, j]) * 0.01 for j in range(3)]
    print(f"{nm:10s} blocks {len(r):3d} gap-before {gap:5.2f} us | span {(end - start) * 0.01:5.2f} | first-WG start spread {(r[:,0].max()-start)*0.01:4.2f} | per-WG medians: loads+mfma {ph[0]:5.2f}  lds/sync {ph[1]:5.2f}  epilogue {ph[2]:5.2f}")
    prev_end = end
