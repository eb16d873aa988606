"""Device wall-clock stamps of the skinny schedule's ten launches (GMVAE_SK_STAMPS=1): per launch the gap to the previous
launch's last workgroup end, the in-kernel span, and the phase medians of its workgroups."""
import sys, os, ctypes as C
os.environ["GMVAE_SK_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
L.check(L.lib.gmvae_debug_sk_stamps(None), "arm")
B, G = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 8
Lz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
model = sys.argv[4] if len(sys.argv) > 4 else "gmvae"
e = Engine(model, 784, Lz, 10, [512], random_seed=0)
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, 784)) < 0.87).astype(np.uint8)).cuda())
for _ in range(300): replay()
torch.cuda.synchronize()
buf = np.zeros(10 * 1024 * 8, np.uint64)
L.check(L.lib.gmvae_debug_sk_stamps(buf.ctypes.data_as(C.c_void_p)), "stamps")
st = buf.reshape(10, 1024, 8).astype(np.float64)
names = ["F1 first", "F2 ypath", "F3 qhead", "F4 dechid", "F5 decout", "B1 dhd", "B2 dz", "B3 dhg", "B4 ybwd" if model == "gmvae" else "GMP prior", "W dw"]
order = list(range(10)) if model == "gmvae" else [0, 1, 2, 3, 4, 5, 6, 8, 7, 9]      # (VAE_GMP: sk_gmp_bwd stamps slot 8 and runs behind B2)
prev_end = None
for i in order:
    nm = names[i]
    r = st[i][st[i][:, 0] > 0]
    if not len(r): continue
    start, end = r[:, 0].min(), r[:, 3].max()
    gap = (start - prev_end) * 0.01 if prev_end is not None else float("nan")
    ph = [np.median(r[:, j + 1] - r[:, j]) * 0.01 for j in range(3)]
    print(f"{nm:10s} blocks {len(r):3d} gap-before {gap:5.2f} us | span {(end - start) * 0.01:5.2f} | first-WG start spread {(r[:,0].max()-start)*0.01:4.2f} | per-WG medians: loads+mfma {ph[0]:5.2f}  lds/sync {ph[1]:5.2f}  epilogue {ph[2]:5.2f}")
    prev_end = end
    if len(sys.argv) > 3 and (sys.argv[3] == "all" or nm.split()[0] in sys.argv[3].split(",")):
        o = np.argsort(r[:, 0])
        print("   start quantiles (us):", " ".join("%6.2f" % ((np.quantile(r[:, 0], q) - start) * 0.01) for q in (0, .25, .5, .75, .9, 1)),
              "| end quantiles:", " ".join("%6.2f" % ((np.quantile(r[:, 3], q) - start) * 0.01) for q in (0, .25, .5, .75, .9, 1)))
        for k in o[:: max(1, len(o) // 16)]:
            print("   wg start %7.2f  contraction %6.2f  meet %6.2f  epilogue %6.2f  end %7.2f" % ((r[k, 0] - start) * 0.01, (r[k, 1] - r[k, 0]) * 0.01, (r[k, 2] - r[k, 1]) * 0.01, (r[k, 3] - r[k, 2]) * 0.01, (r[k, 3] - start) * 0.01))
