"""Wall-clock (100 MHz) stamps of the two grouped-GEMM launches of the mega schedule (GMVAE_STAMPS=1)."""
import sys, os, ctypes as C
os.environ["GMVAE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B = 1024
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
sx, replay = e.capture_train_step(B, 1e-3)          # the hipGraph the bench replays
sx.copy_(x)
for _ in range(3000): replay()                      # long enough for the clocks to ramp
torch.cuda.synchronize()
d, ws = e._workspace(B)
off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"gstamps", C.byref(off)), "off")
raw = ws.view(torch.int64)[off.value // 8: off.value // 8 + 2 * 2048 * 8].cpu().numpy().reshape(2, 2048, 8)
for k, name in enumerate(["P1 first layers + aux", "dW all"]):
    r = raw[k]; r = r[r[:, 0] > 0]
    t0 = r[:, 0].min()
    print(f"== {name}: {len(r)} blocks, span {(r[:,4].max()-t0)/100:.2f} us (start spread {(r[:,0].max()-t0)/100:.2f} us)")
    for pi in sorted(set(r[:, 5])):
        q = r[r[:, 5] == pi]
        if pi == 100:
            print(f"  aux: n={len(q)} start {np.median(q[:,0]-t0)/100:.2f} dur med {np.median(q[:,4]-q[:,0])/100:.2f} max {np.max(q[:,4]-q[:,0])/100:.2f} end max {(q[:,4].max()-t0)/100:.2f}")
            continue
        ph = np.diff(q[:, :5], axis=1) / 100.0
        print(f"      prologue split: kernargs {np.median(q[:,6]-q[:,0])/100:.2f} | loads issued+returned {np.median(q[:,7]-q[:,6])/100:.2f} | lds store+barrier {np.median(q[:,1]-q[:,7])/100:.2f}")
        print(f"  prob {pi}: n={len(q)} start med {np.median(q[:,0]-t0)/100:.2f} max {np.max(q[:,0]-t0)/100:.2f} | prologue {np.median(ph[:,0]):.2f} loop {np.median(ph[:,1]):.2f} stage {np.median(ph[:,2]):.2f} epi {np.median(ph[:,3]):.2f} | dur med {np.median(q[:,4]-q[:,0])/100:.2f} max {np.max(q[:,4]-q[:,0])/100:.2f} | end max {(q[:,4].max()-t0)/100:.2f}")
if os.environ.get("GSTAMPS_DETAIL"):
    r = raw[0]; n = int((r[:, 0] > 0).sum()); r = r[:n]
    t0 = r[:, 0].min(); st = (r[:, 0] - t0) / 100.0; en = (r[:, 4] - t0) / 100.0
    late = np.nonzero(st > 1.0)[0]
    print("P1 late-starting blocks:", len(late), "ids", late[:10], "...", late[-5:], "start", np.round(st[late][:8], 2))
    print("P1 block end times by id (every 32nd):", np.round(en[::32], 2))
    print("aux block durations:", np.round((r[:99, 4] - r[:99, 0]) / 100.0, 2))
