"""Timing experiments on the FIRST form of the plane GEMM loop (plane_rounds, GMVAE_PLANES_FORM=1; results are wrong with a flag
set): GMVAE_PLANES_DBG 2 = no global loads in the loop, 4 = no LDS stores, 6 = neither.  (The third form's experiments were
run with temporary switches that are not in the tree: profiles/round3_notes.md.)"""
import sys, os, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gemm_bench import bench
    for name, M, N, K, tr, ns in [("fwd NN", 25600, 3072, 512, 0, 1), ("dX NT", 25600, 512, 3072, 1, 1), ("dW TN", 512, 3072, 25600, 2, 16)]:
        bench(M, N, K, tr, False, 4, ns, iters=2)
        us, tf = bench(M, N, K, tr, False, 5, ns, iters=20)
        print(f"  {name}: {us:8.1f} us", end="")
    print()
else:
    for f in ("0", "2", "4", "6"):
        print("DBG", f, end=": ", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, GMVAE_PLANES_DBG=f, GMVAE_PLANES_FORM="1"))
