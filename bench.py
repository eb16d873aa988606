#!/usr/bin/env python
"""ELBO-samples/sec of the GMVAE training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = noise (Philox) + forward + backward + (RCCL all-reduce of the flat
[P+8] gradient buffer when N>1) + TF-Adam, on a synthetic MNIST-shaped uint8
batch already resident in HBM.  Workload = BASELINE.json configs[2]/[3]:
GMVAE, D=784, K=10, latent 64, batch 1024 PER GPU (weak scaling: 8 GPUs = the
B=8192 config), hidden = the reference default 64 (scripts/run_gmvae.py:19).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


# Named workloads (per-GPU shapes).  configs0..2 = BASELINE.json configs[0..2]; configs4_shard = the per-GPU shard of
# configs[4] (B = 4096 / 8); run_train = the reference's bin/run_train.sh sizes.  Multi-GPU runs of configs2 ARE configs[3].
CONFIGS = {
    "configs0": dict(model="vae", batch=100, latent=2, components=1, hidden=64, layers=1, data_dim=784, n_samples=1),
    "configs1": dict(model="vae_gmp", batch=256, latent=64, components=10, hidden=64, layers=1, data_dim=784, n_samples=1),
    "configs2": dict(model="gmvae", batch=1024, latent=64, components=10, hidden=64, layers=1, data_dim=784, n_samples=1),
    "configs4_shard": dict(model="gmvae", batch=512, latent=64, components=64, hidden=512, layers=1, data_dim=3072, n_samples=50),
    "run_train": dict(model="gmvae", batch=64, latent=128, components=10, hidden=512, layers=1, data_dim=784, n_samples=1),
}


def workload_name(a, n_gpus):
    key = dict(model=a.model, batch=a.batch, latent=a.latent, components=a.components if a.model != "vae" else 1,
               hidden=a.hidden, layers=a.layers, data_dim=a.data_dim, n_samples=a.n_samples)
    for name, c in CONFIGS.items():
        if c == key:
            if name == "configs2":
                return "BASELINE configs[2]" if n_gpus == 1 else f"BASELINE configs[3] shape: 1024 rows per GPU x {n_gpus}"
            return {"configs0": "BASELINE configs[0]", "configs1": "BASELINE configs[1]",
                    "configs4_shard": "per-GPU shard of BASELINE configs[4]", "run_train": "bin/run_train.sh sizes"}[name]
    return "custom sizes"


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, BEFORE anything
    touches the GPU (one process per GPU over RCCL, the same command line the driver uses), and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--model", default="gmvae", choices=["gmvae", "vae", "vae_gmp"])
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--components", type=int, default=10)
    ap.add_argument("--data-dim", type=int, default=784)
    ap.add_argument("--n-samples", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--pipeline", action="store_true",
                    help="start every step from raw uint8 pixels resident in HBM: the reference's dynamic binarisation "
                         "(gmvae_binarize) runs inside the train graph (single GPU)")
    ap.add_argument("--graph-steps", type=int, default=40, help="consecutive steps captured in one hipGraph launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--levels", action="store_true", help="also print the per-launch table to stderr")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="a named workload (BASELINE.json configs / SURVEY.md 8(d)); the default line is configs2")
    a = ap.parse_args()
    if a.config:
        for k, v in CONFIGS[a.config].items():
            setattr(a, k, v)
    return a


def cpu_baseline(model, dims, B, flat, x, eps, u, budget_s):
    """CPU leg, part 2: the oracle's fp32 restatement of the SAME step (fwd+bwd+TF-Adam) timed on the host cores."""
    import oracle as O
    d = O.Dims(**dims)
    model_id = O.MODEL_NAMES[model]
    res = {}
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    ncores = len(os.sched_getaffinity(0))
    for label, nthreads in (("all", ncores), ("one", 1)):
        import contextlib
        ctx = threadpool_limits(limits=nthreads) if threadpool_limits else contextlib.nullcontext()
        with ctx:
            f, m, v = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
            O.train_step(model_id, d, f, m, v, 1, x, eps, u, dtype=np.float32)      # warm
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < budget_s / 2 and n < 200:
                f, m, v, _, _ = O.train_step(model_id, d, f, m, v, n + 1, x, eps, u, dtype=np.float32)
                n += 1
            dt = time.perf_counter() - t0
            res[label] = (B * d.S * n / dt, n, nthreads)
    return res


def flops_per_step(model: str, D: int, L: int, K: int, hidden, S: int, B: int) -> float:
    """SURVEY.md A.1 FLOP rule: 2*(fwd + dW + dX MACs) per sample, the x-input dX excluded (it is never computed)."""
    def mac(n_in, n_out):
        dims = [n_in] + list(hidden) + [n_out]
        return sum(p * q for p, q in zip(dims[:-1], dims[1:]))
    h0 = hidden[0]
    if model == "gmvae":
        fwd = mac(D, K) + S * (K * 2 * L + mac(D + K, 2 * L) + mac(L, D))
        dx = fwd - D * h0 * (1 + S)
    else:
        fwd = mac(D, 2 * L) + S * mac(L, D)
        dx = fwd - D * h0
    return 2.0 * B * (2 * fwd + dx)


def cpu_parity(model: str, dims: dict, flat0, x_np, eps_np, u_np):
    """CPU leg, part 1 (the checker): ELBO of the fp64 oracle on the inputs the HIP step just ran on."""
    import oracle as O
    d = O.Dims(**dims)
    mid = O.MODEL_NAMES[model]
    C64 = O.forward(mid, d, O.unpack(mid, d, flat0.astype(np.float64)), x_np, eps_np, u_np, np.float64)
    return float(-C64["loss"])


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = world

    from types import SimpleNamespace
    from gmvae_amd.engine import Engine

    hidden = [a.hidden] * a.layers
    dims = dict(D=a.data_dim, L=a.latent, K=a.components if a.model != "vae" else 1, hidden=tuple(hidden), S=a.n_samples)
    d = SimpleNamespace(**dims)
    B = a.batch
    # random_seed=None is the reference's default (scripts/run_gmvae.py:30): every process seeds itself from entropy, so the
    # replicas START different; sync_replicas() makes rank 0's parameters, moments, step and noise seed everyone's.  The
    # noise itself differs per rank by construction: the Philox counter holds the GLOBAL row (rank * B + b).
    eng = Engine(a.model, d.D, d.L, d.K, hidden, n_samples=d.S, random_seed=0 if world == 1 else None)
    eng.sync_replicas()
    # synthetic MNIST-shaped batch: Bernoulli(0.87) uint8, per-rank shard of the global batch
    x_np = (np.random.default_rng(1234 + rank).random((B, d.D)) < 0.87).astype(np.uint8)
    x = torch.from_numpy(x_np).cuda()

    # ---- parity in the same run: HIP vs the fp64 CPU restatement on identical (x, eps, u, parameters)
    rn = np.random.default_rng(42)
    tiny = float(np.finfo(np.float32).tiny)
    eps_np = rn.standard_normal((B * d.S, d.L)).astype(np.float32)
    u_np = np.clip(rn.uniform(tiny, 1.0, (B * d.S, d.K)).astype(np.float32), tiny, np.nextafter(np.float32(1), np.float32(0)))
    if a.model != "gmvae":
        u_np = None
    flat0 = eng.params.detach().cpu().numpy()
    buf = eng.step(x, torch.from_numpy(eps_np), None if u_np is None else torch.from_numpy(u_np))
    torch.cuda.synchronize()
    tail = buf[eng.P:].cpu().numpy().astype(np.float64)
    elbo_hip = -tail[0] / tail[4]
    parity = {}
    if rank == 0:
        elbo_cpu = cpu_parity(a.model, dims, flat0, x_np, eps_np, u_np)
        parity = {"elbo_hip": elbo_hip, "elbo_cpu_fp64": elbo_cpu, "rel_err": float(abs(elbo_hip - elbo_cpu) / abs(elbo_cpu))}

    # ---- the timed loop
    use_graph = not a.no_graph
    if world > 1:
        try:
            eng.enable_rccl()           # RCCL communicator inside libgmvae_hip.so: step + all-reduce + Adam in one graph
        except Exception as e:
            if rank == 0:
                print(f"[bench] in-library RCCL unavailable ({type(e).__name__}: {e}); torch.distributed all-reduce", file=sys.stderr)
    # One graph launch runs G consecutive steps, each on its own resident batch (the next G batches of an input
    # pipeline): the GPU idles ~6 us between two graph launches, nothing between the kernels inside one.
    # (the largest divisor of K up to --graph-steps, so that the timed K steps are whole launches of one graph)
    G = max(g for g in range(1, max(1, min(a.graph_steps, a.steps)) + 1) if a.steps % g == 0)
    if G < 8 <= a.steps:                 # no useful divisor: whole launches of --graph-steps + single steps for the rest
        G = min(a.graph_steps, a.steps)
    multi_fn = None
    if use_graph:
        try:
            static_x, replay = eng.capture_train_step(B, lr=1e-3, all_reduce=world > 1)
            static_x.copy_(x)
            step_fn = replay
            if a.pipeline and world == 1:
                from gmvae_amd.data import DeviceDataset
                ds = DeviceDataset(np.random.default_rng(99).integers(0, 256, (60000, d.D), dtype=np.uint8), shuffle=True, seed=1)
                multi_fn = eng.capture_train_pipeline(ds, B, lr=1e-3, n_steps=G)
                step_fn = eng.capture_train_pipeline(ds, B, lr=1e-3, n_steps=1)
            elif G > 1:
                xs, multi_fn = eng.capture_train_step(B, lr=1e-3, all_reduce=world > 1, n_steps=G)
                rng = np.random.default_rng(4321 + rank)
                xs.copy_(torch.from_numpy((rng.random((G, B, d.D)) < 0.87).astype(np.uint8)))
        except Exception as e:                      # e.g. RCCL inside capture unsupported
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            use_graph = False
            multi_fn = None
    if not use_graph:
        def step_fn():
            eng.train_step(x, lr=1e-3, all_reduce=world > 1)

    def run_steps(k):                               # exactly k training steps
        if multi_fn is not None:
            for _ in range(k // G):
                multi_fn()
            k = k % G
        for _ in range(k):
            step_fn()

    # The GPU drops its clocks while the host runs the fp64 parity check above; a step is ~100 us, so W
    # warm-up steps alone can be shorter than the DVFS ramp (measured: 2x slower timed region).  Spin the
    # same step untimed for a fixed wall time first, THEN do the W warm-up steps the contract asks for.
    if world == 1:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.75:
            run_steps(2 * G)
            torch.cuda.synchronize()
    else:
        # every rank must issue the SAME number of steps (each contains a collective): fixed count, not wall time
        run_steps(4000)
        torch.cuda.synchronize()
    # Safety net for unattended runs: the 3-launch schedule's workgroups wait for each other inside mega_fwd_bwd and
    # need the whole chip.  If a wait ever timed out during the pre-warm (something else held CUs), every rank
    # switches to the schedule without mutual waits and re-captures -- slower, never stalling.
    bad = torch.tensor([eng.handoff_timeouts()], dtype=torch.int32, device="cuda")
    if world > 1:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    safe_schedule = bool(bad.item())
    if safe_schedule and use_graph:
        os.environ["GMVAE_NO_FL"] = "1"
        eng.drop_graphs()
        eng.init_parameters(0)                       # poisoned steps were skipped by the optimizer, but start clean anyway
        eng.sync_replicas()
        static_x, step_fn = eng.capture_train_step(B, lr=1e-3, all_reduce=world > 1)
        static_x.copy_(x)
        multi_fn = None
        if G > 1:
            xs, multi_fn = eng.capture_train_step(B, lr=1e-3, all_reduce=world > 1, n_steps=G)
            xs.copy_(torch.from_numpy((np.random.default_rng(4321 + rank).random((G, B, d.D)) < 0.87).astype(np.uint8)))
        run_steps(2 * G)
        torch.cuda.synchronize()
    run_steps(a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_tail = eng.grads[eng.P:].cpu().numpy().astype(np.float64)
    # ---- the same loop once more with HIP events around every graph launch (median, events on the compute
    # stream).  `value` above stays the whole-region host clock the contract prescribes; this is the per-launch view.
    unit = G if multi_fn is not None else 1
    launch_fn = multi_fn if multi_fn is not None else step_fn
    n_ev = int(max(8, min(200, a.steps // unit)))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for e0, e1 in evs:
        e0.record()
        launch_fn()
        e1.record()
    torch.cuda.synchronize()
    ev_ms = sorted(e0.elapsed_time(e1) / unit for e0, e1 in evs)
    ms_median_events = ev_ms[len(ev_ms) // 2]
    if world > 1:
        dist.barrier()
    replicas_identical = None
    if world > 1:                                   # every rank must hold bit-identical parameters
        cs = eng.params.detach().double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool((lo == hi).item())
    value = n_gpus * B * d.S * a.steps / dt

    if rank == 0:
        # ---- roofline of the dominant kernel: hipEvents around every launch of the step
        # (the steady-state training step of the train graph when a single device runs it; the eager forward+backward
        #  step otherwise: the data-parallel step has an RCCL launch between its halves)
        try:
            levels = eng.profile_train_levels(x, lr=1e-3, iters=30) if world == 1 else eng.profile_levels(x, iters=30)
        except Exception:
            levels = eng.profile_levels(x, iters=30)
        gemms = [l for l in levels if l[2] > 0]          # launches that do MFMA work (grouped GEMMs, mega kernel)
        dom = max(gemms, key=lambda l: l[1])
        step_flops = flops_per_step(a.model, d.D, d.L, d.K, hidden, d.S, B)
        sum_us = sum(l[1] for l in levels)
        roof = {"bound": "mfma", "kernel": dom[0] if dom[0].startswith("mega") else f"gemm_grouped<{dom[0]}>", "achieved": dom[2] / dom[1] * 1e-6,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom[2] / dom[1] * 1e-6 / PEAK_F32_MFMA_TFLOPS,
                "traffic": None, "usec_per_launch": dom[1], "flops_per_launch": dom[2],
                # step_flops_alg: the reference's arithmetic for this step (every layer over all B*S rows);
                # step_flops_executed: what the launches run (S > 1: the layers over the S-times repeated input are
                # computed once per batch row) -- the fraction of the MFMA peak is quoted on the EXECUTED count
                "step_flops_alg": step_flops, "step_tflops_alg": step_flops / (dt / a.steps) * 1e-12,
                "step_flops_executed": sum(l[2] for l in levels),
                "step_frac_of_mfma_peak": sum(l[2] for l in levels) / (dt / a.steps) * 1e-12 / PEAK_F32_MFMA_TFLOPS,
                "launches_per_step": len(levels), "sum_kernel_usec": sum_us,
                "timing": "in-kernel wall-clock stamps (s_memrealtime): last workgroup end - first workgroup start, "
                          "steady-state step replayed inside a hipGraph" if world == 1 else "hipEvents around eager launches",
                "levels": [[nm, round(us, 2)] for nm, us, _ in levels]}
        # HBM-side bytes per launch of that kernel from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE;
        # rocprofv3 --pmc cannot run inside this process) and the committed rocprofv3 --kernel-trace --stats average of the
        # same kernel on the same command (profiles/round2_*, tools/profile_round2.sh) -- default workload only
        try:
            if workload_name(a, n_gpus) == "BASELINE configs[2]" and world == 1:
                kname = "gmvae::" + dom[0] if not dom[0].startswith("gemm") else dom[0]
                tj = json.load(open(os.path.join(ROOT, "profiles", "round2_traffic.json")))
                key = [k for k in tj if k.startswith(kname + " ") or k.startswith(kname + "(")]
                if key:
                    roof["traffic"] = tj[key[0]]["hbm_bytes_per_launch"]
                    roof["traffic_source"] = "profiles/round2_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                import csv
                for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "round2_bench_kernel_stats.csv"))):
                    if r["Name"].startswith(kname + "("):
                        us = float(r["AverageNs"]) * 1e-3
                        roof["rocprof_usec_per_launch"] = us
                        roof["frac_rocprof"] = dom[2] / us * 1e-6 / PEAK_F32_MFMA_TFLOPS
                        roof["rocprof_source"] = "profiles/round2_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command; the profiled run is slower than the timed one)"
            elif a.config == "configs4_shard" and world == 1:
                # the config-5 shard's dominant launch: the 128x128 GEMM launch with the most fabric traffic (decoder backward)
                tj = json.load(open(os.path.join(ROOT, "profiles", "round2_traffic_config5.json")))
                big = [v["hbm_bytes_per_launch"] for k, v in tj.items() if "gemm_grouped<gmvae::Cfg<128, 128" in k]
                if big and dom[0] == "bwd_dec_top":
                    roof["traffic"] = max(big)
                    roof["traffic_source"] = "profiles/round2_traffic_config5.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
        except Exception:
            pass
        if a.levels:
            for nm, us, fl in levels:
                print(f"  {nm:28s} {us:9.2f} us  {fl / max(us, 1e-9) * 1e-6:8.2f} TFLOP/s", file=sys.stderr)
        cpu = None
        if not a.no_cpu_baseline:
            r = cpu_baseline(a.model, dims, B, flat0, x_np, eps_np, u_np, a.cpu_seconds)
            best = "one" if r["one"][0] >= r["all"][0] else "all"      # the faster CPU variant is the baseline
            cpu = {"value": r[best][0], "unit": "ELBO-samples/sec", "cores": r[best][2], "kind": "port",
                   "sample": f"{r[best][1]} full steps (fwd+bwd+TF-Adam, fp32 NumPy/BLAS oracle) of the same "
                             f"B={B} batch on {r[best][2]} thread(s); all {r['all'][2]} cores: {r['all'][0]:.0f}/s, "
                             f"1 thread (the reference pins intra_op=inter_op=1): {r['one'][0]:.0f}/s",
                   "value_all_cores": r["all"][0], "value_1thread": r["one"][0]}
        out = {
            "metric": "ELBO-samples/sec", "value": value, "unit": "samples/sec", "n_gpus": n_gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "ms_per_step_median_hip_events": ms_median_events,
            "steps_per_graph_launch": unit, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.model} train step (noise+fwd+bwd+allreduce+TF-Adam), D={d.D} K={d.K} "
                                   f"L={d.L} hidden={hidden} S={d.S}, batch {B}/GPU x {n_gpus} GPU "
                                   f"({workload_name(a, n_gpus)})",
                       "global_batch": B * n_gpus, "parallelism": f"dp{n_gpus}", "hipgraph": use_graph, "input_pipeline_on_device": bool(a.pipeline and world == 1), "safe_schedule": safe_schedule,
                       "all_reduce": getattr(eng, "dp_mode", None), "replicas_identical": replicas_identical},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
            "final_loss": final_tail[0] / max(final_tail[4], 1.0),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
