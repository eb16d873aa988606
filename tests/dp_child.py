"""One rank of a world > 1 run on ONE GPU (started by tests/test_parallel.py as a fresh child process; never imported by
pytest).  torch.distributed over gloo (RCCL refuses two ranks on one device); GMVAE_NO_FL=1 / GMVAE_MEGA_Q=1 so that no
workgroup of one rank ever waits for a workgroup of its own launch while the other rank holds CUs.

  dp_child.py engine <outdir> <n_steps> <B_local>   Engine.sync_replicas() + train_step(all_reduce=True) on this rank's rows
                                                    of a fixed global batch sequence (data-parallel product path)
  dp_child.py runner <outdir> <args...>             run_gmvae --mode=train with the given flags
Each rank leaves <outdir>/rank<r>.pt: parameters, Adam moments, per-step tails, and (engine) the synced start state."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if mode == "engine":
        from gmvae_amd.engine import Engine
        n, B = int(sys.argv[3]), int(sys.argv[4])
        dist.init_process_group("gloo")
        e = Engine("gmvae", 784, 64, 10, [64], random_seed=None)       # the reference's default: every process seeds itself
        mine = e.params.detach().clone()
        e.sync_replicas()
        start = dict(params=e.params.detach().cpu().clone(), noise_seed=e.noise_seed, differed=bool((mine != e.params).any().item()))
        xs = (np.random.default_rng(77).random((n, world * B, 784)) < 0.87).astype(np.uint8)      # the GLOBAL batches
        tails = []
        for t in range(n):
            x = torch.from_numpy(xs[t, rank * B:(rank + 1) * B]).cuda()
            tails.append(e.train_step(x, lr=1e-3, all_reduce=True).clone().cpu())
        torch.cuda.synchronize()
        torch.save(dict(params=e.params.detach().cpu(), m=e.m.cpu(), v=e.v.cpu(), tails=torch.stack(tails), start=start,
                        global_step=e.global_step, timeouts=e.handoff_timeouts()), os.path.join(out, f"rank{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
    else:
        from gmvae_amd import run_gmvae, runners
        m = run_gmvae.main(sys.argv[3:])
        e = m._engine
        torch.cuda.synchronize()
        torch.save(dict(params=e.params.detach().cpu(), m=e.m.cpu(), v=e.v.cpu(), global_step=e.global_step,
                        path=runners.run_train.last_path, dp_mode=getattr(e, "dp_mode", None),
                        loss=(e.grads[e.P] / e.grads[e.P + 4]).item(), timeouts=e.handoff_timeouts()),
                   os.path.join(out, f"rank{rank}.pt"))
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
