"""`python -m gmvae_amd.run_gmvae` -- the flag table of scripts/run_gmvae.py:11-58 (same names, defaults)."""
import argparse

from . import runners


def build_parser():
    p = argparse.ArgumentParser(description=__doc__)
    a = p.add_argument
    a("--mode", default="train", choices=["train", "eval"])
    a("--model", default="gmvae", choices=["gmvae", "vae", "vae_gmp"])
    a("--latent_size", type=int, default=8)
    a("--hidden_size", type=int, default=64)
    a("--num_layers", type=int, default=1)
    a("--mixture_components", type=int, default=10)
    a("--batch_size", type=int, default=16)
    a("--logdir", default="/tmp/smc_vi")
    a("--random_seed", type=int, default=None)          # any int, including 0, is honoured (the reference drops 0)
    a("--learning_rate", type=float, default=0.001)
    a("--max_steps", type=int, default=int(1e9))
    a("--early_stop_rounds", type=int, default=1000)
    a("--early_stop_threshold", type=float, default=0.001)
    a("--summarise_every", type=int, default=50)
    a("--gpu_id", default="0")                          # index INTO --gpu_num's list (runners.select_device); a launcher's
    a("--gpu_num", default="0")                         # LOCAL_RANK takes precedence (one process per GPU)
    a("--num_samples", type=int, default=10)            # eval prior draws (NOT IWAE samples)
    a("--num_generations", type=int, default=10)
    a("--split", default="train", choices=["train", "test"])
    # build-side additions
    a("--n_samples", type=int, default=1, help="IWAE samples per x (SURVEY.md A15); 1 == the reference")
    a("--data_dim", type=int, default=784)
    a("--data_dir", default=None, help="directory with mnist.npz or the IDX files; synthetic data otherwise")
    a("--synthetic_size", type=int, default=8192)
    a("--eager", action="store_true", help="eager launches instead of summarise_every-aligned train graphs")
    a("--checkpoint_poll_seconds", type=float, default=60.0)     # eval: scripts/utils.py:100-111 sleeps 60 s
    a("--checkpoint_max_wait", type=float, default=None, help="eval: give up waiting after this many seconds")
    return p


def main(argv=None):
    cfg = build_parser().parse_args(argv)
    return runners.run_train(cfg) if cfg.mode == "train" else runners.run_eval(cfg)


if __name__ == "__main__":
    main()
