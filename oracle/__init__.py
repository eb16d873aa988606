"""CPU oracle for the GMVAE / VAE / VAE_GMP ELBO training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gmvae_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline -- never as the product path.

PARITY UNPINNED.  The reference (mazrk7/gmvae) ships no tests, golden vectors
or fixtures, and its arithmetic lives in un-vendored TensorFlow 1.13.1 /
TensorFlow-Probability 0.6.0 / Sonnet v1 which are not installable here
(SURVEY.md section 8(c)).  This oracle restates the published algorithms of
those libraries at the reference's own call sites; it is pinned by the
closed-form known-answer values of SURVEY.md section 4 and by an independent
torch.distributions + autograd (fp64) statement in tests/test_oracle.py.
"""
from .gmvae_oracle import *  # noqa: F401,F403
