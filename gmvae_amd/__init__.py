"""gmvae_amd -- MI355X-native implementation of the mazrk7/gmvae ELBO training step.

Drop-in surface (same names as the reference's scripts/): create_vae,
create_gmvae, TrainableVAE, TrainableGMVAE, ConditionalNormal /
ConditionalBernoulli / ConditionalCategorical.  Importing this package loads
libgmvae_hip.so and fails loudly if it has not been built.
"""
from . import _lib                     # noqa: F401  (raises ImportError when the HIP extension is missing)
from .base import ConditionalBernoulli, ConditionalCategorical, ConditionalNormal  # noqa: F401
from .engine import Engine             # noqa: F401
from .gmvae import GMVAE, TrainableGMVAE, create_gmvae  # noqa: F401
from .vae import VAE, TrainableVAE, create_vae          # noqa: F401

__all__ = ["create_vae", "create_gmvae", "VAE", "TrainableVAE", "GMVAE", "TrainableGMVAE", "Engine",
           "ConditionalNormal", "ConditionalBernoulli", "ConditionalCategorical"]
