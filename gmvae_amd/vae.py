"""VAE / VAE with a learned Gaussian-mixture prior (mirror of scripts/vae.py).

``VAE`` (scripts/vae.py:11-123), ``TrainableVAE`` (126-188), ``create_vae``
(191-271) keep the reference's names, argument orders and semantics;
``run_model`` is served by the fused HIP step (gmvae_step) and returns a scalar
torch tensor whose ``.backward()`` fills ``model.params.grad``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L
from . import base
from .engine import Engine


class VAE:
    def __init__(self, prior, decoder, encoder, mix_components, random_seed):
        self._prior, self._decoder, self._encoder = prior, decoder, encoder
        self.mix_components = mix_components
        self.random_seed = random_seed
        self._engine = None

    def prior(self):
        """p(z): fixed N(0,I) or the learned mixture (scripts/vae.py:41-48)."""
        return self._prior() if callable(self._prior) else self._prior

    def decoder(self, z):
        return self._decoder(z)

    def encoder(self, x):
        return self._encoder(x)      # the uint8 -> fp32 cast of scripts/vae.py:75 happens in the GEMM loader

    def reconstruct_images(self, images):
        q_z = self.encoder(images)
        z = q_z.sample(seed=self.random_seed)
        return self.decoder(z).mean(name="reconstructions")

    def generate_sample_images(self, z=None, num_samples=1):
        if z is None:
            z = self.generate_samples(num_samples)
        return self.decoder(z).mean(name="sample_images")

    def transform(self, inputs):
        """MEAN latent code (scripts/vae.py:108-114)."""
        return self.encoder(inputs).mean(name="code")

    def generate_samples(self, num_samples):
        z = self.prior().sample(num_samples, seed=self.random_seed, name="samples")
        return z.reshape(num_samples, -1)

    # north_star aliases
    encode = encoder
    decode = decoder


class TrainableVAE(VAE):
    def __init__(self, prior, decoder, encoder, mix_components=1, random_seed=None):
        super().__init__(prior, decoder, encoder, mix_components, random_seed)

    def _need_engine(self):
        if self._engine is None:
            raise RuntimeError("run_model needs the fused HIP engine: build the model with create_vae()")
        return self._engine

    def run_model(self, images, targets, eps=None):
        """Batch-mean loss = nll + kl_div_z (scripts/vae.py:153-188); ELBO = -loss.
        ``targets`` must be ``images`` (every reference call site passes the same
        tensor, scripts/runners.py:130).  eps: optional N(0,1) noise [B*S, L]."""
        return base._targets_guard(self._need_engine().loss(images, eps, None), images, targets)

    def compute_loss(self, images, n_samples=None, eps=None):
        e = self._need_engine()
        if n_samples is not None and n_samples != e.S:
            raise ValueError(f"model was created with n_samples={e.S}")
        return e.loss(images, eps, None)

    @property
    def summaries(self):
        """nll_scalar / kl_div_z / elbo of the last run_model (scripts/vae.py:178,182,186)."""
        e = self._need_engine()
        t = e.grads[e.P:].detach()
        return {"nll_scalar": t[1] / t[4], "kl_div_z": t[2] / t[4], "elbo": -t[0] / t[4]}

    @property
    def params(self):
        return self._need_engine().params

    def state_dict(self):
        return self._need_engine().state_dict()

    def load_state_dict(self, sd):
        self._need_engine().load_state_dict(sd)


def create_vae(data_size, latent_size, mixture_components=1, fcnet_hidden_sizes=None,
               hidden_activation_fn=torch.relu, sigma_min=0.001, raw_sigma_bias=0.25, gen_bias_init=0.0,
               random_seed=None, n_samples=1):
    """Factory with the signature of scripts/vae.py:191-200 (+ n_samples, the
    IWAE extension of SURVEY.md A15; 1 == the reference)."""
    if fcnet_hidden_sizes is None:
        fcnet_hidden_sizes = [latent_size]                     # scripts/vae.py:228-229
    name = "vae_gmp" if mixture_components > 1 else "vae"
    engine = Engine(name, data_size, latent_size, mixture_components, fcnet_hidden_sizes, n_samples=n_samples,
                    sigma_min=sigma_min, raw_sigma_bias=raw_sigma_bias, gen_bias_init=gen_bias_init,
                    random_seed=random_seed, hidden_act=base.activation_name(hidden_activation_fn))
    if mixture_components > 1:
        def prior():
            v = engine.views()
            return base.MixtureSameFamily(v["mixture_logits"], v["loc"], F.softplus(v["raw_scale_diag"]), name="prior")
    else:
        def prior():
            return base.MultivariateNormalDiag(torch.zeros(latent_size, device=engine.device),
                                               torch.ones(latent_size, device=engine.device), name="prior")
    decoder = base.ConditionalBernoulli(size=data_size, hidden_layer_sizes=fcnet_hidden_sizes,
                                        hidden_activation_fn=hidden_activation_fn, bias_init=gen_bias_init,
                                        name="decoder").bind(engine, L.NET_DECODER)
    encoder = base.ConditionalNormal(size=latent_size, hidden_layer_sizes=fcnet_hidden_sizes,
                                     hidden_activation_fn=hidden_activation_fn, sigma_min=sigma_min,
                                     raw_sigma_bias=raw_sigma_bias, name="encoder").bind(engine, L.NET_ENCODER)
    model = TrainableVAE(prior, decoder, encoder, mixture_components, random_seed=random_seed)
    model._engine = engine
    return model
