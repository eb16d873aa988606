"""Gaussian Mixture VAE (mirror of scripts/gmvae.py).

``GMVAE`` (scripts/gmvae.py:12-188), ``TrainableGMVAE`` (191-274),
``create_gmvae`` (277-356): same names, argument orders and semantics.
``run_model`` is the fused HIP step.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L
from . import base
from . import utils
from .engine import Engine


class GMVAE:
    def __init__(self, mix_components, prior_gmm, decoder, encoder_y, encoder_gmm, random_seed):
        self._prior_gmm, self._decoder = prior_gmm, decoder
        self._encoder_y, self._encoder_gmm = encoder_y, encoder_gmm
        self.mix_components = mix_components
        self.random_seed = random_seed
        self._engine = None

    def prior_gmm(self, y):
        return self._prior_gmm(y)

    def decoder(self, z):
        return self._decoder(z)

    def encoder_y(self, x):
        return self._encoder_y(x)

    def encoder_gmm(self, x, y):
        return self._encoder_gmm(x, y)

    def reconstruct_images(self, images):
        y = self.encoder_y(images).sample(seed=self.random_seed)
        z = self.encoder_gmm(images, y).sample(seed=self.random_seed)
        return self.decoder(z).mean(name="reconstructions")

    def generate_sample_images(self, z=None, num_samples=1, name="sample_images"):
        if z is None:
            z = self.generate_samples(num_samples)
        return self.decoder(z).mean(name=name)

    def transform(self, inputs):
        """SAMPLED latent code (scripts/gmvae.py:140-149; the VAE returns the mean)."""
        y = self.encoder_y(inputs).sample(seed=self.random_seed)
        return self.encoder_gmm(inputs, y).sample(seed=self.random_seed, name="code")

    def generate_samples(self, num_samples, clusters=None):
        """[num_samples * K, L] draws from every p(z|y=k), or from the given clusters
        (scripts/gmvae.py:152-188)."""
        dev = self._engine.device if self._engine is not None else None
        if clusters is None:
            y = F.one_hot(torch.arange(self.mix_components, device=dev), self.mix_components).float()
        else:
            clusters = torch.as_tensor(clusters, device=dev).long()
            y = F.one_hot(clusters, self.mix_components).float()
        z = self.prior_gmm(y).sample(num_samples, seed=self.random_seed)
        return z.reshape(num_samples * y.shape[0], -1)

    def encode(self, x):
        q_y = self.encoder_y(x)
        return q_y, self.encoder_gmm(x, q_y.sample(seed=self.random_seed))

    decode = decoder


class TrainableGMVAE(GMVAE):
    def __init__(self, mix_components, prior_gmm, decoder, encoder_y, encoder_gmm, random_seed=None):
        super().__init__(mix_components, prior_gmm, decoder, encoder_y, encoder_gmm, random_seed=random_seed)
        self._last_labels = None

    def _need_engine(self):
        if self._engine is None:
            raise RuntimeError("run_model needs the fused HIP engine: build the model with create_gmvae()")
        return self._engine

    def run_model(self, images, targets, labels=None, eps=None, u=None):
        """Batch-mean loss = nll + kl_div_z + nent (scripts/gmvae.py:223-274); ELBO = -loss
        (the +ln K constant is omitted, as in the reference).  eps [B*S,L] / u [B*S,K]:
        optional explicit noise (parity mode); default is in-kernel Philox."""
        self._last_labels = labels
        return base._targets_guard(self._need_engine().loss(images, eps, u), images, targets)

    def compute_loss(self, images, n_samples=None, labels=None, eps=None, u=None):
        e = self._need_engine()
        if n_samples is not None and n_samples != e.S:
            raise ValueError(f"model was created with n_samples={e.S}")
        self._last_labels = labels
        return e.loss(images, eps, u)

    @property
    def summaries(self):
        """nll_scalar, kl_div_z, nent, elbo, cluster_acc of the last run_model
        (scripts/gmvae.py:255,259,264,268,272).  cluster_acc is evaluated lazily,
        like TF evaluates it only when the summary is fetched."""
        e = self._need_engine()
        t = e.grads[e.P:].detach()
        out = {"nll_scalar": t[1] / t[4], "kl_div_z": t[2] / t[4], "nent": t[3] / t[4], "elbo": -t[0] / t[4]}
        if self._last_labels is not None:
            d, ws = e._workspace(int(t[4].item()))
            out["cluster_acc"] = utils.cluster_acc(self.last_logits(), self._last_labels, self.mix_components)
        return out

    def last_logits(self):
        """q_y.distribution.logits of the last run_model batch (read from the step's workspace)."""
        e = self._need_engine()
        x = e._keep[0]
        return e.mlp(L.NET_ENCODER_Y, x)

    @property
    def params(self):
        return self._need_engine().params

    def state_dict(self):
        return self._need_engine().state_dict()

    def load_state_dict(self, sd):
        self._need_engine().load_state_dict(sd)


def create_gmvae(data_size, latent_size, mixture_components=1, fcnet_hidden_sizes=None,
                 hidden_activation_fn=torch.relu, sigma_min=0.001, raw_sigma_bias=0.25, gen_bias_init=0.0,
                 temperature=1.0, random_seed=None, n_samples=1):
    """Factory with the signature of scripts/gmvae.py:277-287 (+ n_samples)."""
    if fcnet_hidden_sizes is None:
        fcnet_hidden_sizes = [latent_size]                     # scripts/gmvae.py:316-317
    engine = Engine("gmvae", data_size, latent_size, mixture_components, fcnet_hidden_sizes, n_samples=n_samples,
                    sigma_min=sigma_min, raw_sigma_bias=raw_sigma_bias, temperature=temperature,
                    gen_bias_init=gen_bias_init, random_seed=random_seed, hidden_act=base.activation_name(hidden_activation_fn))
    prior_gmm = base.ConditionalNormal(size=latent_size, hidden_layer_sizes=None,
                                       hidden_activation_fn=hidden_activation_fn, sigma_min=sigma_min,
                                       raw_sigma_bias=raw_sigma_bias, name="prior_gmm").bind(engine, L.NET_PRIOR_GMM)
    decoder = base.ConditionalBernoulli(size=data_size, hidden_layer_sizes=fcnet_hidden_sizes,
                                        hidden_activation_fn=hidden_activation_fn, bias_init=gen_bias_init,
                                        name="decoder").bind(engine, L.NET_DECODER)
    encoder_y = base.ConditionalCategorical(size=mixture_components, temperature=temperature,
                                            hidden_layer_sizes=fcnet_hidden_sizes,
                                            hidden_activation_fn=hidden_activation_fn,
                                            name="encoder_y").bind(engine, L.NET_ENCODER_Y)
    encoder_gmm = base.ConditionalNormal(size=latent_size, hidden_layer_sizes=fcnet_hidden_sizes,
                                         hidden_activation_fn=hidden_activation_fn, sigma_min=sigma_min,
                                         raw_sigma_bias=raw_sigma_bias, name="encoder_gmm").bind(engine, L.NET_ENCODER_GMM)
    model = TrainableGMVAE(mixture_components, prior_gmm, decoder, encoder_y, encoder_gmm, random_seed=random_seed)
    model._engine = engine
    return model
