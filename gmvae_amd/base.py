"""Conditional distributions: MLP -> distribution (mirror of scripts/base.py).

Same class names, constructor arguments and methods as the reference
(``ConditionalNormal`` scripts/base.py:15-83, ``ConditionalBernoulli``
scripts/base.py:86-146, ``ConditionalCategorical`` scripts/base.py:149-209).
The MLP runs in the HIP library (Engine.mlp -> gmvae_mlp_forward); the light
distribution objects below play the role of the tfd.* objects the reference
returns and are used only by the forward-only auxiliary methods.  The training
loss does not go through them: it is the fused gmvae_step.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib as L

TINY = 1.1754943508222875e-38


def activation_name(fn) -> str:
    """hidden_activation_fn (scripts/base.py:19,90,153: any callable; the factories pass ONE to every network, gmvae.py:282,
    vae.py:196) -> the kind the kernels implement: relu (the reference's default tf.nn.relu), tanh, sigmoid, elu.  A name
    or the torch / torch.nn.functional callable; anything else raises (an arbitrary Python callable cannot run in a kernel)."""
    if fn is None:
        return "relu"
    table = {"relu": (torch.relu, F.relu), "tanh": (torch.tanh, F.tanh), "sigmoid": (torch.sigmoid, F.sigmoid), "elu": (F.elu,)}
    for name, fns in table.items():
        # by NAME only for the libraries' own functions (tf.nn.relu, tf.tanh, tf.nn.elu ... of a caller that still imports
        # TensorFlow): a user function that happens to be called `elu` may compute anything (another alpha)
        lib_fn = getattr(fn, "__name__", "") == name and (getattr(fn, "__module__", "") or "").split(".")[0] in ("torch", "tensorflow")
        if (isinstance(fn, str) and fn == name) or any(fn is f for f in fns) or lib_fn:
            return name
    raise NotImplementedError(f"hidden_activation_fn={fn!r}: the HIP kernels implement relu (the reference default, "
                              f"scripts/vae.py:196), tanh, sigmoid and elu")


def _check_relu(fn):          # (kept name: validates, returns nothing)
    activation_name(fn)


# scripts/base.py:12 DEFAULT_INITIALIZERS = {'w': xavier, 'b': zeros}: what Engine.init_parameters draws.  A custom
# `initializers` dict maps 'w' and/or 'b' to a callable shape -> array-like (the stand-in for a TF initializer op, which
# cannot exist here); it is applied to the network's tensors of the flat parameter buffer when the conditional is bound.
DEFAULT_INITIALIZERS = {"w": "xavier_uniform", "b": "zeros"}


def _check_initializers(init):
    if init is None or init is DEFAULT_INITIALIZERS or init == DEFAULT_INITIALIZERS:
        return None
    if not isinstance(init, dict) or not init or any(k not in ("w", "b") or not callable(v) for k, v in init.items()):
        raise TypeError("initializers must be None, base.DEFAULT_INITIALIZERS or a dict mapping 'w' / 'b' to callables "
                        f"shape -> array (scripts/base.py:18,49-50); got {init!r}")
    return dict(init)


def _gen(seed, device):
    if seed is None:
        return None
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


# ----------------------------------------------------------- distributions
class MultivariateNormalDiag:
    """tfd.MultivariateNormalDiag(loc, scale_diag) subset: sample / log_prob / mean."""

    def __init__(self, loc, scale_diag, name="MultivariateNormalDiag"):
        self.loc, self.scale_diag, self.name = loc, scale_diag, name

    def mean(self, name=None):
        return self.loc

    def sample(self, sample_shape=(), seed=None, name=None):
        shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
        full = shape + tuple(torch.broadcast_shapes(self.loc.shape, self.scale_diag.shape))
        eps = torch.randn(full, device=self.loc.device, generator=_gen(seed, self.loc.device))
        return self.loc + self.scale_diag * eps

    def log_prob(self, z):
        e = (z - self.loc) / self.scale_diag
        return (-0.5 * e * e - 0.5 * math.log(2 * math.pi) - torch.log(self.scale_diag)).sum(-1)


class IndependentBernoulli:
    """tfd.Independent(tfd.Bernoulli(logits), 1) subset."""

    def __init__(self, logits, name="Bernoulli"):
        self.logits, self.name = logits, name

    def mean(self, name=None):
        return torch.sigmoid(self.logits)

    def sample(self, seed=None):
        p = torch.sigmoid(self.logits)
        return torch.rand(p.shape, device=p.device, generator=_gen(seed, p.device)) < p

    def log_prob(self, x):
        x = x.to(self.logits.dtype)
        return (x * self.logits - F.softplus(self.logits)).sum(-1)


class _Categorical:
    def __init__(self, logits):
        self.logits = logits


class RelaxedOneHotCategorical:
    """tfd.RelaxedOneHotCategorical(temperature, logits) subset; ``.distribution.logits``
    is what scripts/gmvae.py:263,271 reads."""

    def __init__(self, temperature, logits, name="RelaxedOneHotCategorical"):
        self.temperature, self.logits, self.name = temperature, logits, name
        self.distribution = _Categorical(logits)

    def sample(self, seed=None, uniform=None):
        if uniform is None:
            uniform = torch.rand(self.logits.shape, device=self.logits.device,
                                 generator=_gen(seed, self.logits.device)).clamp_min(TINY)
        g = -torch.log(-torch.log(uniform))
        return torch.softmax((self.logits + g) / self.temperature, -1)


class MixtureSameFamily:
    """tfd.MixtureSameFamily(Categorical(logits), MVNDiag(loc[K,L], scale[K,L])) subset (scripts/vae.py:240-244)."""

    def __init__(self, mixture_logits, loc, scale_diag, name="prior"):
        self.mixture_logits, self.loc, self.scale_diag, self.name = mixture_logits, loc, scale_diag, name

    def log_prob(self, z):
        t = (z[..., None, :] - self.loc) / self.scale_diag
        ln = (-0.5 * t * t - 0.5 * math.log(2 * math.pi) - torch.log(self.scale_diag)).sum(-1)
        return torch.logsumexp(torch.log_softmax(self.mixture_logits, -1) + ln, -1)

    def sample(self, sample_shape=(), seed=None, name=None):
        n = sample_shape if isinstance(sample_shape, int) else int(torch.tensor(tuple(sample_shape)).prod())
        g = _gen(seed, self.loc.device)
        k = torch.multinomial(torch.softmax(self.mixture_logits, -1), n, replacement=True, generator=g)
        eps = torch.randn((n, self.loc.shape[1]), device=self.loc.device, generator=g)
        return self.loc[k] + self.scale_diag[k] * eps

    def mean(self, name=None):
        return (torch.softmax(self.mixture_logits, -1)[:, None] * self.loc).sum(0)


# -------------------------------------------------- conditional networks
class _Conditional:
    def __init__(self, size, hidden_layer_sizes, hidden_activation_fn, name, initializers=None):
        self._act = activation_name(hidden_activation_fn)
        self._initializers = _check_initializers(initializers)
        self._name, self._size = name, size
        self._hidden = None if hidden_layer_sizes is None else list(hidden_layer_sizes)
        self._engine = None
        self._net = None

    def bind(self, engine, net_id):
        """Attach to the flat parameter buffer (the factories do this; it stands in
        for Sonnet's lazy variable creation, scripts/base.py:47-60)."""
        # the activation is a field of the ENGINE's dims (one for every network, as the factories pass one): a conditional built
        # with another one would silently evaluate with the engine's
        if self._hidden and self._act != engine.hidden_act:
            raise ValueError(f"{self._name}: hidden_activation_fn is {self._act!r} but the Engine it is bound to was created "
                             f"with hidden_act={engine.hidden_act!r}")
        self._engine, self._net = engine, net_id
        if self._initializers:                      # custom initializers: re-draw this network's tensors
            prefix = self._name + "_fcnet/"
            hit = False
            with torch.no_grad():
                for name, view in engine.views().items():
                    fn = self._initializers.get(name[-1]) if name.startswith(prefix) else None
                    if fn is not None:
                        val = torch.as_tensor(fn(tuple(view.shape)), dtype=torch.float32).reshape(view.shape)
                        view.copy_(val.to(view.device))
                        hit = True
            if not hit:
                raise ValueError(f"{self._name}: no variables named {prefix}* in the engine's parameter layout")
            engine.drop_graphs()                    # (captured graphs hold weight images of the old values)
        return self

    def _mlp(self, tensor_list):
        if self._engine is None:
            raise RuntimeError(f"{self._name}: not bound to an Engine; build models with create_vae/create_gmvae")
        tensors = list(tensor_list)
        if self._net == L.NET_ENCODER_GMM:
            if len(tensors) != 2:
                raise ValueError("encoder_gmm expects (x, y)")      # concat([x, y], 1), scripts/base.py:66
            return self._engine.mlp(self._net, tensors[0], tensors[1])
        if len(tensors) != 1:
            raise ValueError(f"{self._name} expects one input tensor")
        return self._engine.mlp(self._net, tensors[0])


class ConditionalNormal(_Conditional):
    def __init__(self, size, hidden_layer_sizes=None, initializers=None, sigma_min=0.0, raw_sigma_bias=0.25,
                 hidden_activation_fn=torch.relu, name="cond_normal"):
        super().__init__(size, hidden_layer_sizes, hidden_activation_fn, name, initializers)
        self._sigma_min, self._raw_sigma_bias = sigma_min, raw_sigma_bias

    def condition(self, tensor_list, **unused_kwargs):
        outs = self._mlp(tensor_list)
        mu, raw = outs[:, :self._size], outs[:, self._size:]            # tf.split(outs, 2, axis=1)
        sigma = torch.clamp_min(F.softplus(raw + self._raw_sigma_bias), self._sigma_min)
        return mu, sigma

    def __call__(self, *args, **kwargs):
        mu, sigma = self.condition(args, **kwargs)
        return MultivariateNormalDiag(loc=mu, scale_diag=sigma, name=self._name)


class ConditionalBernoulli(_Conditional):
    def __init__(self, size, hidden_layer_sizes=None, initializers=None, bias_init=0.0,
                 hidden_activation_fn=torch.relu, name="cond_bernoulli"):
        super().__init__(size, hidden_layer_sizes, hidden_activation_fn, name, initializers)
        self._bias_init = bias_init

    def condition(self, tensor_list, **unused_kwargs):
        # + bias_init (scalar or vector, scripts/base.py:135) is applied inside gmvae_mlp_forward: the Engine this
        # conditional is bound to was created with the same value (gen_bias_init / gen_bias_vec of GmvaeDims)
        return self._mlp(tensor_list)

    def __call__(self, *args, **kwargs):
        return IndependentBernoulli(self.condition(args, **kwargs), name=self._name)


class ConditionalCategorical(_Conditional):
    def __init__(self, size, hidden_layer_sizes=None, temperature=1.0, initializers=None,
                 hidden_activation_fn=torch.relu, name="cond_categorical"):
        super().__init__(size, hidden_layer_sizes, hidden_activation_fn, name, initializers)
        self._temperature = temperature

    def condition(self, tensor_list, **unused_kwargs):
        return self._mlp(tensor_list)

    def __call__(self, *args, **kwargs):
        return RelaxedOneHotCategorical(self._temperature, logits=self.condition(args, **kwargs), name=self._name)


def _targets_guard(loss, images, targets):
    """Every reference call site passes targets = images (scripts/runners.py:127-130).  A distinct tensor is compared ON
    THE DEVICE and a mismatch turns the loss into NaN: loud, and without the device-to-host sync a torch.equal costs."""
    if targets is images or (targets.data_ptr() == images.data_ptr() and targets.shape == images.shape):
        return loss
    if targets.numel() != images.numel():
        raise NotImplementedError("targets != images is not used by the reference and not supported")
    bad = (targets.reshape(images.shape).to(images.device) != images).any()
    return loss + torch.where(bad, torch.full_like(loss, float("nan")), torch.zeros_like(loss))
