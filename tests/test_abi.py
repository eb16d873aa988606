"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/gmvae_hip.h declares, and its host-only functions (layout, sizes,
argument validation) agree with the oracle.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import build_hip
    build_hip.build(verbose=False)
    from gmvae_amd import _lib
    return _lib


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "gmvae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\bint\s+(\w+)\s*\(", hdr))
    assert len(declared) >= 12
    raw = C.CDLL(L.LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), f"{sym} declared in include/gmvae_hip.h but not exported"
    assert declared == set(L.EXPORTS)
    assert L.lib.gmvae_abi_version() == L.ABI_VERSION == 7


@pytest.mark.parametrize("name,d", [
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,))),
    ("gmvae", O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)),
    ("gmvae", O.Dims(D=100, L=5, K=7, hidden=(24, 24))),
    ("gmvae", O.Dims(D=97, L=5, K=3, hidden=())),
    ("vae", O.Dims(D=784, L=2, K=1, hidden=(64,))),
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,))),
])
def test_layout_matches_oracle(L, name, d):
    model = O.MODEL_NAMES[name]
    cd = L.make_dims(16, d.D, d.L, d.K, d.hidden, S=d.S)
    lay, P, real = O.param_layout(model, d)
    assert L.param_count(cd, model) == (P, real)
    got = L.param_layout(cd, model)
    assert [g[0] for g in got] == [l[0] for l in lay]
    assert [g[2] for g in got] == [l[2] for l in lay]
    for (gn, (r, c), _), (_, shape, _) in zip(got, lay):
        assert r * c == int(np.prod(shape))
    assert L.workspace_bytes(cd, model) > 0


def test_argument_validation(L):
    cd = L.make_dims(16, 784, 64, 10, [64])
    pp = C.c_uint64()
    assert L.lib.gmvae_param_count(None, 2, C.byref(pp), None) == -1
    assert L.lib.gmvae_param_count(C.byref(cd), 7, C.byref(pp), None) == -3
    bad = L.make_dims(0, 784, 64, 10, [64])
    assert L.lib.gmvae_param_count(C.byref(bad), 2, C.byref(pp), None) == -2
    big_k = L.make_dims(16, 784, 64, 65, [64])
    assert L.lib.gmvae_param_count(C.byref(big_k), 1, C.byref(pp), None) == 0      # GMP prior: any K (tiled log-prob)
    vec = L.make_dims(16, 784, 64, 10, [64])
    vec.gen_bias_len = 10                                                          # ABI v3: a length without a vector,
    assert L.lib.gmvae_param_count(C.byref(vec), 2, C.byref(pp), None) == -2       # or one that is not D, is refused
    vec.gen_bias_vec, vec.gen_bias_len = 4096, 783
    assert L.lib.gmvae_param_count(C.byref(vec), 2, C.byref(pp), None) == -2
    vec.gen_bias_len = 784                                                         # (host-side size queries never read it)
    assert L.lib.gmvae_param_count(C.byref(vec), 2, C.byref(pp), None) == 0
    assert L.lib.gmvae_step(C.byref(cd), 2, None, None, None, None, None, None, 0, 0, None, None) == -1
    assert L.lib.adam_tf_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, None, 1.0, None, None, None) == -1
    n = C.c_int()
    arr = (L.GmvaeParamEntry * 2)()
    assert L.lib.gmvae_param_layout(C.byref(cd), 2, arr, 2, C.byref(n)) == -6
    with pytest.raises(ValueError):
        L.make_dims(1, 1, 1, 1, [1] * 9)


def test_dims_struct_matches_the_header(L):
    """The ctypes mirror of GmvaeDims has the header's field order (ABI v3 appended gen_bias_vec / gen_bias_len)."""
    hdr = open(os.path.join(ROOT, "include", "gmvae_hip.h")).read()
    body = re.search(r"typedef struct GmvaeDims \{(.*?)\} GmvaeDims;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(\w+)(?:\[\w+\])?;", body)
    assert names == [f[0] for f in L.GmvaeDims._fields_]
    assert C.sizeof(L.GmvaeDims) == 6 * 4 + 8 * 4 + 4 * 4 + 8 + 8 + 4 + 4 + 4 + 4      # (v6: + hidden_act, + tail padding to 8)


def test_initializers_and_device_flags_are_checked_on_the_host(monkeypatch):
    """scripts/base.py:18,49-50 `initializers`: None / the default pass, anything else must be a dict of callables
    (and is applied at bind time); scripts/run_gmvae.py:45-48 --gpu_id indexes --gpu_num's visible list."""
    import torch
    from types import SimpleNamespace
    from gmvae_amd import base, runners
    base.ConditionalNormal(4, [8], initializers=None)
    base.ConditionalBernoulli(4, [8], initializers=base.DEFAULT_INITIALIZERS)
    for bad in ({"w": 3}, {"q": lambda s: 0}, "xavier", {}):
        with pytest.raises(TypeError):
            base.ConditionalCategorical(4, [8], initializers=bad)
    for ok in (torch.relu, torch.tanh, torch.sigmoid, torch.nn.functional.elu, "elu", None):      # scripts/base.py:19: the kinds the kernels implement
        base.ConditionalNormal(4, [8], hidden_activation_fn=ok)
    with pytest.raises(NotImplementedError):
        base.ConditionalNormal(4, [8], hidden_activation_fn=torch.sin)

    def elu(x, alpha=0.3):                       # a user function that merely SHARES a name with an implemented kind
        return torch.where(x > 0, x, alpha * torch.expm1(x))
    with pytest.raises(NotImplementedError):
        base.ConditionalNormal(4, [8], hidden_activation_fn=elu)
    # a conditional bound to an Engine created with another activation would silently evaluate with the engine's
    with pytest.raises(ValueError):
        base.ConditionalNormal(4, [8], hidden_activation_fn=torch.tanh).bind(SimpleNamespace(hidden_act="relu"), 0)
    base.ConditionalNormal(4, None, hidden_activation_fn=torch.tanh).bind(SimpleNamespace(hidden_act="relu"), 0)   # no hidden layer: nothing to apply
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    sel = runners.select_device
    assert sel(SimpleNamespace(gpu_id="0", gpu_num="0"), 0) == 0
    assert sel(SimpleNamespace(gpu_id="1", gpu_num="2,3"), 0) == 3
    with pytest.raises(ValueError):
        sel(SimpleNamespace(gpu_id="2", gpu_num="2,3"), 0)
    with pytest.raises(ValueError):
        sel(SimpleNamespace(gpu_id="0", gpu_num="7"), 0)
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert sel(SimpleNamespace(gpu_id="0", gpu_num="0"), 1) == 1


def test_step_schedule_names(L, monkeypatch):
    """gmvae_step_schedule: which schedule a training step of given sizes takes (host-side; bench.py prices its roofline
    with it) -- the reference defaults, bin/run_train.sh's sizes, the config-5 shard, two hidden layers."""
    for k in ("GMVAE_NO_MEGA", "GMVAE_NO_MEGA2", "GMVAE_NO_SKINNY", "GMVAE_NO_FUSED", "GMVAE_NO_PLANES", "GMVAE_PLANES_MINROWS", "GMVAE_MEGA_Q"):
        monkeypatch.delenv(k, raising=False)
    G = L.MODEL_IDS["gmvae"]
    assert L.step_schedule(L.make_dims(1024, 784, 64, 10, [64]), G) == "mega2"
    assert L.step_schedule(L.make_dims(256, 784, 64, 10, [64]), L.MODEL_IDS["vae_gmp"]) == "mega2v"      # BASELINE configs[1]
    assert L.step_schedule(L.make_dims(100, 784, 2, 1, [64]), L.MODEL_IDS["vae"]) == "mega2v"           # BASELINE configs[0]
    assert L.step_schedule(L.make_dims(1024, 784, 64, 10, [64]), L.MODEL_IDS["vae_gmp"]) == "mega"      # (7 workgroups per panel no longer fit)
    assert L.step_schedule(L.make_dims(256, 784, 64, 10, [64], sched_flags=L.SCHED_SAFE), L.MODEL_IDS["vae_gmp"]) == "mega"
    assert L.step_schedule(L.make_dims(64, 784, 128, 10, [512]), G) == "skinny"
    assert L.step_schedule(L.make_dims(512, 3072, 64, 64, [512], S=50), G) == "general+planes"
    assert L.step_schedule(L.make_dims(40, 784, 16, 10, [64, 64]), G) == "general"
    monkeypatch.setenv("GMVAE_NO_PLANES", "1")
    assert L.step_schedule(L.make_dims(512, 3072, 64, 64, [512], S=50), G) == "general"


def test_workspace_layout_does_not_depend_on_schedule_switches(L, monkeypatch):
    """carve() sizes every buffer from the dims alone (the *_shape predicates): a workspace allocated under one set of
    GMVAE_NO_* schedule switches must fit -- and lay out identically -- under any other (a workspace sized with
    GMVAE_NO_SKINNY=1 and used without it ran past its end)."""
    cases = [("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64), ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 1024),
             ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(512,)), 256), ("vae", O.Dims(D=784, L=2, K=1, hidden=(64,)), 100),
             ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,)), 256)]
    names = [b"hy1", b"hd1", b"g", b"dz", b"dqp", b"slabs", b"z"]

    def probe(cd, mid):
        out = [L.workspace_bytes(cd, mid)]
        for nm in names:
            off = C.c_uint64()
            out.append(off.value if L.lib.gmvae_workspace_offset(C.byref(cd), mid, nm, C.byref(off)) == 0 else None)
        return out

    for name, d, B in cases:
        mid = O.MODEL_NAMES[name]
        cd = L.make_dims(B, d.D, d.L, d.K, d.hidden)
        for k in ("GMVAE_NO_SKINNY", "GMVAE_SKINNY_MAXB", "GMVAE_NO_MEGA", "GMVAE_NO_MEGA2", "GMVAE_NO_FUSED", "GMVAE_NO_PLANES",
                  "GMVAE_NO_FL", "GMVAE_MEGA_Q"):
            monkeypatch.delenv(k, raising=False)
        base = probe(cd, mid)
        for k, v in (("GMVAE_NO_SKINNY", "1"), ("GMVAE_SKINNY_MAXB", "8"), ("GMVAE_NO_MEGA", "1"), ("GMVAE_NO_MEGA2", "1"),
                     ("GMVAE_NO_FUSED", "1"), ("GMVAE_NO_PLANES", "1"), ("GMVAE_NO_FL", "1"), ("GMVAE_MEGA_Q", "1")):
            monkeypatch.setenv(k, v)
            assert probe(cd, mid) == base, (name, k)
            monkeypatch.delenv(k)
        cd.sched_flags = L.SCHED_SAFE
        assert probe(cd, mid) == base, (name, "sched_flags")
