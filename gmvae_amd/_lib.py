"""ctypes binding of libgmvae_hip.so (the C ABI declared in include/gmvae_hip.h).

There is NO CPU fallback: if the shared library is missing this module raises,
and every op that needs the GPU raises when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# GMVAE_HIP_LIB: another build of the same ABI (A/B measurements of two kernels on one GPU box)
LIB_PATH = os.environ.get("GMVAE_HIP_LIB") or os.path.join(HERE, "lib", "libgmvae_hip.so")

MAX_HIDDEN = 8
TAIL = 8
ABI_VERSION = 7       # include/gmvae_hip.h GMVAE_ABI_VERSION: the layout of GmvaeDims below and the entry points bound in _load
MODEL_VAE, MODEL_VAE_GMP, MODEL_GMVAE = 0, 1, 2
MODEL_IDS = {"vae": MODEL_VAE, "vae_gmp": MODEL_VAE_GMP, "gmvae": MODEL_GMVAE}
NET_ENCODER_Y, NET_PRIOR_GMM, NET_ENCODER_GMM, NET_DECODER, NET_ENCODER = range(5)

ERRORS = {-1: "GMVAE_E_NULL", -2: "GMVAE_E_DIMS", -3: "GMVAE_E_MODEL", -4: "GMVAE_E_ALIGN",
          -5: "GMVAE_E_NET", -6: "GMVAE_E_SMALL", -7: "GMVAE_E_TIMEOUT"}


class GmvaeDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("D", C.c_int32), ("L", C.c_int32), ("K", C.c_int32), ("S", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN),
                ("sigma_min", C.c_float), ("raw_sigma_bias", C.c_float), ("temperature", C.c_float),
                ("gen_bias_init", C.c_float), ("row0", C.c_uint64),
                ("gen_bias_vec", C.c_void_p), ("gen_bias_len", C.c_int32), ("sched_flags", C.c_int32),    # ABI v4
                ("hidden_act", C.c_int32)]                                                                  # ABI v6


class GmvaeParamEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("rows", C.c_int32), ("cols", C.c_int32), ("offset", C.c_uint64)]


class GmvaeError(RuntimeError):
    pass


def _load():
    import torch  # noqa: F401  -- loads PyTorch's HIP runtime first; libgmvae_hip.so binds to the same one
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python build_hip.py` (or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32, f32 = C.c_void_p, C.c_uint64, C.c_int, C.c_float
    dp = C.POINTER(GmvaeDims)
    sigs = {
        "gmvae_abi_version": ([], i32),
        "gmvae_param_count": ([dp, i32, C.POINTER(u64), C.POINTER(u64)], i32),
        "gmvae_param_layout": ([dp, i32, C.POINTER(GmvaeParamEntry), i32, C.POINTER(i32)], i32),
        "gmvae_workspace_bytes": ([dp, i32, C.POINTER(u64)], i32),
        "gmvae_step": ([dp, i32, vp, vp, vp, vp, vp, vp, u64, u64, vp, vp], i32),
        "gmvae_forward": ([dp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u64, u64, vp], i32),
        "adam_tf_step": ([vp, vp, vp, vp, u64, f32, f32, f32, f32, u64, vp, f32, vp, vp, vp], i32),
        "gmvae_mlp_forward": ([dp, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp], i32),
        "gmvae_noise_fill": ([vp, vp, u64, i32, i32, u64, u64, u64, vp, vp], i32),
        "gmvae_cluster_acc": ([vp, vp, i32, i32, i32, vp, vp, vp], i32),
        "gmvae_gemm_test": ([vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
        "gmvae_bench_loop": ([dp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(f32), vp], i32),
        "gmvae_train_graph_create": ([dp, i32, vp, i32, vp, vp, vp, vp, vp, u64, vp, f32, f32, f32, f32, vp, C.POINTER(vp)], i32),
        "gmvae_train_graph_create_pipeline": ([dp, i32, vp, u64, vp, vp, i32, vp, vp, vp, vp, vp, u64, vp, f32, f32, f32, f32, vp, C.POINTER(vp)], i32),
        "gmvae_train_graph_launch": ([vp, vp], i32),
        "gmvae_train_graph_destroy": ([vp], i32),
        "gmvae_comm_unique_id": ([C.c_char_p, vp], i32),
        "gmvae_comm_init": ([C.c_char_p, vp, i32, i32, C.POINTER(vp)], i32),
        "gmvae_comm_destroy": ([vp], i32),
        "gmvae_comm_count": ([vp, C.POINTER(i32)], i32),
        "gmvae_dp_step": ([dp, i32, vp, vp, vp, vp, vp, vp, u64, vp, f32, f32, f32, f32, vp, vp], i32),
        "gmvae_dp_graph_create": ([dp, i32, vp, i32, vp, vp, vp, vp, vp, u64, vp, f32, f32, f32, f32, vp, vp, C.POINTER(vp)], i32),
        "gmvae_workspace_offset": ([dp, i32, C.c_char_p, C.POINTER(u64)], i32),
        "gmvae_binarize": ([vp, u64, vp, u64, i32, i32, u64, u64, vp, vp, u64, vp], i32),
        "gmvae_kernel_occupancy": ([i32, C.POINTER(i32)], i32),
        "gmvae_step_schedule": ([dp, i32, vp], i32),
        "gmvae_debug_sk_stamps": ([vp], i32),
        "gmvae_debug_sk_stamps_free": ([], i32),
        "gmvae_forward_profile": ([dp, i32, vp, vp, vp, vp, u64, i32, i32, C.POINTER(i32), vp, vp, vp, vp, vp], i32),
        "gmvae_dp_profile": ([dp, i32, vp, vp, vp, vp, vp, vp, u64, vp, f32, vp, i32, vp, i32, C.POINTER(i32), vp, vp], i32),
        "gmvae_train_profile": ([dp, i32, vp, vp, vp, vp, vp, vp, u64, vp, f32, i32, i32, C.POINTER(i32), vp, vp, vp, vp, vp], i32),
        "gmvae_step_profile": ([dp, i32, vp, vp, vp, vp, vp, vp, u64, i32, i32, C.POINTER(i32), vp, vp, vp, vp], i32),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.argtypes = args
        fn.restype = res
    got = lib.gmvae_abi_version()        # (GMVAE_HIP_LIB may name another build: an older one would ignore GmvaeDims' newer fields)
    if got != ABI_VERSION:
        raise GmvaeError(f"{LIB_PATH} has ABI v{got}, this binding is for v{ABI_VERSION} (include/gmvae_hip.h): rebuild with "
                         f"`python build_hip.py --force`")
    return lib, sorted(sigs)


lib, EXPORTS = _load()


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise GmvaeError(f"{what}: {ERRORS.get(rc, rc)}")
    raise GmvaeError(f"{what}: hipError_t {rc}")


SCHED_SAFE = 1        # GmvaeDims.sched_flags: only schedules without waits between the workgroups of a launch
SCHED_EVAL_IMAGES_VALID = 2      # ... forward-only: the images a previous gmvae_forward left in this workspace are current


ACTS = {"relu": 0, "tanh": 1, "sigmoid": 2, "elu": 3}       # GMVAE_ACT_*: GmvaeDims.hidden_act


def make_dims(B, D, L, K, hidden, S=1, sigma_min=0.0, raw_sigma_bias=0.5, temperature=1.0, gen_bias_init=0.0, row0=0,
              gen_bias_vec=None, sched_flags=0, hidden_act="relu"):
    """gen_bias_vec: fp32 device tensor [D] (ConditionalBernoulli's vector bias_init, scripts/base.py:102-103) or None;
    the caller keeps it alive for as long as the dims are used."""
    hidden = list(hidden)
    if len(hidden) > MAX_HIDDEN:
        raise ValueError(f"at most {MAX_HIDDEN} hidden layers")
    d = GmvaeDims()
    d.B, d.D, d.L, d.K, d.S, d.n_hidden = int(B), int(D), int(L), int(K), int(S), len(hidden)
    for i, h in enumerate(hidden):
        d.hidden[i] = int(h)
    d.sigma_min, d.raw_sigma_bias = float(sigma_min), float(raw_sigma_bias)
    d.temperature, d.gen_bias_init = float(temperature), float(gen_bias_init)
    d.sched_flags = int(sched_flags)
    d.hidden_act = ACTS[hidden_act] if isinstance(hidden_act, str) else int(hidden_act)
    d.row0 = int(row0)          # data parallel: global index of this device's first batch row (Philox counters only)
    if gen_bias_vec is not None:
        if gen_bias_vec.numel() != int(D) or not gen_bias_vec.is_cuda or not gen_bias_vec.is_contiguous():
            raise ValueError(f"gen_bias_vec must be a contiguous fp32 device tensor of {D} elements")
        d.gen_bias_vec, d.gen_bias_len = gen_bias_vec.data_ptr(), int(D)
    return d


def param_count(dims, model):
    pp, pr = C.c_uint64(), C.c_uint64()
    check(lib.gmvae_param_count(C.byref(dims), model, C.byref(pp), C.byref(pr)), "gmvae_param_count")
    return pp.value, pr.value


def param_layout(dims, model):
    """[(name, (rows, cols), offset)] in the reference's variable-creation order."""
    n = C.c_int()
    check(lib.gmvae_param_layout(C.byref(dims), model, None, 0, C.byref(n)), "gmvae_param_layout")
    arr = (GmvaeParamEntry * n.value)()
    check(lib.gmvae_param_layout(C.byref(dims), model, arr, n.value, C.byref(n)), "gmvae_param_layout")
    return [(e.name.decode(), (e.rows, e.cols), int(e.offset)) for e in arr]


def workspace_bytes(dims, model):
    b = C.c_uint64()
    check(lib.gmvae_workspace_bytes(C.byref(dims), model, C.byref(b)), "gmvae_workspace_bytes")
    return b.value


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise GmvaeError("no HIP device visible: gmvae_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def rccl_path():
    """The librccl.so PyTorch itself loaded (one RCCL per process)."""
    import torch
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p.encode() if os.path.exists(p) else b"librccl.so"


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def step_schedule(dims, model):
    """Name of the schedule a training step of these sizes takes (include/gmvae_hip.h gmvae_step_schedule)."""
    buf = C.create_string_buffer(48)
    check(lib.gmvae_step_schedule(C.byref(dims), int(model), buf), "gmvae_step_schedule")
    return buf.value.decode()
