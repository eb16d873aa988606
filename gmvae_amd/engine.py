"""Host-side owner of the flat parameter buffer and driver of the HIP step.

Python / PyTorch here is plumbing only (device memory, streams, RCCL via
torch.distributed, hipGraph capture through torch.cuda.graph).  All arithmetic
of the hot path happens inside libgmvae_hip.so; there is no CPU fallback.

Replaces, for one device: the TF graph that scripts/runners.py:162-185 builds
(model -> loss -> AdamOptimizer.compute_gradients / apply_gradients) and the
``sess.run([train_op, global_step])`` of scripts/runners.py:231-232.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence

import torch

from . import _lib as L


class _ElboFn(torch.autograd.Function):
    """loss = mean_b loss_b with d loss / d params delivered by the fused HIP
    forward+backward (the backward pass has already run when this returns)."""

    @staticmethod
    def forward(ctx, params, engine, x, eps, u):
        buf = engine.step(x, eps, u)
        P = engine.P
        ctx.engine_buf = buf
        ctx.P = P
        return buf[P] / buf[P + 4]

    @staticmethod
    def backward(ctx, grad_out):
        buf, P = ctx.engine_buf, ctx.P
        return buf[:P] * (grad_out / buf[P + 4]), None, None, None, None


class Engine:
    def __init__(self, model: str, data_size: int, latent_size: int, mixture_components: int,
                 hidden: Sequence[int], n_samples: int = 1, sigma_min: float = 0.0, raw_sigma_bias: float = 0.5,
                 temperature: float = 1.0, gen_bias_init=0.0, random_seed: Optional[int] = None, hidden_act: str = "relu"):
        """gen_bias_init: a scalar or a vector of data_size values (scripts/base.py:102-103: "a scalar or vector Tensor
        that is added to the output of the fully connected network", e.g. the logit of the training-set mean)."""
        self.device = L.require_gpu()
        # data parallel: this process's shard index.  Row b of a local batch of B rows is global row rank*B + b for the
        # Philox counters (GmvaeDims.row0), so G ranks draw the noise of ONE step on the global batch of G*B rows.
        import torch.distributed as dist
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.model_name = model
        self.model = L.MODEL_IDS[model]
        self.D, self.Lz, self.K, self.S = int(data_size), int(latent_size), int(mixture_components), int(n_samples)
        self.hidden = [int(h) for h in hidden]
        self.gen_bias_vec = None
        if not isinstance(gen_bias_init, (int, float)):
            gb = torch.as_tensor(gen_bias_init, dtype=torch.float32).reshape(-1)
            if gb.numel() == 1:
                gen_bias_init = float(gb.item())
            elif gb.numel() == self.D:
                self.gen_bias_vec = gb.to(self.device).contiguous()
                gen_bias_init = 0.0
            else:
                raise ValueError(f"gen_bias_init must be a scalar or a vector of data_size = {self.D} values, got {gb.numel()}")
        if hidden_act not in L.ACTS:
            raise ValueError(f"hidden_act must be one of {sorted(L.ACTS)} (hidden_activation_fn, scripts/base.py:19), got {hidden_act!r}")
        self.hidden_act = hidden_act
        self.hp = dict(sigma_min=sigma_min, raw_sigma_bias=raw_sigma_bias, temperature=temperature,
                       gen_bias_init=float(gen_bias_init), hidden_act=hidden_act)
        self.safe_schedule = False              # use_safe_schedule(): the schedules without mutual waits (per engine)
        d0 = self.dims(1)
        self.P, self.P_real = L.param_count(d0, self.model)
        self.layout = L.param_layout(d0, self.model)
        self.random_seed = random_seed
        self.noise_seed = int(random_seed) if random_seed is not None else int(torch.seed() & 0x7FFFFFFFFFFFFFFF)
        self.params = torch.zeros(self.P, dtype=torch.float32, device=self.device, requires_grad=True)
        self.grads = torch.zeros(self.P + L.TAIL, dtype=torch.float32, device=self.device)
        self.m = torch.zeros(self.P, dtype=torch.float32, device=self.device)
        self.v = torch.zeros(self.P, dtype=torch.float32, device=self.device)
        self.step_dev = torch.zeros(2, dtype=torch.int64, device=self.device)   # global_step (device resident) + scratch copy
        self.global_step = 0
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._graphs: Dict[tuple, tuple] = {}
        self._graph_gen = 0                     # bumped by drop_graphs: replay closures of dropped graphs raise
        self._param_epoch = 0                   # bumped by every Engine call that enqueues a writer of the parameter buffer
        self._eval_imgs: Dict[tuple, tuple] = {}   # (B, S) -> the state of the parameters a forward-only pass left images for
        self.init_parameters(random_seed)

    # ------------------------------------------------------------ parameters
    def dims(self, B: int, S: Optional[int] = None, row0: Optional[int] = None, extra_flags: int = 0):
        """GmvaeDims for a local batch of B rows; row0 = global index of its first row (default rank * B)."""
        return L.make_dims(B, self.D, self.Lz, self.K, self.hidden, S=self.S if S is None else S,
                           row0=self.rank * B if row0 is None else int(row0), gen_bias_vec=self.gen_bias_vec,
                           sched_flags=(L.SCHED_SAFE if self.safe_schedule else 0) | extra_flags, **self.hp)

    def _params_state(self):
        """What identifies the parameter VALUES: torch's version counter of the buffer (in-place writes through torch) and the
        Engine's own count of the HIP writers it enqueued (the optimizer kernels, train graphs: invisible to torch)."""
        return (self._param_epoch, self.params._version, self.params.data_ptr())

    def sync_replicas(self, src: int = 0):
        """Data parallel: every rank takes rank `src`'s parameters, Adam moments, step counter and noise seed (the
        reference's default random_seed=None seeds each process from entropy, scripts/run_gmvae.py:30).  Afterwards
        replicas stay bit-identical: every rank applies the same all-reduced gradient."""
        import torch.distributed as dist
        from . import parallel
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        self._param_epoch += 1
        self.noise_seed, self.global_step = parallel.broadcast_state((self.params, self.m, self.v),
                                                                     (self.noise_seed, self.global_step), src)
        self.step_dev.fill_(self.global_step)
        self.drop_graphs()                       # captured graphs baked the old seed

    def init_parameters(self, seed: Optional[int] = None):
        """Xavier-uniform weights, zero biases (scripts/base.py:12); Glorot-uniform
        prior variables (tf.get_variable default, scripts/vae.py:233-238)."""
        gen = torch.Generator(device="cpu")
        if seed is not None:
            gen.manual_seed(int(seed))
        else:
            gen.seed()
        flat = torch.zeros(self.P, dtype=torch.float32)
        for name, (rows, cols), off in self.layout:
            if name.endswith("/b"):
                continue
            fan_in, fan_out = (cols, cols) if name == "mixture_logits" else (rows, cols)
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            flat[off:off + rows * cols] = (torch.rand(rows * cols, generator=gen) * 2 - 1) * lim
        self._param_epoch += 1
        with torch.no_grad():
            self.params.copy_(flat.to(self.device))
            self.m.zero_()
            self.v.zero_()
            self.step_dev.zero_()
        self.global_step = 0

    def views(self) -> Dict[str, torch.Tensor]:
        """Named views of the flat buffer, keyed by the reference's TF variable names."""
        out = {}
        for name, (rows, cols), off in self.layout:
            v = self.params.detach()[off:off + rows * cols]
            if name.endswith("/b") or name == "mixture_logits":
                out[name] = v
            else:
                out[name] = v.view(rows, cols)
        return out

    def _slot_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for name, (rows, cols), off in self.layout:
            v = flat[off:off + rows * cols]
            out[name] = v if (name.endswith("/b") or name == "mixture_logits") else v.view(rows, cols)
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Keys = the reference's TF checkpoint variable names: every model variable, its two Adam
        slots `<var>/Adam` (m) and `<var>/Adam_1` (v) as tf.train.AdamOptimizer names them (scripts/runners.py:181),
        the optimizer's `beta1_power` / `beta2_power` accumulators (= beta^(global_step+1) in TF 1.13: the value the
        NEXT apply uses) and `global_step`.  `noise_seed` (not a TF variable) keeps the Philox stream across a restart."""
        sd = {k: v.clone() for k, v in self.views().items()}
        for k, v in self._slot_views(self.m).items():
            sd[k + "/Adam"] = v.clone()
        for k, v in self._slot_views(self.v).items():
            sd[k + "/Adam_1"] = v.clone()
        sd["global_step"] = torch.tensor(self.global_step)
        sd["beta1_power"] = torch.tensor(0.9 ** (self.global_step + 1), dtype=torch.float32)
        sd["beta2_power"] = torch.tensor(0.999 ** (self.global_step + 1), dtype=torch.float32)
        sd["noise_seed"] = torch.tensor(self.noise_seed, dtype=torch.int64)
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        self._param_epoch += 1
        views = self.views()
        with torch.no_grad():
            for k, v in views.items():
                v.copy_(sd[k].to(self.device).reshape(v.shape))
            if "adam/m" in sd:                    # round-1 checkpoints: two flat blobs
                self.m.copy_(sd["adam/m"].to(self.device))
                self.v.copy_(sd["adam/v"].to(self.device))
            else:
                for slot, flat in (("/Adam", self.m), ("/Adam_1", self.v)):
                    for k, v in self._slot_views(flat).items():
                        if k + slot in sd:
                            v.copy_(sd[k + slot].to(self.device).reshape(v.shape))
            self.global_step = int(sd.get("global_step", 0))
            if "noise_seed" in sd:
                self.noise_seed = int(sd["noise_seed"])
            self.step_dev.fill_(self.global_step)
        self.drop_graphs()                        # graphs bake the noise seed

    # ------------------------------------------------------------- workspace
    def _workspace(self, B: int, S: Optional[int] = None, row0: Optional[int] = None):
        key = (B, self.S if S is None else S)
        d = self.dims(B, S, row0)
        # the size is re-queried on every call: the library sizes the layout from the dims alone, but its few remaining
        # test / tuning switches (GMVAE_NSPLIT_SMALL) enter the slab count -- a cached allocation must never be too small
        n = L.workspace_bytes(d, self.model) // 4 + 64
        if key in self._ws and self._ws[key].numel() < n:
            self.drop_graphs(clear_handoff_errors=False)      # captured graphs hold pointers into the old allocation
            del self._ws[key]
        if key not in self._ws:
            self._ws[key] = torch.zeros(n, dtype=torch.float32, device=self.device)
        return d, self._ws[key]

    @staticmethod
    def _as_u8(x: torch.Tensor) -> torch.Tensor:
        if x.dtype == torch.bool:
            x = x.view(torch.uint8) if x.is_contiguous() else x.to(torch.uint8)
        elif x.dtype != torch.uint8:
            x = x.to(torch.uint8)
        return x.contiguous()

    def _prep_x(self, x: torch.Tensor) -> torch.Tensor:
        x = self._as_u8(x.to(self.device))
        x = x.reshape(x.shape[0], -1)           # utils.flatten_tensor (scripts/utils.py:140-141)
        if x.shape[1] != self.D:
            raise ValueError(f"expected [B,{self.D}] inputs, got {tuple(x.shape)}")
        return x

    def _prep_noise(self, t, rows, cols):
        if t is None:
            return None
        t = t.to(self.device, torch.float32).contiguous()
        if t.numel() != rows * cols:
            raise ValueError(f"noise must have {rows}x{cols} elements")
        return t

    # ------------------------------------------------------------------ ops
    def step(self, x, eps=None, u=None, use_step_dev: bool = False, row0: Optional[int] = None) -> torch.Tensor:
        """Fused forward + backward.  Returns the [P + TAIL] buffer of gradient
        SUMS and loss sums (see include/gmvae_hip.h).  eps/u None -> Philox, keyed by
        (noise_seed, global_step, global row = row0 + b; row0 defaults to rank * B)."""
        x = self._prep_x(x)
        B = x.shape[0]
        d, ws = self._workspace(B, row0=row0)
        eps = self._prep_noise(eps, B * self.S, self.Lz)
        u = self._prep_noise(u, B * self.S, self.K) if self.model == L.MODEL_GMVAE else None
        rc = L.lib.gmvae_step(C.byref(d), self.model, L.ptr(x), L.ptr(eps), L.ptr(u), L.ptr(self.params),
                              L.ptr(self.grads), L.ptr(ws), self.noise_seed, self.global_step,
                              L.ptr(self.step_dev) if use_step_dev else None, L.current_stream())
        L.check(rc, "gmvae_step")
        self._keep = (x, eps, u)
        return self.grads

    def loss(self, x, eps=None, u=None) -> torch.Tensor:
        """Differentiable scalar: loss.backward() fills params.grad."""
        return _ElboFn.apply(self.params, self, x, eps, u)

    def forward(self, x, eps=None, u=None, n_samples: Optional[int] = None):
        """Forward only.  dict(tail[8], rows[R,4]=(logpx,logq,logp,logw), z, y, logits)."""
        x = self._prep_x(x)
        B = x.shape[0]
        S = self.S if n_samples is None else int(n_samples)
        d, ws = self._workspace(B, S)
        # an evaluation walks a split batch by batch on fixed parameters (scripts/runners.py:320-333): the operand images the
        # previous pass left in this workspace are reused while nothing has written the parameters since
        state = self._params_state() + (ws.data_ptr(),)
        if self._eval_imgs.get((B, S)) == state:
            d = self.dims(B, S, extra_flags=L.SCHED_EVAL_IMAGES_VALID)
        R = B * S
        eps = self._prep_noise(eps, R, self.Lz)
        gm = self.model == L.MODEL_GMVAE
        u = self._prep_noise(u, R, self.K) if gm else None
        f32 = dict(dtype=torch.float32, device=self.device)
        o = dict(tail=torch.empty(L.TAIL, **f32), rows=torch.empty(R, 4, **f32), z=torch.empty(R, self.Lz, **f32),
                 y=torch.empty(R, self.K, **f32) if gm else None,
                 logits=torch.empty(B, self.K, **f32) if gm else None)
        rc = L.lib.gmvae_forward(C.byref(d), self.model, L.ptr(x), L.ptr(eps), L.ptr(u), L.ptr(self.params),
                                 L.ptr(o["tail"]), L.ptr(o["rows"]), L.ptr(o["z"]), L.ptr(o["y"]), L.ptr(o["logits"]),
                                 L.ptr(ws), self.noise_seed, self.global_step, L.current_stream())
        L.check(rc, "gmvae_forward")
        self._eval_imgs[(B, S)] = state
        return o

    def mlp(self, net: int, inp: torch.Tensor, in2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One conditional network's MLP on the device (snt.nets.MLP, scripts/base.py:47-60)."""
        is_u8 = inp.dtype in (torch.uint8, torch.bool)
        inp = self._as_u8(inp.to(self.device)) if is_u8 else inp.to(self.device, torch.float32).contiguous()
        inp = inp.reshape(inp.shape[0], -1)
        rows = inp.shape[0]
        if in2 is not None:
            in2 = in2.to(self.device, torch.float32).contiguous()
        d, ws = self._workspace(rows, 1)
        out_dim = {L.NET_ENCODER_Y: self.K, L.NET_PRIOR_GMM: 2 * self.Lz, L.NET_ENCODER_GMM: 2 * self.Lz,
                   L.NET_DECODER: self.D, L.NET_ENCODER: 2 * self.Lz}[net]
        out = torch.empty(rows, out_dim, dtype=torch.float32, device=self.device)
        rc = L.lib.gmvae_mlp_forward(C.byref(d), self.model, net, L.ptr(inp), int(is_u8), L.ptr(in2), rows,
                                     L.ptr(self.params), L.ptr(out), L.ptr(ws), L.current_stream())
        L.check(rc, "gmvae_mlp_forward")
        return out

    def adam(self, lr: float = 1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, grads: Optional[torch.Tensor] = None,
             use_step_dev: bool = False):
        """TF-formula Adam over the flat buffer; gradient sums are scaled by
        1/count read from the (possibly all-reduced) tail on the device."""
        g = self.grads if grads is None else grads
        self._param_epoch += 1
        if not use_step_dev:
            self.global_step += 1
        count = g[self.P + 4:self.P + 5]
        loss_sum = g[self.P:self.P + 1]          # non-finite (poisoned step, on any rank) -> the update is skipped
        rc = L.lib.adam_tf_step(L.ptr(self.params), L.ptr(self.m), L.ptr(self.v), L.ptr(g), self.P, lr, beta1, beta2,
                                epsilon, self.global_step, L.ptr(self.step_dev) if use_step_dev else None, 1.0,
                                L.ptr(count), L.ptr(loss_sum), L.current_stream())
        L.check(rc, "adam_tf_step")
        if not use_step_dev:
            self.step_dev.fill_(self.global_step)   # one source of truth: graphs replayed later start from here

    def train_step(self, x, eps=None, u=None, lr: float = 1e-3, all_reduce: bool = True,
                   row0: Optional[int] = None) -> torch.Tensor:
        """One full reference step: fwd + bwd (+ RCCL all-reduce) + Adam.
        Returns the [TAIL] loss sums (device tensor; no host sync)."""
        import torch.distributed as dist
        self.step(x, eps, u, row0=row0)
        if all_reduce and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from . import parallel
            parallel.all_reduce_flat(self.grads)     # ONE collective: grads + loss sums + count
        self.adam(lr)
        return self.grads[self.P:]

    # -------------------------------------------------- data parallel over RCCL inside the C library
    def _agree(self, ok: bool) -> bool:
        """True only if `ok` on EVERY rank (one all-reduce(MIN)): ranks must never split over a fallback decision,
        or they would issue mismatched collectives and hang."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        t = t.to(self.device) if dist.get_backend() == "nccl" else t
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def enable_rccl(self):
        """Creates this rank's RCCL communicator inside libgmvae_hip.so (the 128-byte unique id travels over
        torch.distributed).  Afterwards train_step / capture_train_step(all_reduce=True) enqueue
        gradients -> ONE all-reduce -> Adam from a single C call (or one hipGraph).  Every rank takes every
        collective of this function whatever happens locally; a failure on any rank raises on ALL ranks."""
        import torch.distributed as dist
        if getattr(self, "_comm", None):
            return self._comm
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        buf = C.create_string_buffer(128)
        ok, why = True, ""
        if rank == 0:
            rc = L.lib.gmvae_comm_unique_id(L.rccl_path(), buf)
            ok, why = rc == 0, f"gmvae_comm_unique_id rc={rc}"
        if world > 1:
            t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            t = t.to(self.device) if dist.get_backend() == "nccl" else t
            dist.broadcast(t, src=0)
            buf = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        if not self._agree(ok):
            raise L.GmvaeError(f"in-library RCCL unavailable on some rank ({why or 'another rank failed'})")
        comm = C.c_void_p()
        torch.cuda.synchronize()
        rc = L.lib.gmvae_comm_init(L.rccl_path(), buf, rank, world, C.byref(comm))
        if not self._agree(rc == 0):
            raise L.GmvaeError(f"gmvae_comm_init failed on some rank (this rank rc={rc}"
                               f"{': a rank did not join within GMVAE_COMM_INIT_TIMEOUT seconds' if rc == -7 else ''})")
        n = C.c_int(0)
        rc = L.lib.gmvae_comm_count(comm, C.byref(n))
        if not self._agree(rc == 0 and n.value == world):
            raise L.GmvaeError(f"the RCCL communicator spans {n.value} ranks (rc={rc}), torch.distributed's world is {world}")
        self._comm = comm
        self.rccl_nranks = n.value
        return comm

    def dp_step(self, x, lr: float = 1e-3):
        """gmvae_dp_step: one C call enqueues step + RCCL all-reduce + Adam on the current stream."""
        x = self._prep_x(x)
        d, ws = self._workspace(x.shape[0])
        self._keep = (x, None, None)
        self._param_epoch += 1
        rc = L.lib.gmvae_dp_step(C.byref(d), self.model, L.ptr(x), L.ptr(self.params), L.ptr(self.m), L.ptr(self.v),
                                 L.ptr(self.grads), L.ptr(ws), self.noise_seed, L.ptr(self.step_dev), lr, 0.9, 0.999,
                                 1e-8, self._comm, L.current_stream())
        L.check(rc, "gmvae_dp_step")
        self.global_step += 1
        return self.grads[self.P:]

    # -------------------------------------------------- hipGraph fast path
    def capture_train_step(self, B: int, lr: float = 1e-3, all_reduce: bool = False, n_steps: int = 1):
        """One hipGraph for noise + fwd + bwd + Adam at batch size B, captured and owned by the HIP
        library (gmvae_train_graph_*).  Returns (static_x, replay).  With n_steps > 1 the graph holds
        that many consecutive steps and static_x is [n_steps, B, D] (the next n_steps batches): one
        launch per n_steps steps hides the idle time between graph launches.  With all_reduce (data
        parallel) the RCCL all-reduce is captured too when the library owns the communicator
        (enable_rccl); otherwise the step is two eager halves around torch.distributed.all_reduce."""
        import torch.distributed as dist
        do_ar = all_reduce and ((dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
                                or getattr(self, "_comm", None) is not None)
        n_steps = int(n_steps)
        key = (B, lr, do_ar, n_steps)
        if key in self._graphs:
            self.step_dev.fill_(self.global_step)   # eager steps may have run since the capture
            return self._graphs[key][:2]
        if n_steps == 1:
            static_x = torch.zeros(B, self.D, dtype=torch.uint8, device=self.device)
        else:
            static_x = torch.zeros(n_steps, B, self.D, dtype=torch.uint8, device=self.device)
        d, ws = self._workspace(B)
        self.step_dev.fill_(self.global_step)
        # per-step tails of one launch (loss sums + count; all-reduced under data parallelism): replay.tail_log
        tail_log = torch.zeros(n_steps, L.TAIL, dtype=torch.float32, device=self.device)
        if do_ar and getattr(self, "_comm", None):
            torch.cuda.synchronize()
            handle = C.c_void_p()
            rc = L.lib.gmvae_dp_graph_create(C.byref(d), self.model, L.ptr(static_x), n_steps, L.ptr(self.params), L.ptr(self.m),
                                             L.ptr(self.v), L.ptr(self.grads), L.ptr(ws), self.noise_seed,
                                             L.ptr(self.step_dev), lr, 0.9, 0.999, 1e-8, self._comm, L.ptr(tail_log),
                                             C.byref(handle))
            if self._agree(rc == 0):                # all ranks jointly: the graph, or (below) the eager C-side step
                launch = L.lib.gmvae_train_graph_launch
                self.dp_mode = "rccl-in-hipgraph"
                gen = self._graph_gen

                def replay():
                    self._check_alive(gen)
                    rc2 = launch(handle, L.current_stream())
                    if rc2:
                        L.check(rc2, "gmvae_train_graph_launch")
                    self.global_step += n_steps
                replay.tail_log = tail_log
                self._graphs[key] = (static_x, replay, handle)
                return static_x, replay
            if rc == 0:
                L.lib.gmvae_train_graph_destroy(handle)
            self.step_dev.fill_(self.global_step)   # capture refused somewhere: eager C-side step instead
            self.dp_mode = "rccl-eager-c"
            batches = [static_x] if n_steps == 1 else list(static_x.unbind(0))

            def replay():
                for i, xb in enumerate(batches):
                    self.dp_step(xb, lr)
                    tail_log[i].copy_(self.grads[self.P:])
            replay.tail_log = tail_log
            self._graphs[key] = (static_x, replay, None)
            return static_x, replay
        if do_ar:
            from . import parallel
            self.dp_mode = "torch.distributed"
            batches = [static_x] if n_steps == 1 else list(static_x.unbind(0))

            def replay():
                for i, xb in enumerate(batches):
                    self.step(xb, use_step_dev=True)
                    parallel.all_reduce_flat(self.grads)
                    self.adam(lr, use_step_dev=True)
                    self.global_step += 1
                    tail_log[i].copy_(self.grads[self.P:])
            replay.tail_log = tail_log
            self._graphs[key] = (static_x, replay, None)
            return static_x, replay
        torch.cuda.synchronize()
        handle = C.c_void_p()
        rc = L.lib.gmvae_train_graph_create(C.byref(d), self.model, L.ptr(static_x), n_steps, L.ptr(self.params), L.ptr(self.m),
                                            L.ptr(self.v), L.ptr(self.grads), L.ptr(ws), self.noise_seed,
                                            L.ptr(self.step_dev), lr, 0.9, 0.999, 1e-8, L.ptr(tail_log), C.byref(handle))
        L.check(rc, "gmvae_train_graph_create")
        launch = L.lib.gmvae_train_graph_launch
        gen = self._graph_gen

        def replay():
            self._check_alive(gen)
            rc = launch(handle, L.current_stream())
            if rc:
                L.check(rc, "gmvae_train_graph_launch")
            self.global_step += n_steps

        replay.tail_log = tail_log
        self._graphs[key] = (static_x, replay, handle)
        return static_x, replay

    BINARIZE_SEED_XOR = 0x62696E6172697A65      # the pipeline graph keys its binarisation uniforms by noise_seed ^ this

    def capture_train_pipeline(self, dataset, B: int, lr: float = 1e-3, n_steps: int = 16):
        """A train graph that starts from the RAW pixels (gmvae_train_graph_create_pipeline): each of its n_steps
        steps first binarises its own batch on the device (scripts/runners.py:44-47), rows taken from `dataset`
        (gmvae_amd.data.DeviceDataset: resident uint8 pixels + an epoch permutation on the device).  Returns
        replay(): refills the row indices (device-to-device) and launches the graph; nothing crosses PCIe."""
        n_steps = int(n_steps)
        key = ("pipeline", id(dataset), B, lr, n_steps)
        if key in self._graphs:
            self.step_dev.fill_(self.global_step)
            return self._graphs[key][1]
        if dataset.D != self.D:
            raise ValueError("dataset rows must have D pixels")
        d, ws = self._workspace(B)
        idx = torch.zeros(n_steps, B, dtype=torch.int32, device=self.device)
        xs = torch.zeros(n_steps, B, self.D, dtype=torch.uint8, device=self.device)
        tail_log = torch.zeros(n_steps, L.TAIL, dtype=torch.float32, device=self.device)
        self.step_dev.fill_(self.global_step)
        torch.cuda.synchronize()
        handle = C.c_void_p()
        rc = L.lib.gmvae_train_graph_create_pipeline(C.byref(d), self.model, L.ptr(dataset.pixels), dataset.N, L.ptr(idx),
                                                     L.ptr(xs), n_steps, L.ptr(self.params), L.ptr(self.m), L.ptr(self.v),
                                                     L.ptr(self.grads), L.ptr(ws), self.noise_seed, L.ptr(self.step_dev), lr,
                                                     0.9, 0.999, 1e-8, L.ptr(tail_log), C.byref(handle))
        L.check(rc, "gmvae_train_graph_create_pipeline")
        launch = L.lib.gmvae_train_graph_launch
        gen = self._graph_gen

        def replay():
            self._check_alive(gen)
            idx.copy_(dataset.next_rows(n_steps * B).view(n_steps, B))
            rc2 = launch(handle, L.current_stream())
            if rc2:
                L.check(rc2, "gmvae_train_graph_launch")
            self.global_step += n_steps

        replay.rows, replay.batches, replay.tail_log = idx, xs, tail_log   # the buffers of the last launch (tests, summaries)
        self._graphs[key] = (xs, replay, handle)
        return replay

    def _check_alive(self, gen: int):
        self._param_epoch += 1                  # (every train-graph replay passes here: its kernels write the parameters)
        if gen != self._graph_gen:
            raise L.GmvaeError("this replay closure belongs to a train graph that drop_graphs() destroyed; capture again")

    def _sync_word(self, key, ws):
        off = C.c_uint64()
        d = self.dims(key[0], key[1])
        if L.lib.gmvae_workspace_offset(C.byref(d), self.model, b"sync", C.byref(off)) != 0:
            return None
        return ws.view(torch.int32)[off.value // 4 + 1:off.value // 4 + 2]

    def handoff_timeouts(self) -> int:
        """Number of workspaces whose in-launch hand-off (mega_fwd_bwd's tagged-granule exchange between the
        workgroups of a panel) ever gave up waiting.  Such a step poisons its loss and gradients with NaN and the
        optimizer skips it (params, m, v untouched); this is the explicit flag.  0 on a healthy device."""
        bad = 0
        for key, ws in self._ws.items():
            w = self._sync_word(key, ws)
            if w is not None:
                bad += int(w.item() != 0)
        return bad

    def inject_handoff_fault(self):
        """Test / diagnostics hook: set the hand-off error word of every workspace, exactly what a timed-out wait of the
        fused schedule leaves behind (a co-tenant or a partitioned device kept part of a panel's workgroups off the chip)."""
        for key, ws in self._ws.items():
            w = self._sync_word(key, ws)
            if w is not None:
                w.fill_(1)

    def use_safe_schedule(self):
        """Switch THIS ENGINE to the schedules WITHOUT mutual waits between workgroups (first layer as its own launch, one
        workgroup per panel: GmvaeDims.sched_flags = GMVAE_SCHED_SAFE on every call from now on -- no process-wide state),
        destroy the captured graphs and clear the error words.  Slower (4 launches per step instead of 2), never stalling:
        what run_train and bench.py degrade to when a hand-off of the fused schedule timed out."""
        self.safe_schedule = True
        self.drop_graphs(clear_handoff_errors=True)

    def drop_graphs(self, clear_handoff_errors: bool = True):
        """Destroy every captured train graph (they are re-captured on the next capture_* call, under whatever
        GMVAE_* schedule switches are set by then; replay closures handed out before raise from now on) and,
        optionally, clear the workspaces' hand-off error words."""
        for _, _, handle in self._graphs.values():
            if handle:
                L.lib.gmvae_train_graph_destroy(handle)
        self._graphs.clear()
        self._graph_gen += 1
        if clear_handoff_errors:
            for key, ws in self._ws.items():
                w = self._sync_word(key, ws)
                if w is not None:
                    w.zero_()

    def __del__(self):
        try:
            for _, _, handle in self._graphs.values():
                if handle:
                    L.lib.gmvae_train_graph_destroy(handle)
            self._graphs.clear()
            self._graph_gen += 1
        except Exception:
            pass

    def profile_train_levels(self, x, lr: float = 1e-3, iters: int = 20):
        """Per-launch timing of the steady-state training step of a train graph (gmvae_train_profile): a list of
        (name, in-kernel span us, algorithmic FLOPs, timeline share us).  The timeline share runs from the launch's first
        workgroup start to the next launch's (dispatch + end-of-kernel write-back included: what rocprofv3 reports); the
        shares add up to the step.  Advances the optimizer by 3 * iters steps."""
        x = self._prep_x(x)
        d, ws = self._workspace(x.shape[0])
        self._param_epoch += 1
        self.step_dev.fill_(self.global_step)
        n = C.c_int()
        names = C.create_string_buffer(96 * 48)
        usec = (C.c_float * 96)()
        usec_tl = (C.c_float * 96)()
        flops = (C.c_double * 96)()
        rc = L.lib.gmvae_train_profile(C.byref(d), self.model, L.ptr(x), L.ptr(self.params), L.ptr(self.m), L.ptr(self.v),
                                       L.ptr(self.grads), L.ptr(ws), self.noise_seed, L.ptr(self.step_dev), lr, iters, 96,
                                       C.byref(n), names, usec, usec_tl, flops, L.current_stream())
        L.check(rc, "gmvae_train_profile")
        self.global_step += 3 * iters
        out = []
        for i in range(n.value):
            nm = names.raw[i * 48:(i + 1) * 48].split(b"\0")[0].decode()
            out.append((nm, float(usec[i]), float(flops[i]), float(usec_tl[i])))
        return out

    def profile_forward(self, x, n_samples: Optional[int] = None, iters: int = 20):
        """gmvae_forward_profile: the forward-only evaluation (in-kernel Philox noise) at n_samples importance samples.
        Returns ([(launch, usec, flops)], usec per forward of a replayed hipGraph, the pass's [TAIL] loss sums)."""
        x = self._prep_x(x)
        S = self.S if n_samples is None else int(n_samples)
        d, ws = self._workspace(x.shape[0], S)
        tail = torch.empty(L.TAIL, dtype=torch.float32, device=self.device)
        n = C.c_int(0)
        names = C.create_string_buffer(96 * 48)
        usec = (C.c_float * 96)()
        flops = (C.c_double * 96)()
        total = C.c_float(0)
        rc = L.lib.gmvae_forward_profile(C.byref(d), self.model, L.ptr(x), L.ptr(self.params), L.ptr(tail), L.ptr(ws),
                                         self.noise_seed, iters, 96, C.byref(n), names, usec, flops, C.byref(total),
                                         L.current_stream())
        L.check(rc, "gmvae_forward_profile")
        lev = [(names.raw[i * 48:(i + 1) * 48].split(b"\0")[0].decode(), usec[i], flops[i]) for i in range(n.value)]
        return lev, total.value, tail

    def profile_dp_step(self, x, lr: float = 1e-3, iters: int = 10):
        """gmvae_dp_profile: the data-parallel step's timeline with this engine's RCCL communicator (enable_rccl first; a
        one-rank communicator is allowed).  COLLECTIVE: every rank calls it with the same `iters`.  Returns a dict of
        microseconds: grad_span, allreduce_window, adam_span, gap_to_next_step, step; and the gradient launches' names."""
        if not getattr(self, "_comm", None):
            raise L.GmvaeError("profile_dp_step needs enable_rccl()")
        x = self._prep_x(x)
        d, ws = self._workspace(x.shape[0])
        self._param_epoch += 1
        out = (C.c_float * 8)()
        n = C.c_int(0)
        names = C.create_string_buffer(16 * 48)
        rc = L.lib.gmvae_dp_profile(C.byref(d), self.model, L.ptr(x), L.ptr(self.params), L.ptr(self.m), L.ptr(self.v),
                                    L.ptr(self.grads), L.ptr(ws), self.noise_seed, L.ptr(self.step_dev), lr, self._comm, iters,
                                    out, 16, C.byref(n), names, L.current_stream())
        L.check(rc, "gmvae_dp_profile")
        self.global_step = int(self.step_dev[0].item())
        return {"grad_span": out[0], "allreduce_window": out[1], "adam_span": out[2], "gap_to_next_step": out[3], "step": out[4],
                "grad_launches": [names.raw[i * 48:(i + 1) * 48].split(b"\0")[0].decode() for i in range(n.value)]}

    def profile_levels(self, x, iters: int = 20):
        """Per-launch timing of the step with hipEvents (gmvae_step_profile)."""
        x = self._prep_x(x)
        d, ws = self._workspace(x.shape[0])
        n = C.c_int()
        names = C.create_string_buffer(96 * 48)
        usec = (C.c_float * 96)()
        flops = (C.c_double * 96)()
        rc = L.lib.gmvae_step_profile(C.byref(d), self.model, L.ptr(x), None, None, L.ptr(self.params),
                                      L.ptr(self.grads), L.ptr(ws), self.noise_seed, iters, 96, C.byref(n), names,
                                      usec, flops, L.current_stream())
        L.check(rc, "gmvae_step_profile")
        out = []
        for i in range(n.value):
            nm = names.raw[i * 48:(i + 1) * 48].split(b"\0")[0].decode()
            out.append((nm, float(usec[i]), float(flops[i])))
        return out

    def profile_skinny_levels(self, x, lr: float = 1e-3, n_steps: int = 8, launches: int = 60):
        """Per-launch durations of the skinny schedule (csrc/skinny.hpp) inside a replayed train graph, from the device
        wall-clock stamps its kernels leave (gmvae_debug_sk_stamps): [(name, in-kernel span us, None, timeline share us)],
        the share running from the launch's first workgroup start to the next launch's (the last launch: to the next step's
        first).  Returns None when the configuration does not take that schedule.  Advances training by the replayed steps."""
        import numpy as np
        if L.lib.gmvae_debug_sk_stamps(None) != 0:
            return None
        x = self._prep_x(x)
        B = x.shape[0]
        self.drop_graphs(clear_handoff_errors=False)              # graphs captured before the buffer existed do not stamp
        sx, replay = self.capture_train_step(B, lr=lr, n_steps=n_steps)
        sx.copy_(x.unsqueeze(0).expand(n_steps, -1, -1) if n_steps > 1 else x)
        for _ in range(launches):
            replay()
        torch.cuda.synchronize()
        buf = np.zeros(10 * 1024 * 8, np.uint64)
        L.check(L.lib.gmvae_debug_sk_stamps(buf.ctypes.data_as(C.c_void_p)), "gmvae_debug_sk_stamps")
        st = buf.reshape(10, 1024, 8).astype(np.float64)
        names = ["sk_first_layers", "sk_y_path", "sk_q_head_z", "sk_dec_hidden", "sk_dec_bernoulli", "sk_bwd_dhd", "sk_bwd_dz_heads",
                 "sk_bwd_dhg", "sk_y_path_bwd", "sk_dw_adam"]
        order = list(range(10))
        if self.model == L.MODEL_IDS["vae_gmp"]:                   # the mixture prior's launch stamps the (free) slot of the y
            names[8] = "sk_gmp_bwd"                                #  path's reverse and runs behind B2
            order = [0, 1, 2, 3, 4, 5, 6, 8, 7, 9]
        starts, ends, present = [], [], []
        for i in order:
            r = st[i][st[i][:, 0] > 0]
            if not len(r):
                continue                                           # (the VAE has no y path: slots 1 and 8 stay empty)
            present.append(names[i])
            starts.append(r[:, 0].min())
            ends.append(r[:, 3].max())
        n = len(present)
        self.drop_graphs(clear_handoff_errors=False)              # the stamped graphs hold the buffer's address: destroy them,
        L.check(L.lib.gmvae_debug_sk_stamps_free(), "gmvae_debug_sk_stamps_free")      # then disarm (later steps do not stamp)
        if n < 8 or not all(starts[i + 1] > starts[i] for i in range(n - 1)):
            return None                                            # (stamps of different steps: a launch was mid-flight)
        out = []
        for i in range(n):
            share = (starts[i + 1] - starts[i]) * 0.01 if i < n - 1 else None
            out.append([present[i], (ends[i] - starts[i]) * 0.01, None, share])
        return out
