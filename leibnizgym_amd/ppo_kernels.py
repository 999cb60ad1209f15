"""ctypes binding and autograd wrapper of the trainer's hand-written HIP kernel (leibnizgym_amd/csrc/ppo_kernels.hip).

`fused_ppo_loss` evaluates the whole PPO objective and its gradients in ONE launch (the eager form is ~60 elementwise and
reduction launches forwards and backwards).  It launches on torch's current stream.  GPU only: the CPU tests of the trainer run its plain-torch form, which is also the fp32
reference the GPU tests compare the kernel with (tests/test_ppo_kernels.py).

Tried and dropped (measured on MI355X, batch 8192): a fused ELU-derivative + bias-gradient backward kernel - the column sums
need either ~50 k atomics on 400 addresses (28 us, twice torch's two launches) or a second pass; not worth 2 % of the step."""
import ctypes as C
import os

import torch

_LIB = None
_F = C.POINTER(C.c_float)


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtrifinger_ppo.so")


def load():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C leibnizgym_amd/csrc` (python __graft_entry__.py)")
        lib = C.CDLL(path)
        lib.tfp_api_version.restype = C.c_int
        lib.tfp_ppo_loss.restype = C.c_int
        lib.tfp_ppo_loss.argtypes = [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 6
        lib.tfp_clip_adam.restype = C.c_int
        lib.tfp_clip_adam.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32] + [C.c_void_p] * 3 + [C.c_float] * 5 + [C.c_void_p]
        lib.tfp_linear_fwd.restype = C.c_int
        lib.tfp_linear_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
        lib.tfp_gemm_nn.restype = C.c_int
        lib.tfp_gemm_nn.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p]
        lib.tfp_gemm_nn_dz.restype = C.c_int
        lib.tfp_gemm_nn_dz.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p]
        lib.tfp_gemm_nn_dz_group.restype = C.c_int
        lib.tfp_gemm_nn_dz_group.argtypes = [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]
        lib.tfp_gemm_tn_bias.restype = C.c_int
        lib.tfp_gemm_tn_bias.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_void_p]
        lib.tfp_gemm_tn_partials.restype = C.c_int
        lib.tfp_gemm_tn_partials.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
        lib.tfp_linear_fwd_group.restype = C.c_int
        lib.tfp_linear_fwd_group.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_void_p]
        lib.tfp_gemm_nn_group.restype = C.c_int
        lib.tfp_gemm_nn_group.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p]
        lib.tfp_gemm_tn_partials_group.restype = C.c_int
        lib.tfp_gemm_tn_partials_group.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_void_p]
        lib.tfp_sum_partials_multi.restype = C.c_int
        lib.tfp_sum_partials_multi.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_void_p]
        lib.tfp_rollout_record.restype = C.c_int
        lib.tfp_rollout_record.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32, C.c_int32] + [C.c_void_p] * 7
        lib.tfp_rollout_reward.restype = C.c_int
        lib.tfp_rollout_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.tfp_gae.restype = C.c_int
        lib.tfp_gae.argtypes = [C.c_void_p] * 3 + [C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.tfp_gather_rows.restype = C.c_int
        lib.tfp_gather_rows.argtypes = [C.c_void_p] * 3 + [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        lib.tfp_gemm_tn_partials_direct_chunk.restype = C.c_int
        lib.tfp_gemm_tn_partials_direct_chunk.argtypes = []
        lib.tfp_gemm_tn_partials_direct.restype = C.c_int
        lib.tfp_gemm_tn_partials_direct.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_void_p]
        lib.tfp_reset_state.restype = C.c_int
        lib.tfp_reset_state.argtypes = [C.c_void_p]
        for name in ("tfp_mlp_forward", "tfp_mlp_backward"):
            getattr(lib, name).restype = C.c_int
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        _LIB = lib
    return _LIB


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}")


class _FusedPPOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, e_clip, v_coef, ent_coef, bounds_coef):
        lib = load()
        mu, v = mu.contiguous(), v.contiguous()
        B, A = mu.shape
        d_mu, d_v = torch.empty_like(mu), torch.empty_like(v)
        out = torch.empty(A + 1, device=mu.device, dtype=torch.float32)      # d_logstd [A] and the loss, zeroed by one fill
        d_ls, loss = out[:A], out[A]
        _chk(lib.tfp_ppo_loss(mu.data_ptr(), log_std.data_ptr(), act.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(), old_mu.data_ptr(),
                              v.data_ptr(), ret.data_ptr(), B, A, e_clip, v_coef, ent_coef, bounds_coef,
                              d_mu.data_ptr(), d_v.data_ptr(), d_ls.data_ptr(), loss.data_ptr(), stats.data_ptr(), _stream(mu)),
             "tfp_ppo_loss")
        ctx.save_for_backward(d_mu, d_v, d_ls)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_mu, d_v, d_ls = ctx.saved_tensors
        return d_mu * g, d_ls * g, d_v * g, None, None, None, None, None, None, None, None, None, None


def ppo_loss_and_grads(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, e_clip, v_coef, ent_coef, bounds_coef, d_ls_out=None):
    """(loss, d loss / d mu, d loss / d v, d loss / d log_std) straight from the kernel - for a caller that starts the backward pass at
    the network outputs itself (`torch.autograd.backward((mu, v), (d_mu, d_v))`), without a loss node multiplying them by one"""
    lib = load()
    mu, v = mu.contiguous(), v.contiguous()
    B, A = mu.shape
    d_mu, d_v = torch.empty_like(mu), torch.empty_like(v)
    out = torch.empty(A + 1, device=mu.device, dtype=torch.float32)
    d_ls, loss = (out[:A] if d_ls_out is None else d_ls_out), out[A]      # d_ls_out: e.g. the parameter's slot of a flat gradient buffer
    _chk(lib.tfp_ppo_loss(mu.data_ptr(), log_std.data_ptr(), act.data_ptr(), old_nlp.data_ptr(), adv.data_ptr(), old_mu.data_ptr(),
                          v.data_ptr(), ret.data_ptr(), B, A, float(e_clip), float(v_coef), float(ent_coef), float(bounds_coef),
                          d_mu.data_ptr(), d_v.data_ptr(), d_ls.data_ptr(), loss.data_ptr(), stats.data_ptr(), _stream(mu)), "tfp_ppo_loss")
    return loss, d_mu, d_v, d_ls


def reset_state(device="cuda:0"):
    """the zero state of the objective kernel's accumulators (only after a launch of it was aborted: a completed one leaves them clean)"""
    _chk(load().tfp_reset_state(C.c_void_p(torch.cuda.current_stream(torch.device(device)).cuda_stream)), "tfp_reset_state")


def fused_ppo_loss(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, e_clip, v_coef, ent_coef, bounds_coef):
    """loss (0-d tensor with grad); `stats` [4] accumulates (loss, a_loss, c_loss, kl) of the call.  All float32, contiguous,
    on one GPU; `act`, `old_nlp`, `adv`, `ret`, `old_mu` are data (no gradient)."""
    return _FusedPPOLoss.apply(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, float(e_clip), float(v_coef), float(ent_coef),
                               float(bounds_coef))


class FlatClipAdam:
    """Gradient-norm truncation + Adam for two parameter groups over ONE flat buffer, two hand-written launches per step
    (csrc/ppo_kernels.hip: tfp_clip_adam) instead of torch's multi-tensor clip + two fused-Adam launches (~110 us for the 32
    small tensors of the two MLPs).  Same arithmetic as `torch.nn.utils.clip_grad_norm_` per group followed by `torch.optim.Adam`
    (no weight decay, no amsgrad).  The parameters are re-pointed to views of the flat buffer (their values are kept); learning
    rates and the step counter live on the device (no host value enters a launch)."""

    def __init__(self, group0, group1, lr0, lr1, max_norm0, max_norm1, betas=(0.9, 0.999), eps=1e-8):
        params = list(group0) + list(group1)
        dev = params[0].device
        self.params = params
        # every parameter starts on a 16-byte boundary of the flat buffer (the forward GEMM reads weight rows with dwordx4 loads); the
        # padding floats are zero, receive zero gradients and are never read by the networks
        al = lambda n: (n + 3) & ~3  # noqa: E731
        self.offsets, off = [], 0
        for k, p in enumerate(params):
            if k == len(list(group0)):
                self.n0 = off
            self.offsets.append(off)
            off += al(p.numel())
        if not list(group1):
            self.n0 = off
        self.n1 = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(params, self.offsets):             # parameters become views of the flat buffer
            self.flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[o:o + p.numel()].view_as(p)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.m, self.v = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
        self.step_count = torch.zeros(2, device=dev)            # (this step, completed steps): include/trifinger_ppo.h
        self.lr = torch.tensor([float(lr0), float(lr1)], device=dev)
        self.sq = torch.zeros(4, device=dev)                    # two halves used by alternate steps; the kernels keep the next one clear
        self.max_norms, self.betas, self.eps = (float(max_norm0), float(max_norm1)), betas, float(eps)

    def set_lr(self, group, value):
        self.lr[group] = float(value)

    def grad_view(self, p):
        """the slot of parameter `p` in the flat gradient buffer, shaped like `p` (the MFMA linear layers write their weight and bias
        gradients straight into it)"""
        k = next(i for i, q in enumerate(self.params) if q is p)
        return self.flat_g[self.offsets[k]:self.offsets[k] + p.numel()].view_as(p)

    def gather_grads(self):
        """gradients that autograd left in `.grad` -> their slots (one multi-tensor copy); the layers that wrote their slots
        themselves have `.grad` None"""
        dst, src = [], []
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None:
                dst.append(self.flat_g[o:o + p.numel()])
                src.append(p.grad.reshape(-1))
        if dst:
            torch._foreach_copy_(dst, src)
        return self.flat_g

    def step(self, gathered=False):
        if not gathered:
            self.gather_grads()
        _chk(load().tfp_clip_adam(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.n0, self.n1,
                                  self.sq.data_ptr(), self.step_count.data_ptr(), self.lr.data_ptr(), self.max_norms[0], self.max_norms[1],
                                  self.betas[0], self.betas[1], self.eps, _stream(self.flat_p)), "tfp_clip_adam")

    def snapshot(self):
        return self.m.clone(), self.v.clone(), self.step_count.clone()

    def restore_snapshot(self, snap):
        self.m.copy_(snap[0]); self.v.copy_(snap[1]); self._set_step(snap[2])

    def state_dict(self):
        return {"kind": "flat_clip_adam", "m": self.m.clone(), "v": self.v.clone(), "step": self.step_count[1:2].clone(), "lr": self.lr.clone()}

    def _set_step(self, t):
        """continue from `t` completed steps (a 1-element tensor, or the two-slot counter of a snapshot)"""
        self.step_count.fill_(float(t.reshape(-1)[-1]))
        self.sq.zero_()

    def load_state_dict(self, sd):
        if sd.get("kind") != "flat_clip_adam" or sd["m"].numel() != self.m.numel():
            raise ValueError("optimizer state of another kind / size")
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self._set_step(sd["step"]); self.lr.copy_(sd["lr"])


# ---- fp32 MFMA GEMMs of the two MLPs (csrc/ppo_kernels.hip: k_gemm) --------------------------------------------------------------
def linear_fwd(x, w, b, act):
    """act(x @ w.T + b) for contiguous float32 x [M, K], w [N, K], b [N]; act: 0 none, 1 ELU"""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _chk(load().tfp_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, int(act), _stream(x)), "tfp_linear_fwd")
    return y


def gemm_nn(a, b, y=None, y_out=None):
    """((a * elu'(y)) @ b) * elu'(y_out)  (y: the ELU output the gradient a belongs to, or None; y_out: the ELU output of the layer the product is the
    gradient of, or None - with it the result is that layer's dZ, ready for its own two products without a factor in their operand loads)"""
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _chk(load().tfp_gemm_nn_dz(a.data_ptr(), y.data_ptr() if y is not None else None, b.data_ptr(), y_out.data_ptr() if y_out is not None else None,
                               c.data_ptr(), M, N, K, _stream(a)), "tfp_gemm_nn_dz")
    return c


_PENDING_SUMS = []        # (part, gw, gb, splits, N1, N2) of the layers whose chunk products wait for flush_partial_sums()


def discard_partial_sums():
    """drop whatever is still queued (a launch raised between `mlp_backward` and `flush_partial_sums`)"""
    global _PENDING_SUMS
    _PENDING_SUMS = []


def flush_partial_sums():
    """the chunk sums of every layer queued by `gemm_tn_bias(..., defer=True)` since the last call, in ONE launch per eight layers
    (the trainer calls it once after the backward pass: eight launches become one)"""
    global _PENDING_SUMS
    pend, _PENDING_SUMS = _PENDING_SUMS, []
    for i in range(0, len(pend), 8):
        grp = pend[i:i + 8]
        n = len(grp)
        vp = lambda k: (C.c_void_p * n)(*[g[k].data_ptr() for g in grp])     # noqa: E731
        ip = lambda k: (C.c_int32 * n)(*[g[k] for g in grp])                 # noqa: E731
        _chk(load().tfp_sum_partials_multi(vp(0), vp(1), vp(2), ip(3), ip(4), ip(5), n, _stream(grp[0][0])), "tfp_sum_partials_multi")


def gather_rows(srcs, idx, outs=None):
    """[s[idx] for s in srcs] for up to 8 row-major float32 arrays (1-D arrays count as width 1) in ONE launch"""
    assert 0 < len(srcs) <= 8 and idx.dtype == torch.long and idx.is_contiguous()
    rows = idx.numel()
    widths = [int(s[0].numel()) for s in srcs]
    if outs is None:
        outs = [torch.empty((rows,) + tuple(s.shape[1:]), device=s.device, dtype=torch.float32) for s in srcs]
    n = len(srcs)
    _chk(load().tfp_gather_rows((C.c_void_p * n)(*[s.data_ptr() for s in srcs]), (C.c_void_p * n)(*[o.data_ptr() for o in outs]),
                                (C.c_int32 * n)(*widths), n, idx.data_ptr(), rows, _stream(idx)), "tfp_gather_rows")
    return outs


# ---- the rollout's bookkeeping (ppo.PPOTrainer.rollout) -----------------------------------------------------------------------------------
def rollout_record(obs, states, mu, log_std, sigma, eps, val, buf, t):
    """a = mu + sigma * eps (sigma = log_std.exp(), formed once per rollout), its negative log-likelihood, and the step filed into slot t of the rollout buffers (obs, states, act, mu, nlp, val) in
    ONE launch; returns the action (a view of buf["act"][t])"""
    n, A = mu.shape
    f = lambda x: x if x.is_contiguous() else x.contiguous()          # noqa: E731
    obs, mu, eps, val = f(obs), f(mu), f(eps), f(val)
    states = f(states) if states is not None else None
    _chk(load().tfp_rollout_record(obs.data_ptr(), obs.shape[1], states.data_ptr() if states is not None else None, states.shape[1] if states is not None else 0,
                                   mu.data_ptr(), log_std.data_ptr(), sigma.data_ptr(), eps.data_ptr(), val.data_ptr(), n, A, buf["obs"][t].data_ptr(),
                                   buf["states"][t].data_ptr() if states is not None else None, buf["act"][t].data_ptr(), buf["mu"][t].data_ptr(),
                                   buf["nlp"][t].data_ptr(), buf["val"][t].data_ptr(), _stream(mu)), "tfp_rollout_record")
    return buf["act"][t]


def rollout_reward(r, d, scale, rew_t, done_t):
    """rew_t = r * scale, done_t = float(d) in one launch; d: torch.bool / uint8"""
    assert d.dtype in (torch.bool, torch.uint8) and d.is_contiguous() and r.is_contiguous() and r.dtype == torch.float32
    _chk(load().tfp_rollout_reward(r.data_ptr(), d.data_ptr(), float(scale), r.numel(), rew_t.data_ptr(), done_t.data_ptr(), _stream(r)), "tfp_rollout_reward")


def gae(rew, done, val, gamma, tau):
    """(adv, ret) over the horizon in one launch: rew, done [T, n], val [T + 1, n] -> adv, ret [T, n]; the arithmetic of the backward loop it replaces"""
    T, n = rew.shape
    adv, ret = torch.empty_like(rew), torch.empty_like(rew)
    _chk(load().tfp_gae(rew.data_ptr(), done.data_ptr(), val.data_ptr(), float(gamma), float(gamma * tau), T, n, adv.data_ptr(), ret.data_ptr(), _stream(rew)), "tfp_gae")
    return adv, ret


def gemm_tn_bias(a, b, y=None, chunk=256, out=None, defer=False):
    """dz = a * elu'(y) (or a); returns (dz.T @ b, dz.sum(0)): products of [b | 1] over row chunks (one workgroup set per chunk),
    then their sum in a fixed order - the bias gradient is the extra column"""
    rows, N1 = a.shape
    N2 = b.shape[1]
    splits = (rows + chunk - 1) // chunk
    part = torch.empty(splits * N1 * (N2 + 1), device=a.device, dtype=torch.float32)
    if out is not None:
        gw, gb = out
    else:
        gw = torch.empty(N1, N2, device=a.device, dtype=torch.float32)
        gb = torch.empty(N1, device=a.device, dtype=torch.float32)
    if defer:                                             # products now, the chunk sums with the other layers' in flush_partial_sums()
        _chk(load().tfp_gemm_tn_partials(a.data_ptr(), y.data_ptr() if y is not None else None, b.data_ptr(), part.data_ptr(), rows, N1, N2, chunk,
                                         _stream(a)), "tfp_gemm_tn_partials")
        _PENDING_SUMS.append((part, gw, gb, splits, N1, N2))
        return gw, gb
    _chk(load().tfp_gemm_tn_bias(a.data_ptr(), y.data_ptr() if y is not None else None, b.data_ptr(), part.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                 rows, N1, N2, chunk, _stream(a)), "tfp_gemm_tn_bias")
    return gw, gb


# ---- the same products for several independent problems in one launch (tfp_*_group) -------------------------------------------------
def _vp(ts):
    return (C.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def _ip(vals):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


def linear_fwd_group(xs, ws, bs, act):
    """[act(x @ w.T + b)] for up to 8 independent (x, w, b) in ONE launch; None when the problems are not of one kind (the caller falls back)"""
    n = len(xs)
    ys = [torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32) for x, w in zip(xs, ws)]
    rc = load().tfp_linear_fwd_group(_vp(xs), _vp(ws), _vp(bs), _vp(ys), _ip([x.shape[0] for x in xs]), _ip([w.shape[0] for w in ws]),
                                     _ip([x.shape[1] for x in xs]), int(act), n, _stream(xs[0]))
    if rc == -4:
        return None
    _chk(rc, "tfp_linear_fwd_group")
    return ys


def gemm_nn_group(as_, bs, ys=None, y_outs=None):
    """[((a * elu'(y)) @ b) * elu'(y_out)] for up to 8 independent problems in ONE launch (ys: all given or None; y_outs: a list with None entries
    allowed, or None); None when they are not of one kind"""
    n = len(as_)
    cs = [torch.empty(a.shape[0], b.shape[1], device=a.device, dtype=torch.float32) for a, b in zip(as_, bs)]
    eo = None
    if y_outs is not None and any(e is not None for e in y_outs):
        eo = (C.c_void_p * n)(*[e.data_ptr() if e is not None else None for e in y_outs])
    rc = load().tfp_gemm_nn_dz_group(_vp(as_), _vp(ys) if ys is not None else None, _vp(bs), eo, _vp(cs), _ip([a.shape[0] for a in as_]),
                                     _ip([b.shape[1] for b in bs]), _ip([a.shape[1] for a in as_]), n, _stream(as_[0]))
    if rc == -4:
        return None
    _chk(rc, "tfp_gemm_nn_dz_group")
    return cs


def gemm_tn_bias_group(as_, bs, ys, outs, chunk=256):
    """the chunk products of up to 8 weight / bias gradients in ONE launch, their sums deferred to flush_partial_sums() (as gemm_tn_bias(defer=True));
    ys: all given or None.  Returns False when the problems are not of one kind (nothing is queued then)"""
    n = len(as_)
    rows = [a.shape[0] for a in as_]
    n1 = [a.shape[1] for a in as_]
    n2 = [b.shape[1] for b in bs]
    parts = [torch.empty(((r + chunk - 1) // chunk) * a1 * (b2 + 1), device=as_[0].device, dtype=torch.float32) for r, a1, b2 in zip(rows, n1, n2)]
    rc = load().tfp_gemm_tn_partials_group(_vp(as_), _vp(ys) if ys is not None else None, _vp(bs), _vp(parts), _ip(rows), _ip(n1), _ip(n2), int(chunk), n,
                                           _stream(as_[0]))
    if rc == -4:
        return False
    _chk(rc, "tfp_gemm_tn_partials_group")
    for k in range(n):
        _PENDING_SUMS.append((parts[k], outs[k][0], outs[k][1], (rows[k] + chunk - 1) // chunk, n1[k], n2[k]))
    return True


def gemm_tn_bias_direct(as_, bs, outs):
    """the chunk slabs of up to 8 weight / bias gradients [a.T @ b | a.sum(0)] in ONE launch of the direct kernel (csrc/ppo_dw_direct.hip: 64 x 64 blocks,
    operands straight from memory), their sums deferred to flush_partial_sums().  a: dZ [rows, N1] (plain: no activation factor), b: [rows, N2].
    Returns False when the call does not fit (nothing is queued then)."""
    n = len(as_)
    as_ = [a if a.is_contiguous() else a.contiguous() for a in as_]
    bs = [b if b.is_contiguous() else b.contiguous() for b in bs]
    rows = [a.shape[0] for a in as_]
    n1 = [a.shape[1] for a in as_]
    n2 = [b.shape[1] for b in bs]
    chunk = load().tfp_gemm_tn_partials_direct_chunk()          # rows per slab of the build
    splits = [(r + chunk - 1) // chunk for r in rows]
    parts = [torch.empty(sp * a1 * (b2 + 1), device=as_[0].device, dtype=torch.float32) for sp, a1, b2 in zip(splits, n1, n2)]
    rc = load().tfp_gemm_tn_partials_direct(_vp(as_), _vp(bs), _vp(parts), _ip(rows), _ip(n1), _ip(n2), n, _stream(as_[0]))
    if rc == -4:
        return False
    _chk(rc, "tfp_gemm_tn_partials_direct")
    for k in range(n):
        _PENDING_SUMS.append((parts[k], outs[k][0], outs[k][1], splits[k], n1[k], n2[k]))
    return True


USE_DIRECT_DW = True


class _MfmaLinear(torch.autograd.Function):
    """act(x W^T + b) with every matrix product on the hand-written fp32 MFMA kernels: forward with bias and ELU fused into the
    store; backward with the ELU derivative formed in the operand loads and the bias gradient as an extra column of the weight
    gradient product - three launches (+ one reduction of the batch chunks) where the eager form takes nine.  With `grad_out` =
    (dW buffer, db buffer) the parameter gradients are WRITTEN there (not accumulated, not returned to autograd) - complete only
    after `flush_partial_sums()`, which sums the row chunks of every such layer of the backward pass in one launch: the trainer
    passes the parameters' slots of its flat gradient buffer and flushes once per minibatch step."""

    @staticmethod
    def forward(ctx, x, w, b, act, grad_out):
        x = x.contiguous()
        y = linear_fwd(x, w, b, act)
        ctx.act, ctx.grad_out = act, grad_out
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        yy = y if ctx.act else None
        gx = gemm_nn(gy, w, yy) if ctx.needs_input_grad[0] else None
        if ctx.grad_out is not None:                      # the trainer's path: sums deferred to one launch after the backward pass
            gemm_tn_bias(gy, x, yy, out=ctx.grad_out, defer=True)
            return gx, None, None, None, None
        gw, gb = gemm_tn_bias(gy, x, yy)
        return gx, gw, gb, None, None


def mfma_linear(x, w, b, act, grad_out=None):
    return _MfmaLinear.apply(x, w, b, int(act), grad_out)


# ---- the two MLPs without autograd: forward keeps the layer outputs, backward walks them in reverse --------------------------------
def mlp_forward(x, layers):
    """`layers`: [(weight, bias, act, grad_out)] in order.  Returns the list of layer outputs (the last one is the network output);
    the inputs of the layers are `x` and the outputs before them."""
    ys = []
    for w, b, act, _ in layers:
        x = linear_fwd(x, w, b, act)
        ys.append(x)
    return ys


def mlp_backward(x, ys, gy, layers):
    """gradients of every layer's weight and bias into its `grad_out` buffers (chunk sums deferred: call flush_partial_sums() after the
    last network), given the gradient `gy` of the network output; the input gradient of the first layer is not formed"""
    # gy is the gradient of layer k's OUTPUT at the top of the walk and the dZ of layer k - already times elu'(ys[k]) - below it: every input-gradient
    # product leaves multiplied by the activation derivative of the layer it is the gradient of (tfp_gemm_nn_dz), so the two products that consume it
    # stage plain operands.  Same bits as multiplying in their operand loads (one fp32 product either way).
    is_dz = False
    for k in range(len(layers) - 1, -1, -1):
        w, _, act, grad_out = layers[k]
        inp = ys[k - 1] if k > 0 else x
        yy = ys[k] if (act and not is_dz) else None
        gemm_tn_bias(gy, inp, yy, out=grad_out, defer=True)
        if k > 0:
            gy = gemm_nn(gy, w, yy, y_out=ys[k - 1] if layers[k - 1][2] else None)
            is_dz = True


# ---- the network walk (csrc/ppo_mlp_walk.hip): all layers of up to two networks in ONE launch per direction ------------------------------------
WALK_MAXL = 4


class TfpMlp(C.Structure):
    """include/trifinger_ppo.h: TfpMlp"""
    _fields_ = [("x", C.c_void_p), ("W", C.c_void_p * WALK_MAXL), ("b", C.c_void_p * WALK_MAXL), ("yin", C.c_void_p * WALK_MAXL),
                ("y", C.c_void_p * WALK_MAXL), ("dim", C.c_int32 * (WALK_MAXL + 1)), ("act", C.c_int32 * WALK_MAXL), ("n_layers", C.c_int32)]


def _walk_struct(x, layers):
    m = TfpMlp()
    m.x = x.data_ptr()
    m.n_layers = len(layers)
    m.dim[0] = layers[0][0].shape[1]
    for l, (w, b, act, _) in enumerate(layers):
        m.W[l] = w.data_ptr()
        m.b[l] = b.data_ptr() if b is not None else None
        m.dim[l + 1] = w.shape[0]
        m.act[l] = int(act)
    return m


def _walkable(nets):
    return (0 < len(nets) <= 2 and all(0 < len(layers) <= WALK_MAXL for _, layers in nets)
            and all(x.dim() == 2 and x.is_contiguous() and x.shape[0] == nets[0][0].shape[0] for x, _ in nets))


def mlp_walk_forward(nets, store_hidden=True):
    """nets: [(x, layers)] for one or two Linear / ELU stacks over the same rows (`layers` as in mlp_forward).  ONE launch; returns the list of layer
    outputs per network (hidden outputs None with store_hidden = False: the rollout needs the network outputs only), or None when the shapes do not
    fit the walk (the caller then runs the layers one by one)."""
    if not _walkable(nets):
        return None
    M = nets[0][0].shape[0]
    arr = (TfpMlp * len(nets))()
    outs = []
    for i, (x, layers) in enumerate(nets):
        m = _walk_struct(x, layers)
        ys = []
        for l, (w, _, _, _) in enumerate(layers):
            keep = store_hidden or l == len(layers) - 1
            y = torch.empty(M, w.shape[0], device=x.device, dtype=torch.float32) if keep else None
            m.y[l] = y.data_ptr() if keep else None
            ys.append(y)
        arr[i] = m
        outs.append(ys)
    rc = load().tfp_mlp_forward(C.cast(arr, C.c_void_p), len(nets), M, _stream(nets[0][0]))
    if rc == -4:
        return None
    _chk(rc, "tfp_mlp_forward")
    return outs


def mlp_walk_backward(nets):
    """nets: [(gy, ys, layers)]: gy = gradient of the network output, ys = the layer outputs the forward kept.  ONE launch for the whole input-gradient
    chain; returns per network the list dz with dz[l] = gradient of layer l's pre-activation (dz[-1] is gy itself), or None when the shapes do not fit."""
    if not (0 < len(nets) <= 2) or any(len(layers) > WALK_MAXL or any(y is None for y in ys) for _, ys, layers in nets):
        return None
    M = nets[0][0].shape[0]
    arr = (TfpMlp * len(nets))()
    outs = []
    for i, (gy, ys, layers) in enumerate(nets):
        gy = gy if gy.is_contiguous() else gy.contiguous()
        m = _walk_struct(gy, layers)
        dz = []
        for l in range(len(layers) - 1):
            d = torch.empty_like(ys[l])
            m.y[l] = d.data_ptr()
            m.yin[l] = ys[l].data_ptr()
            dz.append(d)
        dz.append(gy)
        arr[i] = m
        outs.append(dz)
    rc = load().tfp_mlp_backward(C.cast(arr, C.c_void_p), len(nets), M, _stream(nets[0][0]))
    if rc == -4:
        return None
    _chk(rc, "tfp_mlp_backward")
    return outs


USE_WALK = True       # tools / tests switch it off to time or check the per-layer launches


# ---- the actor and the central value network side by side: layer k of both in one launch, all weight gradients in two ----------------------
def _pairable(la, lc):
    return len(la) == len(lc) and all(a[2] == c[2] for a, c in zip(la, lc))


def mlp_forward_pair(xa, la, xc, lc, store_hidden=True):
    """mlp_forward of two networks of the same depth and activations, one grouped launch per layer (4 launches instead of 8 for the trainer's MLPs)"""
    if USE_WALK:
        out = mlp_walk_forward([(xa, la), (xc, lc)], store_hidden)
        if out is not None:
            return out[0], out[1]
    if not _pairable(la, lc):
        return mlp_forward(xa, la), mlp_forward(xc, lc)
    ya, yc = [], []
    for (wa, ba, act, _), (wc, bc, _, _) in zip(la, lc):
        out = linear_fwd_group([xa, xc], [wa, wc], [ba, bc], act)
        if out is None:
            out = [linear_fwd(xa, wa, ba, act), linear_fwd(xc, wc, bc, act)]
        xa, xc = out
        ya.append(xa); yc.append(xc)
    return ya, yc


def mlp_backward_pair(xa, ya, gya, la, xc, yc, gyc, lc):
    """mlp_backward of both networks: the input-gradient products of layer k of both in one launch, and ALL weight / bias gradient products of the
    step in two launches (layers with an ELU, layers without) behind them - 5 launches instead of 14 for the trainer's MLPs.  Chunk sums deferred
    as in mlp_backward: call flush_partial_sums() afterwards."""
    if USE_WALK:
        dz = mlp_walk_backward([(gya, ya, la), (gyc, yc, lc)])
        if dz is not None:
            # every dZ is in memory: all weight / bias gradients of the step are plain products dZ_l^T [input_l | 1] = one grouped launch
            gys, inps, outs = [], [], []
            for net_x, net_y, d, layers in ((xa, ya, dz[0], la), (xc, yc, dz[1], lc)):
                for k in range(len(layers) - 1, -1, -1):
                    gys.append(d[k]); inps.append(net_y[k - 1] if k > 0 else net_x); outs.append(layers[k][3])
            for i in range(0, len(gys), 8):
                sl = slice(i, i + 8)
                if USE_DIRECT_DW and gemm_tn_bias_direct(gys[sl], inps[sl], outs[sl]):
                    continue
                if not gemm_tn_bias_group(gys[sl], inps[sl], None, outs[sl]):
                    for g, x, o in zip(gys[sl], inps[sl], outs[sl]):
                        gemm_tn_bias(g, x, None, out=o, defer=True)
            return
    if not _pairable(la, lc):
        mlp_backward(xa, ya, gya, la)
        mlp_backward(xc, yc, gyc, lc)
        return
    dw = {True: ([], [], [], []), False: ([], [], [], [])}          # the operand load multiplies by elu' -> (gy, input, y, grad_out)
    is_dz = False                                                    # as in mlp_backward: below the top layer gy arrives as dZ
    for k in range(len(la) - 1, -1, -1):
        act = la[k][2] and not is_dz
        for net_x, net_y, gy, layers in ((xa, ya, gya, la), (xc, yc, gyc, lc)):
            inp = net_y[k - 1] if k > 0 else net_x
            slot = dw[bool(act)]
            slot[0].append(gy); slot[1].append(inp); slot[2].append(net_y[k] if act else None); slot[3].append(layers[k][3])
        if k > 0:
            yy = [ya[k], yc[k]] if act else None
            eo = [ya[k - 1], yc[k - 1]] if la[k - 1][2] else None
            out = gemm_nn_group([gya, gyc], [la[k][0], lc[k][0]], yy, eo)
            if out is None:
                out = [gemm_nn(gya, la[k][0], ya[k] if act else None, y_out=eo[0] if eo else None),
                       gemm_nn(gyc, lc[k][0], yc[k] if act else None, y_out=eo[1] if eo else None)]
            gya, gyc = out
            is_dz = True
    for has_elu, (gys, inps, ys, outs) in dw.items():
        for i in range(0, len(gys), 8):
            sl = slice(i, i + 8)
            if not gemm_tn_bias_group(gys[sl], inps[sl], ys[sl] if has_elu else None, outs[sl]):
                for g, x, y, o in zip(gys[sl], inps[sl], ys[sl], outs[sl]):
                    gemm_tn_bias(g, x, y, out=o, defer=True)
