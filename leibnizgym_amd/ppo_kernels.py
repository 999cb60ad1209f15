"""ctypes binding and autograd wrappers of the trainer's hand-written HIP kernels (leibnizgym_amd/csrc/ppo_kernels.hip).

`fused_ppo_loss` evaluates the whole PPO objective and its gradients in ONE launch (the eager form is ~60 elementwise and
reduction launches forwards and backwards); `linear_elu` is `elu(x W^T + b)` whose backward fuses the ELU derivative with the
bias gradient (one launch instead of a multiply, a compare/select and a column reduction) and keeps the batch-parallel weight
gradient of `ppo._SplitKLinear`.  Both launch on torch's current stream, so they are captured into the trainer's HIP graphs
like any other kernel.  GPU only: the CPU tests of the trainer run its plain-torch form, which is also the fp32 reference
the GPU tests compare these kernels with (tests/test_ppo_kernels.py)."""
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
        lib.tfp_elu_bwd_bias.restype = C.c_int
        lib.tfp_elu_bwd_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        lib.tfp_col_sum.restype = C.c_int
        lib.tfp_col_sum.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
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
        d_ls = torch.empty(A, device=mu.device, dtype=torch.float32)
        loss = torch.empty((), device=mu.device, dtype=torch.float32)
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


def fused_ppo_loss(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, e_clip, v_coef, ent_coef, bounds_coef):
    """loss (0-d tensor with grad); `stats` [4] accumulates (loss, a_loss, c_loss, kl) of the call.  All float32, contiguous,
    on one GPU; `act`, `old_nlp`, `adv`, `ret`, `old_mu` are data (no gradient)."""
    return _FusedPPOLoss.apply(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, float(e_clip), float(v_coef), float(ent_coef),
                               float(bounds_coef))


class _LinearELU(torch.autograd.Function):
    SLICES = 16

    @staticmethod
    def forward(ctx, x, w, b):
        y = torch.nn.functional.elu_(torch.addmm(b, x, w.t()))
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        lib = load()
        dz = gy.contiguous().clone()                      # the kernel works in place; autograd owns gy
        db = torch.empty(w.shape[0], device=gy.device, dtype=torch.float32)
        _chk(lib.tfp_elu_bwd_bias(dz.data_ptr(), y.data_ptr(), dz.shape[0], dz.shape[1], db.data_ptr(), _stream(dz)), "tfp_elu_bwd_bias")
        gx = dz @ w if ctx.needs_input_grad[0] else None
        n, s = x.shape[0], _LinearELU.SLICES
        if n % s == 0 and n >= 64 * s:                    # batch-parallel weight gradient (see ppo._SplitKLinear)
            gw = torch.bmm(dz.view(s, n // s, -1).transpose(1, 2), x.view(s, n // s, -1)).sum(0)
        else:
            gw = dz.t() @ x
        return gx, gw, db


def linear_elu(x, w, b):
    """elu(x @ w.T + b) for float32 [M, K] x [N, K] on the GPU, with the fused ELU-derivative + bias-gradient backward"""
    return _LinearELU.apply(x, w, b)
