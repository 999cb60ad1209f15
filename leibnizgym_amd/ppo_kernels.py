"""ctypes binding and autograd wrapper of the trainer's hand-written HIP kernel (leibnizgym_amd/csrc/ppo_kernels.hip).

`fused_ppo_loss` evaluates the whole PPO objective and its gradients in ONE launch (the eager form is ~60 elementwise and
reduction launches forwards and backwards).  It launches on torch's current stream, so it is captured into the trainer's HIP
graphs like any other kernel.  GPU only: the CPU tests of the trainer run its plain-torch form, which is also the fp32
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


def fused_ppo_loss(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, e_clip, v_coef, ent_coef, bounds_coef):
    """loss (0-d tensor with grad); `stats` [4] accumulates (loss, a_loss, c_loss, kl) of the call.  All float32, contiguous,
    on one GPU; `act`, `old_nlp`, `adv`, `ret`, `old_mu` are data (no gradient)."""
    return _FusedPPOLoss.apply(mu, log_std, v, act, old_nlp, adv, ret, old_mu, stats, float(e_clip), float(v_coef), float(ent_coef),
                               float(bounds_coef))
