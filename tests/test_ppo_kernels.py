"""The trainer's hand-written HIP kernels (leibnizgym_amd/csrc/ppo_kernels.hip) against a plain PyTorch fp32 reference of the
same operations: the fused PPO objective (value and every gradient) and the trainer as a whole with and without it."""
import ctypes as C
import math
import os

import pytest
import torch

from leibnizgym_amd import ppo_kernels as pk
from leibnizgym_amd.ppo import PPOConfig, neglogp


def test_ppo_library_loads_and_exports_its_symbols():
    assert os.path.isfile(pk.library_path()), "run `make -C leibnizgym_amd/csrc` (python __graft_entry__.py)"
    lib = C.CDLL(pk.library_path())
    for name in ("tfp_api_version", "tfp_ppo_loss", "tfp_clip_adam", "tfp_linear_fwd", "tfp_gemm_nn", "tfp_gemm_tn_bias", "tfp_gemm_tn_partials",
                 "tfp_sum_partials_multi", "tfp_gather_rows", "tfp_mlp_forward", "tfp_mlp_backward", "tfp_gemm_tn_partials_direct"):
        assert hasattr(lib, name), name
    assert lib.tfp_api_version() == 3


def reference_loss(mu, ls, v, act, old_nlp, adv, ret, old_mu, e_clip, v_coef, ent_coef, bounds_coef):
    nlp = neglogp(act, mu, ls.expand_as(mu))
    ratio = (old_nlp - nlp).exp()
    a_loss = torch.max(-adv * ratio, -adv * ratio.clamp(1 - e_clip, 1 + e_clip)).mean()
    c_loss = (v - ret).pow(2).mean()
    b_loss = ((mu - 1.1).clamp(min=0).pow(2) + (-1.1 - mu).clamp(min=0).pow(2)).sum(-1).mean()
    ent = (ls + 0.5 + 0.5 * math.log(2 * math.pi)).sum()
    kl = (0.5 * ((mu - old_mu) / ls.exp()).pow(2)).sum(-1).mean()
    return a_loss + v_coef * c_loss - ent_coef * ent + bounds_coef * b_loss, a_loss, c_loss, kl


@pytest.mark.gpu
@pytest.mark.parametrize("B,A,ent_coef", [(8192, 9, 0.0), (1000, 18, 0.01), (77, 9, 0.003)])
def test_fused_objective_matches_torch_fp32(hip, B, A, ent_coef):
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(B + A)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    mu0, ls0, v0 = r(B, A) * 0.8, r(A) * 0.3 - 0.5, r(B)
    act, old_mu = mu0 + r(B, A) * 0.6, mu0 + r(B, A) * 0.05
    adv, ret = r(B), v0 + r(B) * 0.3
    old_nlp = neglogp(act, old_mu, ls0.expand_as(old_mu)) + r(B) * 0.05                        # ratios on both sides of the clip range
    args = dict(e_clip=0.2, v_coef=1.0, ent_coef=ent_coef, bounds_coef=1e-4)
    grads = []
    for fused in (False, True):
        mu, ls, v = (t.clone().requires_grad_(True) for t in (mu0, ls0, v0))
        if fused:
            stats = torch.zeros(4, device=dev)
            loss = pk.fused_ppo_loss(mu, ls, v, act, old_nlp, adv, ret, old_mu, stats, **args)
            got_stats = stats
        else:
            loss, a_loss, c_loss, kl = reference_loss(mu, ls, v, act, old_nlp, adv, ret, old_mu, **args)
            want_stats = torch.stack([loss.detach(), a_loss.detach(), c_loss.detach(), kl.detach()])
        (loss * 1.7).backward()                                                                # a non-trivial upstream gradient
        grads.append((loss.detach(), mu.grad, ls.grad, v.grad))
    (l0, dmu0, dls0, dv0), (l1, dmu1, dls1, dv1) = grads
    assert torch.allclose(l0, l1, rtol=2e-5, atol=1e-6)
    assert torch.allclose(want_stats, got_stats, rtol=2e-5, atol=1e-6)
    assert torch.allclose(dmu0, dmu1, rtol=1e-4, atol=1e-9) and torch.allclose(dv0, dv1, rtol=1e-5, atol=1e-10)
    assert torch.allclose(dls0, dls1, rtol=2e-4, atol=1e-7)
    frac_clipped = float(((old_nlp - neglogp(act, mu0, ls0.expand_as(mu0))).exp() - 1).abs().gt(0.2).float().mean())
    assert 0.02 < frac_clipped < 0.98                                                          # both branches of the surrogate


@pytest.mark.gpu
def test_trainer_with_and_without_the_kernels(hip):
    """two PPO epochs on the HIP env, eager mode: the hand-written kernels take the same optimisation steps as plain torch"""
    from leibnizgym_amd.config import gym_config
    from leibnizgym_amd.envs import TrifingerEnv
    from leibnizgym_amd.ppo import PPOTrainer
    from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter
    from leibnizgym_amd.wrappers import VecTaskPython

    def run(fused):
        cfg = gym_config("trifinger_difficulty_4")
        cfg.update(num_instances=256, seed=1, physics_engine="physx", asymmetric_obs=True, episode_length=20)
        env = TrifingerEnv(config=cfg, device="cuda:0", verbose=False)
        ad = RlGamesGpuEnvAdapter("rlgpu", 256, env=VecTaskPython(env, rl_device="cuda:0"))
        tr = PPOTrainer(ad, 41, 113, 9, PPOConfig(horizon=8, minibatches=4, mini_epochs=2, fused_kernels=fused),
                        device="cuda:0")
        torch.manual_seed(11)
        stats = tr.train(2)
        return [p.detach().clone() for p in tr.net.parameters()], stats
    plain, s0 = run(False)
    fused, s1 = run(True)
    for a, b in zip(plain, fused):
        assert torch.allclose(a, b, atol=3e-5, rtol=1e-3)
    for k in ("loss", "a_loss", "c_loss", "kl"):
        assert abs(s0[-1][k] - s1[-1][k]) < 1e-3 * max(1.0, abs(s0[-1][k])), (k, s0[-1][k], s1[-1][k])


@pytest.mark.gpu
def test_rollout_snapshots_the_observation_of_an_env_that_does_not_promise_stable_buffers(hip):
    """The fused rollout reads obs / states IN PLACE only from an env that declares `buffers_stable_until_next_step` (RlGamesGpuEnvAdapter: the engine's
    own tensors, overwritten by the next step on the same stream).  An env without the attribute may refresh its buffers whenever it likes: the trainer
    must have taken its snapshot when `step` returned.  Emulated here by an env that scribbles NaN over the tensors it handed out as soon as the policy
    is evaluated the next time (i.e. after `step` returned, before the trainer files the step): same rollout as the well-behaved env, no NaN anywhere."""
    from leibnizgym_amd.config import gym_config
    from leibnizgym_amd.envs import TrifingerEnv
    from leibnizgym_amd.ppo import PPOTrainer
    from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter
    from leibnizgym_amd.wrappers import VecTaskPython

    class Unstable:
        """hands out copies held in its own buffers and does NOT promise they stay put"""
        def __init__(self, inner):
            self.inner, self.handed = inner, None

        def _own(self, out):
            self.handed = {k: v.clone() for k, v in out.items()}
            return self.handed

        def reset(self):
            return self._own(self.inner.reset())

        def step(self, a):
            out, r, d, info = self.inner.step(a)
            return self._own(out), r, d, info

        def scribble(self):
            for v in self.handed.values():
                v.fill_(float("nan"))

    def run(unstable):
        cfg = gym_config("trifinger_difficulty_4")
        cfg.update(num_instances=256, seed=1, physics_engine="physx", asymmetric_obs=True, episode_length=20)
        ad = RlGamesGpuEnvAdapter("rlgpu", 256, env=VecTaskPython(TrifingerEnv(config=cfg, device="cuda:0", verbose=False), rl_device="cuda:0"))
        env = Unstable(ad) if unstable else ad
        tr = PPOTrainer(env, 41, 113, 9, PPOConfig(horizon=6, minibatches=3, mini_epochs=1), device="cuda:0")
        if unstable:
            inner = tr.net.dist_and_value

            def hooked(obs, states):
                out = inner(obs, states)              # the trainer's own snapshot is what it passes in ...
                env.scribble()                        # ... the env's buffers change under it right away
                return out
            tr.net.dist_and_value = hooked
        torch.manual_seed(5)
        return tr.rollout()
    good, bad = run(False), run(True)
    for k in ("obs", "states", "act", "rew", "val"):
        assert torch.isfinite(bad[k]).all(), k
        assert torch.equal(good[k], bad[k]), k
    assert RlGamesGpuEnvAdapter.buffers_stable_until_next_step is True


@pytest.mark.gpu
def test_flat_clip_adam_matches_torch(hip):
    """gradient-norm truncation per group + Adam over the flat buffer == clip_grad_norm_ + torch.optim.Adam, 25 steps with
    gradients that exercise both the truncated and the untruncated case and two different learning rates"""
    dev = "cuda:0"
    torch.manual_seed(3)
    shapes0, shapes1 = [(400, 41), (400,), (9, 100), (9,)], [(400, 113), (400,), (1, 100), (1,)]
    init = [torch.randn(*s, device=dev) * 0.1 for s in shapes0 + shapes1]
    a = [torch.nn.Parameter(t.clone()) for t in init]
    b = [torch.nn.Parameter(t.clone()) for t in init]
    ref = torch.optim.Adam([{"params": a[:4], "lr": 3e-4}, {"params": a[4:], "lr": 5e-4}], eps=1e-8)
    flat = pk.FlatClipAdam(b[:4], b[4:], 3e-4, 5e-4, 1.0, 0.5)
    assert all(torch.equal(x, y) for x, y in zip(a, b))                      # re-pointing keeps the values
    for step in range(25):
        scale = 10.0 if step % 3 == 0 else 0.01                              # norm far above / below the thresholds
        grads = [torch.randn_like(p) * scale for p in a]
        for p, q, g in zip(a, b, grads):
            p.grad, q.grad = g.clone(), g.clone()
        if step == 10:
            for grp in ref.param_groups[:1]:
                grp["lr"] = 6.75e-4
            flat.set_lr(0, 6.75e-4)
        torch.nn.utils.clip_grad_norm_(a[:4], 1.0)
        torch.nn.utils.clip_grad_norm_(a[4:], 0.5)
        ref.step()
        flat.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=2e-5, atol=2e-7), float((x - y).abs().max())
    sd = flat.state_dict()
    flat2 = pk.FlatClipAdam([torch.nn.Parameter(t.clone()) for t in init[:4]], [torch.nn.Parameter(t.clone()) for t in init[4:]], 1.0, 1.0, 1.0, 0.5)
    flat2.load_state_dict(sd)
    assert torch.equal(flat2.m, flat.m) and flat2.step_count.tolist() == [25.0, 25.0] and torch.equal(flat2.lr, flat.lr)


@pytest.mark.gpu
def test_step_counter_keeps_alternating_past_2_to_the_23(hip):
    """The float step counter of tfp_clip_adam selects, by its parity, the half of sq a step sums its squared norms into (the other half is cleared for
    the next step).  A float stops counting at 2^24; the counter therefore swings between 2^23 and 2^23 + 1 once it gets there.  Preset just below the
    boundary: over eight steps every step must see the squared norm of ITS gradients only (nothing piles up) and leave the other half zero."""
    dev = "cuda:0"
    p0, p1 = torch.nn.Parameter(torch.ones(1000, device=dev)), torch.nn.Parameter(torch.ones(500, device=dev))
    flat = pk.FlatClipAdam([p0], [p1], 1e-4, 1e-4, 1.0, 1.0)
    flat._set_step(torch.tensor([8388606.0]))                                 # 2^23 - 2 completed steps
    seen = []
    for k in range(8):
        g0, g1 = torch.full((1000,), 0.01 * (k + 1), device=dev), torch.full((500,), 0.02 * (k + 1), device=dev)
        p0.grad, p1.grad = g0, g1
        flat.step()
        t = float(flat.step_count[0])
        par = int(t) & 1
        seen.append(t)
        sq = flat.sq.tolist()
        assert sq[2 * (1 - par)] == 0.0 and sq[2 * (1 - par) + 1] == 0.0, (k, sq)
        assert abs(sq[2 * par] - float((g0 * g0).sum())) <= 1e-5 * float((g0 * g0).sum()), (k, sq)
        assert abs(sq[2 * par + 1] - float((g1 * g1).sum())) <= 1e-5 * float((g1 * g1).sum()), (k, sq)
    assert seen == [8388607.0, 8388608.0, 8388609.0, 8388608.0, 8388609.0, 8388608.0, 8388609.0, 8388608.0], seen
    assert torch.isfinite(p0).all() and float((p0 - 1).abs().max()) > 0            # the parameters kept moving


@pytest.mark.gpu
def test_reset_state_restores_the_objective_accumulators(hip):
    """tfp_reset_state: a no-op on a clean library - the objective gives the same loss before and after"""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    mu, ls, v = r(300, 9), r(9) * 0.1, r(300)
    args = (mu, ls, v, mu + r(300, 9), r(300), r(300), r(300), mu + 0.01 * r(300, 9))
    a = pk.ppo_loss_and_grads(*args, torch.zeros(4, device=dev), 0.2, 1.0, 0.0, 1e-4)[0].clone()
    pk.reset_state(dev)
    b = pk.ppo_loss_and_grads(*args, torch.zeros(4, device=dev), 0.2, 1.0, 0.0, 1e-4)[0].clone()
    assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,act", [(8192, 41, 400, 1), (8192, 400, 200, 1), (8192, 100, 9, 0), (1000, 113, 400, 1), (777, 200, 100, 1),
                                      (65, 100, 1, 0), (64, 32, 64, 1), (3, 5, 7, 1)])
def test_mfma_linear_matches_torch_fp32(hip, M, K, N, act):
    """The fp32 MFMA linear layer (forward with bias + ELU fused; backward with the ELU derivative in the operand loads and the bias
    gradient as the extra column) against torch's linear / elu / autograd in fp32 - MLP shapes of the trainer, ragged sizes, sizes
    below one tile; with and without the gradients written into caller buffers."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    x0, w0, b0, gy = r(M, K), r(N, K) * K ** -0.5, r(N), r(M, N)
    ref = [t.clone().requires_grad_(True) for t in (x0, w0, b0)]
    y_ref = torch.nn.functional.linear(*ref)
    y_ref = torch.nn.functional.elu(y_ref) if act else y_ref
    y_ref.backward(gy)
    close = lambda a, b: torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(b.detach().abs().max()) + 1e-6)   # noqa: E731
    for buffers in (False, True):
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        out = (torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev)) if buffers else None
        y = pk.mfma_linear(x, w, b, act, out)
        y.backward(gy)
        pk.flush_partial_sums()                                                  # caller buffers: the chunk sums are one deferred launch
        assert close(y, y_ref) and close(x.grad, ref[0].grad)
        gw, gb = out if buffers else (w.grad, b.grad)
        if buffers:
            assert w.grad is None and b.grad is None
        assert close(gw, ref[1].grad) and close(gb, ref[2].grad)
    with torch.no_grad():                                                        # inference path of the rollout
        assert close(pk.mfma_linear(x0, w0, b0, act), y_ref)


@pytest.mark.gpu
def test_gather_rows_and_deferred_chunk_sums(hip):
    """The one-launch minibatch gather equals indexing; weight / bias gradients whose chunk sums are deferred to one launch for several
    layers equal the ones summed per layer (bit for bit: same products, same summation order)."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    srcs = [torch.randn(5000, w, device=dev, generator=g) for w in (41, 9, 113)] + [torch.randn(5000, device=dev, generator=g) for _ in range(3)]
    idx = torch.randperm(5000, device=dev, generator=g)[:1777].contiguous()
    for got, src in zip(pk.gather_rows(srcs, idx), srcs):
        assert torch.equal(got, src[idx])
    layers = [(torch.randn(2048, n1, device=dev, generator=g), torch.randn(2048, n2, device=dev, generator=g), torch.rand(2048, n1, device=dev, generator=g) - 0.3)
              for n1, n2 in ((400, 41), (200, 400), (9, 100), (1, 100))]
    direct = [pk.gemm_tn_bias(a, b, y) for a, b, y in layers]
    deferred = [pk.gemm_tn_bias(a, b, y, defer=True) for a, b, y in layers]
    pk.flush_partial_sums()
    for (gw0, gb0), (gw1, gb1) in zip(direct, deferred):
        assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1)


@pytest.mark.gpu
def test_grouped_launches_equal_the_single_products_bit_for_bit(hip):
    """tfp_linear_fwd_group / tfp_gemm_nn_group / tfp_gemm_tn_partials_group (several independent products in ONE launch: the trainer's two networks side
    by side, all weight gradients of a step together) against the single-product entry points: the same tiles in the same order, so the results are
    identical bit for bit - at the trainer's layer shapes (minibatch 8192, obs 41 / states 113, MLP [400, 200, 100], 9 actions / 1 value) and ragged
    ones; a mixed-kind group is refused (the Python wrapper then falls back)."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(17)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    M = 8192
    # forward: layer k of actor and critic
    for (ka, kc, n_a, n_c, act) in ((41, 113, 400, 400, 1), (400, 400, 200, 200, 1), (100, 100, 9, 1, 0), (200, 200, 100, 100, 1)):
        xs, ws, bs = [r(M, ka), r(M, kc)], [r(n_a, ka), r(n_c, kc)], [r(n_a), r(n_c)]
        got = pk.linear_fwd_group(xs, ws, bs, act)
        assert got is not None
        for y, x, w, b in zip(got, xs, ws, bs):
            assert torch.equal(y, pk.linear_fwd(x, w, b, act))
    assert pk.linear_fwd_group([r(64, 41), r(64, 400)], [r(8, 41), r(8, 400)], [r(8), r(8)], 1) is None          # scalar-load and vector-load kinds do not mix
    # input gradients: dZ W for both networks, with and without the ELU derivative; a ragged pair
    for (ka, kc, n_a, n_c, elu) in ((200, 200, 400, 400, True), (9, 1, 100, 100, False), (100, 100, 200, 200, True)):
        as_, bs = [r(M, ka), r(M, kc)], [r(ka, n_a), r(kc, n_c)]
        ys = [torch.rand(M, ka, device=dev, generator=g) - 0.3, torch.rand(M, kc, device=dev, generator=g) - 0.3] if elu else None
        got = pk.gemm_nn_group(as_, bs, ys)
        assert got is not None
        for k in range(2):
            assert torch.equal(got[k], pk.gemm_nn(as_[k], bs[k], ys[k] if elu else None))
    # weight / bias gradients: the six hidden-layer products of a step in one launch, the two output-layer ones in another
    hidden = [(400, 41), (200, 400), (100, 200), (400, 113), (200, 400), (100, 200)]
    as_, bs = [r(M, n1) for n1, _ in hidden], [r(M, n2) for _, n2 in hidden]
    ys = [torch.rand(M, n1, device=dev, generator=g) - 0.3 for n1, _ in hidden]
    outs = [(torch.empty(n1, n2, device=dev), torch.empty(n1, device=dev)) for n1, n2 in hidden]
    assert pk.gemm_tn_bias_group(as_, bs, ys, outs)
    pk.flush_partial_sums()
    for a, b, y, (gw, gb) in zip(as_, bs, ys, outs):
        gw0, gb0 = pk.gemm_tn_bias(a, b, y)
        assert torch.equal(gw, gw0) and torch.equal(gb, gb0)
    outl = [(9, 100), (1, 100)]
    as_, bs = [r(777, n1) for n1, _ in outl], [r(777, n2) for _, n2 in outl]
    outs = [(torch.empty(n1, n2, device=dev), torch.empty(n1, device=dev)) for n1, n2 in outl]
    assert pk.gemm_tn_bias_group(as_, bs, None, outs)
    pk.flush_partial_sums()
    for a, b, (gw, gb) in zip(as_, bs, outs):
        gw0, gb0 = pk.gemm_tn_bias(a, b, None)
        assert torch.equal(gw, gw0) and torch.equal(gb, gb0)


@pytest.mark.gpu
def test_paired_network_walks_equal_the_separate_ones(hip, monkeypatch):
    """mlp_forward_pair / mlp_backward_pair (what the trainer's minibatch step runs) against mlp_forward / mlp_backward network by network: identical
    outputs and identical gradients in the caller's buffers."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(23)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731

    def net(k_in, n_out):
        dims = [k_in, 400, 200, 100, n_out]
        return [(r(dims[i + 1], dims[i]) * dims[i] ** -0.5, r(dims[i + 1]), 1 if i < 3 else 0,
                 (torch.zeros(dims[i + 1], dims[i], device=dev), torch.zeros(dims[i + 1], device=dev))) for i in range(4)]
    la, lc = net(41, 9), net(113, 1)
    xa, xc, gya, gyc = r(8192, 41), r(8192, 113), r(8192, 9), r(8192, 1)
    monkeypatch.setattr(pk, "USE_WALK", False)               # the per-layer grouped launches (what runs when the shapes do not fit the network walk)
    ya, yc = pk.mlp_forward_pair(xa, la, xc, lc)
    for got, want in zip(ya + yc, pk.mlp_forward(xa, la) + pk.mlp_forward(xc, lc)):
        assert torch.equal(got, want)
    pk.mlp_backward_pair(xa, ya, gya, la, xc, yc, gyc, lc)
    pk.flush_partial_sums()
    paired = [(gw.clone(), gb.clone()) for _, _, _, (gw, gb) in la + lc]
    for _, _, _, (gw, gb) in la + lc:
        gw.zero_(); gb.zero_()
    pk.mlp_backward(xa, ya, gya, la)
    pk.mlp_backward(xc, yc, gyc, lc)
    pk.flush_partial_sums()
    for (gw0, gb0), (_, _, _, (gw, gb)) in zip(paired, la + lc):
        assert torch.equal(gw0, gw) and torch.equal(gb0, gb)


def _walk_reference(x, layers, gy):
    """float64 torch: the layer outputs, the dZ of every layer and the parameter gradients of one Linear / ELU stack"""
    ws = [w.double().requires_grad_(True) for w, _, _, _ in layers]
    bs = [b.double().requires_grad_(True) for _, b, _, _ in layers]
    h, ys, zs = x.double(), [], []
    for (w, b, (_, _, act, _)) in zip(ws, bs, layers):
        z = torch.nn.functional.linear(h, w, b)
        z.retain_grad()
        zs.append(z)
        h = torch.nn.functional.elu(z) if act else z
        ys.append(h)
    h.backward(gy.double())
    return ys, [z.grad for z in zs], [w.grad for w in ws], [b.grad for b in bs]


@pytest.mark.gpu
@pytest.mark.parametrize("M,dims_a,dims_c", [(8192, [41, 400, 200, 100, 9], [113, 400, 200, 100, 1]),
                                             (1000, [41, 400, 200, 100, 18], [113, 400, 200, 100, 1]),
                                             (77, [7, 33, 64, 5], [20, 416, 17, 3]),
                                             (64, [16, 16], [3, 1]),
                                             (130, [41, 400, 200, 100, 9], None)])
def test_network_walk_matches_torch(hip, M, dims_a, dims_c):
    """csrc/ppo_mlp_walk.hip - all layers of one or two Linear / ELU stacks in ONE launch per direction (64-row blocks, activations in LDS,
    v_mfma_f32_16x16x4_f32) - against float64 torch: every layer output, every dZ of the input-gradient chain, and the parameter gradients the
    trainer forms from them; the trainer's shapes, ragged row counts, widths that are no multiple of 16, one network alone, one-layer networks."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(M + sum(dims_a))
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731

    def net(dims):
        n = len(dims) - 1
        return [(r(dims[i + 1], dims[i]) * dims[i] ** -0.5, r(dims[i + 1]) * 0.3, 1 if i < n - 1 else 0,
                 (torch.zeros(dims[i + 1], dims[i], device=dev), torch.zeros(dims[i + 1], device=dev))) for i in range(n)]
    nets = [(r(M, d[0]), net(d), r(M, d[-1])) for d in (dims_a, dims_c) if d is not None]
    outs = pk.mlp_walk_forward([(x, layers) for x, layers, _ in nets])
    assert outs is not None
    refs = [_walk_reference(x, layers, gy) for x, layers, gy in nets]
    close = lambda got, want, what: torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5, msg=lambda m: f"{what}: {m}")   # noqa: E731
    for (x, layers, gy), ys, (ys64, dz64, gw64, gb64) in zip(nets, outs, refs):
        for l, (y, y64) in enumerate(zip(ys, ys64)):
            close(y, y64, f"output of layer {l}")
    last = pk.mlp_walk_forward([(x, layers) for x, layers, _ in nets], store_hidden=False)                 # the rollout's form: network outputs only
    for ys, ys_last in zip(outs, last):
        assert all(y is None for y in ys_last[:-1]) and torch.equal(ys_last[-1], ys[-1])
    dzs = pk.mlp_walk_backward([(gy, ys, layers) for (x, layers, gy), ys in zip(nets, outs)])
    assert dzs is not None
    for (x, layers, gy), ys, dz, (ys64, dz64, gw64, gb64) in zip(nets, outs, dzs, refs):
        for l, (d, d64) in enumerate(zip(dz, dz64)):
            close(d, d64, f"dZ of layer {l}")
    if len(nets) == 2:                                                                        # the trainer's backward on top of it: parameter gradients
        (xa, la, gya), (xc, lc, gyc) = nets
        pk.mlp_backward_pair(xa, outs[0], gya, la, xc, outs[1], gyc, lc)
        pk.flush_partial_sums()
        for (x, layers, gy), (ys64, dz64, gw64, gb64) in zip(nets, refs):
            for l, (_, _, _, (gw, gb)) in enumerate(layers):
                scale = float(gw64[l].abs().max()) + 1e-12
                assert float((gw.double() - gw64[l]).abs().max()) <= 3e-5 * scale + 1e-6, f"dW of layer {l}"
                assert float((gb.double() - gb64[l]).abs().max()) <= 3e-5 * (float(gb64[l].abs().max()) + 1e-12) + 1e-6, f"db of layer {l}"


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [8192, 1000, 77])
def test_direct_weight_gradients_match_torch(hip, rows):
    """csrc/ppo_dw_direct.hip - [dZ^T X | column sums of dZ] for the eight layers of a minibatch step in ONE launch, 64 x 64 blocks with both operands
    straight from memory as interleaved MFMA fragments - against float64 torch: the trainer's shapes (widths that are no multiple of 4 or 64, a
    one-column dZ), row counts that are no multiple of the 1024-row chunk or of 4; and the same twice: identical bits (fixed summation order)."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(rows)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    shapes = [(400, 41), (200, 400), (100, 200), (9, 100), (400, 113), (200, 400), (100, 200), (1, 100)]
    as_ = [r(rows, n1) for n1, _ in shapes]
    bs = [r(rows, n2) for _, n2 in shapes]
    outs = [(torch.zeros(n1, n2, device=dev), torch.zeros(n1, device=dev)) for n1, n2 in shapes]
    assert pk.gemm_tn_bias_direct(as_, bs, outs)
    pk.flush_partial_sums()
    first = [(gw.clone(), gb.clone()) for gw, gb in outs]
    for (gw, gb), a, b in zip(outs, as_, bs):
        want_w, want_b = a.double().t() @ b.double(), a.double().sum(0)
        assert float((gw.double() - want_w).abs().max()) <= 2e-5 * float(want_w.abs().max()) + 1e-5
        assert float((gb.double() - want_b).abs().max()) <= 2e-5 * float(want_b.abs().max()) + 1e-5
    for gw, gb in outs:
        gw.zero_(); gb.zero_()
    assert pk.gemm_tn_bias_direct(as_, bs, outs)
    pk.flush_partial_sums()
    for (gw0, gb0), (gw, gb) in zip(first, outs):
        assert torch.equal(gw0, gw) and torch.equal(gb0, gb)


@pytest.mark.gpu
def test_network_walk_declines_what_does_not_fit(hip):
    """a layer wider than 416 (26 column tiles) or activations beyond the LDS budget: None, and the pair entry points fall back to the per-layer launches"""
    dev = "cuda:0"
    x = torch.randn(256, 64, device=dev)
    wide = [(torch.randn(512, 64, device=dev) * 0.1, torch.zeros(512, device=dev), 1, None), (torch.randn(8, 512, device=dev) * 0.05, torch.zeros(8, device=dev), 0, None)]
    assert pk.mlp_walk_forward([(x, wide)]) is None
    ya, yc = pk.mlp_forward_pair(x, wide, x, wide)
    want = torch.nn.functional.linear(torch.nn.functional.elu(torch.nn.functional.linear(x, wide[0][0], wide[0][1])), wide[1][0], wide[1][1])
    torch.testing.assert_close(ya[-1], want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(yc[-1], want, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_activation_gradient_applied_where_the_product_is_stored(hip):
    """tfp_gemm_nn_dz (an input-gradient product that leaves as the dZ of the layer below) against the product followed by the elementwise factor -
    one fp32 multiply per element either way: identical bits, vector and scalar epilogue, ragged sizes; and the backward walk built on it
    (mlp_backward: every product below the top layer stages plain operands) against the walk that multiplies in the operand loads of both
    consumers, which is what the kernels did before: identical weight and bias gradients."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(29)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    elu_grad = lambda y: torch.where(y > 0, torch.ones_like(y), y + 1.0)                      # noqa: E731
    for (M, K, N, with_y) in ((8192, 200, 400, True), (8192, 9, 100, False), (777, 100, 200, True), (130, 36, 41, False), (64, 8, 3, True)):
        a, b = r(M, K), r(K, N)
        y = (torch.rand(M, K, device=dev, generator=g) - 0.3) if with_y else None
        yo = torch.rand(M, N, device=dev, generator=g) - 0.3
        assert torch.equal(pk.gemm_nn(a, b, y, y_out=yo), pk.gemm_nn(a, b, y) * elu_grad(yo))
    got = pk.gemm_nn_group([r(8192, 200), r(8192, 200)], [r(200, 400), r(200, 400)], None, [None, None])
    assert got is not None and len(got) == 2

    def net(k_in, n_out):
        dims = [k_in, 400, 200, 100, n_out]
        return [(r(dims[i + 1], dims[i]) * dims[i] ** -0.5, r(dims[i + 1]), 1 if i < 3 else 0,
                 (torch.zeros(dims[i + 1], dims[i], device=dev), torch.zeros(dims[i + 1], device=dev))) for i in range(4)]
    layers = net(41, 9)
    x, gy = r(8192, 41), r(8192, 9)
    ys = pk.mlp_forward(x, layers)
    pk.mlp_backward(x, ys, gy, layers)
    pk.flush_partial_sums()
    new = [(gw.clone(), gb.clone()) for _, _, _, (gw, gb) in layers]
    for k in range(len(layers) - 1, -1, -1):                     # the former walk: gy stays the gradient of the layer OUTPUT all the way down
        w, _, act, _ = layers[k]
        yy = ys[k] if act else None
        gw, gb = pk.gemm_tn_bias(gy, ys[k - 1] if k > 0 else x, yy)
        assert torch.equal(gw, new[k][0]) and torch.equal(gb, new[k][1]), k
        if k > 0:
            gy = pk.gemm_nn(gy, w, yy)


@pytest.mark.gpu
def test_rollout_bookkeeping_kernels_hold_the_bits_of_the_torch_expressions(hip):
    """tfp_rollout_record / tfp_rollout_reward / tfp_gae (the rollout's ~25 elementwise launches per environment step and 8 per step of the advantage
    loop as three kernels) against the PyTorch expressions of PPOTrainer.rollout: every product and sum is rounded separately in the same order, so
    actions, filed rows, rewards, done flags, advantages and returns are identical; the negative log-likelihood (a sum over the actions, in another
    order) to rounding."""
    from leibnizgym_amd.ppo import neglogp
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(31)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)                                   # noqa: E731
    for n, A, Do, Ds in ((8192, 9, 41, 113), (1000, 18, 50, 0), (3, 9, 41, 113)):
        T = 4
        obs, states = r(n, Do), (r(n, Ds) if Ds else None)
        mu, eps, val, ls = r(n, A), r(n, A), r(n), 0.3 * r(A)
        buf = dict(obs=torch.zeros(T, n, Do, device=dev), states=torch.zeros(T, n, Ds, device=dev) if Ds else None, act=torch.zeros(T, n, A, device=dev),
                   mu=torch.zeros(T, n, A, device=dev), nlp=torch.zeros(T, n, device=dev), val=torch.zeros(T + 1, n, device=dev))
        a = pk.rollout_record(obs, states, mu, ls, ls.exp(), eps, val, buf, 2)
        want = mu + ls.expand_as(mu).exp() * eps
        assert torch.equal(a, want) and a.data_ptr() == buf["act"][2].data_ptr()
        assert torch.equal(buf["obs"][2], obs) and torch.equal(buf["mu"][2], mu) and torch.equal(buf["val"][2], val)
        assert Ds == 0 or torch.equal(buf["states"][2], states)
        torch.testing.assert_close(buf["nlp"][2], neglogp(want, mu, ls.expand_as(mu)), rtol=2e-6, atol=2e-6)
        assert float(buf["obs"][1].abs().max()) == 0.0 and float(buf["act"][3].abs().max()) == 0.0          # the other slots stay untouched
    n, T = 8192, 32
    rew_in, d = r(n), torch.rand(n, device=dev, generator=g) < 0.1
    rew, done = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pk.rollout_reward(rew_in, d, 0.01, rew, done)
    assert torch.equal(rew, rew_in * 0.01) and torch.equal(done, d.float())
    pk.rollout_reward(rew_in, d.to(torch.uint8), 0.01, rew, done)
    assert torch.equal(done, d.float())
    rews, dones, vals = r(T, n), (torch.rand(T, n, device=dev, generator=g) < 0.05).float(), r(T + 1, n)
    gamma, tau = 0.99, 0.95
    adv, last = torch.zeros(T, n, device=dev), torch.zeros(n, device=dev)
    for t in reversed(range(T)):                                       # the loop of PPOTrainer.rollout
        nd = 1.0 - dones[t]
        delta = rews[t] + gamma * vals[t + 1] * nd - vals[t]
        last = delta + gamma * tau * nd * last
        adv[t] = last
    got_adv, got_ret = pk.gae(rews, dones, vals, gamma, tau)
    assert torch.equal(got_adv, adv) and torch.equal(got_ret, adv + vals[:T])
