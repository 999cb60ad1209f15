"""Minimal asymmetric-actor-critic PPO for the TriFinger env (BASELINE config 5 when `rl_games` is absent).

Follows the agent configuration the reference ships for RL-Games (resources/config/rlg/asymm.yaml): continuous
A2C/PPO, actor MLP [400, 200, 100] ELU on `obs`, central value MLP [400, 200, 100] ELU on `states`, state-independent
log-std (`fixed_sigma`), horizon `steps_num` 32, 4 mini-epochs, minibatch = num_envs, gamma 0.99, GAE tau 0.95,
actor lr 3e-4 adaptive on a KL threshold of 0.008, e_clip 0.2, reward scale 0.01, grad-norm 1.0, bounds loss 1e-4,
normalised advantages; the central value network has its OWN optimiser state (asymm.yaml:70-90): lr 5e-4 constant, its
own gradient-norm truncation, unweighted MSE loss (RL-Games drops `critic_coef` from the actor loss when a central value
network exists); initialisers as asymm.yaml:16-18,31-33,84-86 (`variance_scaling_initializer`: truncated normal with
variance scale / fan_in; biases zero).  `PPOConfig.from_rlg` reads all of it from the `rlg` tree of the launcher.
Checkpoints (`save` / `restore`: networks, optimiser moments, learning rates, frame and epoch counters), periodic and
best-so-far saving and a deterministic `play` mode mirror what the reference gets from RL-Games (`args.checkpoint`,
`args.play`, `save_frequency`, `save_best_after`).

This is host-side training glue, NOT part of the measured hot path.  On a GPU the minibatch step runs on the hand-written kernels of
csrc/ppo_kernels.hip (leibnizgym_amd/ppo_kernels.py): one gather launch, the Linear / ELU layers on fp32 MFMA, the objective with all
its gradients in one launch, the chunk sums of the weight gradients in one launch, truncation + Adam over one flat buffer - about 30
launches per step, launched eagerly (a HIP-graph replay of the step existed in round 2; it was slower than eager launches and its
learning curves were never explained, so it was removed in round 3 - DESIGN.md section 8).
On the CPU (tests) and with `fused_kernels=False` the same step is plain torch.
Data parallelism: every rank owns an env shard (leibnizgym_amd.sharding) and its own rollout; gradients are averaged
with ONE all-reduce of a flat buffer per minibatch (`torch.distributed`, backend nccl = RCCL over xGMI on the GPU
box, gloo in the CPU tests) - ~1 MB, latency-bound, so a single fused collective is the right shape.
"""
import math
import os
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn as nn


@dataclass
class PPOConfig:
    units: List[int] = field(default_factory=lambda: [400, 200, 100])
    horizon: int = 32                 # steps_num
    mini_epochs: int = 4
    minibatches: int = 32             # minibatch_size = num_envs  ->  horizon minibatches per epoch
    gamma: float = 0.99
    tau: float = 0.95
    lr: float = 3e-4
    lr_value: float = 5e-4            # central_value_config.lr
    kl_threshold: float = 0.008       # lr_schedule: adaptive
    e_clip: float = 0.2
    critic_coef: float = 4.0
    reward_scale: float = 0.01
    grad_norm: float = 1.0
    bounds_loss_coef: float = 1e-4
    entropy_coef: float = 0.0
    normalize_advantage: bool = True
    value_mini_epochs: int = 0        # central_value_config.mini_epochs; 0 = as mini_epochs (they must be equal: one fused pass serves both)
    value_grad_norm: float = 1.0      # central_value_config.grad_norm
    mu_init_scale: float = 0.02       # network.space.continuous.mu_init (variance scaling)
    actor_init: str = "default"       # network.mlp.initializer.name
    value_init: str = "variance_scaling_initializer"   # central_value_config.network.mlp.initializer
    value_init_scale: float = 2.0
    save_frequency: int = 100
    save_best_after: int = 500
    max_epochs: int = 100000
    name: str = "trifinger"
    seed: int = 7
    fused_kernels: bool = True        # hand-written HIP kernel for the objective, forward and backward in one launch (GPU only)

    @classmethod
    def from_rlg(cls, rlg: dict, num_envs: int = None, **overrides):
        """Hyper-parameters from the launcher's `rlg` tree (leibnizgym_amd/config.py:RLG_ASYMM, i.e. the reference's
        resources/config/rlg/asymm.yaml).  `minibatch_size` is given in samples there; the trainer counts minibatches per
        epoch: horizon * num_envs / minibatch_size."""
        p = rlg["params"]
        c, net = p["config"], p["network"]
        cv = c.get("central_value_config", {})
        kw = dict(units=list(net["mlp"]["units"]), horizon=int(c["steps_num"]), mini_epochs=int(c["mini_epochs"]),
                  gamma=float(c["gamma"]), tau=float(c["tau"]), lr=float(c["learning_rate"]),
                  kl_threshold=float(c["lr_threshold"]), e_clip=float(c["e_clip"]), critic_coef=float(c["critic_coef"]),
                  reward_scale=float(c["reward_shaper"]["scale_value"]), grad_norm=float(c["grad_norm"]),
                  bounds_loss_coef=float(c["bounds_loss_coef"]), entropy_coef=float(c["entropy_coef"]),
                  normalize_advantage=bool(c["normalize_advantage"]),
                  mu_init_scale=float(net["space"]["continuous"]["mu_init"].get("scale", 0.02)),
                  actor_init=str(net["mlp"]["initializer"]["name"]),
                  save_frequency=int(c.get("save_frequency", 100)), save_best_after=int(c.get("save_best_after", 500)),
                  max_epochs=int(c.get("max_epochs", 100000)), name=str(c.get("name", "trifinger")),
                  seed=int(rlg.get("seed", 7)))
        if cv:
            init = cv["network"]["mlp"]["initializer"]
            kw.update(lr_value=float(cv["lr"]), value_mini_epochs=int(cv["mini_epochs"]),
                      value_grad_norm=float(cv["grad_norm"]), value_init=str(init["name"]),
                      value_init_scale=float(init.get("scale", 2.0)))
        if num_envs:
            kw["minibatches"] = max(1, kw["horizon"] * int(num_envs) // int(c["minibatch_size"]))
        kw.update(overrides)
        return cls(**kw)


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T + b with a weight gradient that is parallel over the batch: dW = dY^T X has a tiny output (e.g. 400 x 41)
    and the whole minibatch (8192) as its reduction dimension, which rocBLAS runs as ~80 workgroups of a 256-CU chip
    (55 us per call, the largest single item of the update).  Splitting the batch into S slices turns it into one batched
    GEMM with S times the workgroups plus a sum over S."""
    SLICES = 16

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gy @ w if ctx.needs_input_grad[0] else None
        n = x.shape[0]
        s = _SplitKLinear.SLICES
        if n % s == 0 and n >= 64 * s:
            gw = torch.bmm(gy.view(s, n // s, -1).transpose(1, 2), x.view(s, n // s, -1)).sum(0)
        else:
            gw = gy.t() @ x
        return gx, gw, gy.sum(0)


class SplitKLinear(nn.Linear):
    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and torch.is_grad_enabled():
            return _SplitKLinear.apply(x, self.weight, self.bias)
        return super().forward(x)


class FusedMLP(nn.Sequential):
    """Linear / ELU stack (same modules and state-dict keys as the nn.Sequential it is).  With `mfma` set (the trainer does it on a
    GPU when `fused_kernels` is on) every layer runs on the hand-written fp32 MFMA kernels of csrc/ppo_kernels.hip: bias and ELU
    fused into the forward product, the ELU derivative formed in the operand loads of the two backward products, the bias gradient
    as an extra column of the weight-gradient product."""
    mfma = False

    def layer_list(self):
        """[(weight, bias, act, grad_out)] of the Linear layers, for ppo_kernels.mlp_forward / mlp_backward"""
        mods, out = list(self), []
        for i, m in enumerate(mods):
            if isinstance(m, nn.Linear):
                act = 1 if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.ELU)) else 0
                out.append((m.weight, m.bias, act, getattr(m, "_grad_out", None)))
        return out

    def forward(self, x):
        if not (self.mfma and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
            return super().forward(x)
        from .ppo_kernels import mfma_linear
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                act = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ELU)
                x = mfma_linear(x, m.weight, m.bias, 1 if act else 0, getattr(m, "_grad_out", None) if torch.is_grad_enabled() else None)
                i += 2 if act else 1
            else:
                x = m(x)
                i += 1
        return x


def mlp(inp, units, out):
    layers, last = [], inp
    for u in units:
        layers += [SplitKLinear(last, u), nn.ELU()]
        last = u
    layers.append(SplitKLinear(last, out))
    return FusedMLP(*layers)


def variance_scaling_(w: torch.Tensor, scale: float) -> torch.Tensor:
    """RL-Games' `variance_scaling_initializer` (fan_in mode): normal with variance scale / fan_in, truncated at two
    standard deviations."""
    std = math.sqrt(scale / w.shape[1])
    return nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2.0 * std, b=2.0 * std)


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, state_dim, act_dim, units, cfg: "PPOConfig" = None):
        super().__init__()
        self.actor = mlp(obs_dim, units, act_dim)
        self.critic = mlp(state_dim if state_dim > 0 else obs_dim, units, 1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))          # sigma_init const 0, fixed_sigma
        self.central = state_dim > 0
        if cfg is not None:
            self.init_like_rl_games(cfg)

    def init_like_rl_games(self, cfg: "PPOConfig"):
        """asymm.yaml:12-18,31-33,84-86 through RL-Games' network builder: every Linear gets its MLP initialiser
        (`default` leaves torch's) and a zero bias; the mu head gets `mu_init`."""
        for net, kind, scale in ((self.actor, cfg.actor_init, 2.0), (self.critic, cfg.value_init if self.central else cfg.actor_init, cfg.value_init_scale)):
            for m in net:
                if isinstance(m, nn.Linear):
                    if kind == "variance_scaling_initializer":
                        variance_scaling_(m.weight.data, scale)
                    nn.init.zeros_(m.bias)
        variance_scaling_(self.actor[-1].weight.data, cfg.mu_init_scale)

    def actor_parameters(self):
        return list(self.actor.parameters()) + [self.log_std]

    def critic_parameters(self):
        return list(self.critic.parameters())

    def value(self, obs, states):
        return self.critic(states if self.central else obs).squeeze(-1)

    def dist(self, obs):
        mu = self.actor(obs)
        return mu, self.log_std.expand_as(mu)

    def dist_and_value(self, obs, states):
        """(mu, log_std, value) of one batch: on the hand-written kernels and without autograd (the rollout) layer k of the two networks is ONE launch
        (ppo_kernels.mlp_forward_pair: four launches instead of eight per environment step)"""
        xc = states if self.central else obs
        if self.actor.mfma and self.critic.mfma and obs.is_cuda and obs.dtype == torch.float32 and not torch.is_grad_enabled():
            from . import ppo_kernels as pk
            ya, yc = pk.mlp_forward_pair(obs.contiguous(), self.actor.layer_list(), xc.contiguous(), self.critic.layer_list(), store_hidden=False)
            mu, v = ya[-1], yc[-1].squeeze(-1)
        else:
            mu, v = self.actor(obs), self.critic(xc).squeeze(-1)
        return mu, self.log_std.expand_as(mu), v


def neglogp(x, mu, log_std):
    return (0.5 * ((x - mu) / log_std.exp()).pow(2) + log_std + 0.5 * math.log(2 * math.pi)).sum(-1)


class PPOTrainer:
    """`env` is an RlGamesGpuEnvAdapter-like object: reset() -> {"obs","states"} or obs; step(a) -> (same, r, d, info).  Contract on the tensors it
    returns: with the attribute `buffers_stable_until_next_step = True` they may be the env's live buffers (valid until its next step / reset, on the
    calling stream) and the fused rollout reads them in place; without it every step's observation is cloned before it is used."""

    def __init__(self, env, obs_dim, state_dim, act_dim, cfg: PPOConfig = None, device="cuda:0", group=None):
        self.env, self.cfg, self.device, self.group = env, cfg or PPOConfig(), torch.device(device), group
        c = self.cfg
        if state_dim > 0 and c.value_mini_epochs not in (0, c.mini_epochs):
            raise ValueError("central_value_config.mini_epochs must equal mini_epochs: actor and central value network "
                             "are updated in one fused pass over the same minibatches")
        torch.manual_seed(c.seed)              # identical initial weights on every rank (and a broadcast below)
        self.net = ActorCritic(obs_dim, state_dim, act_dim, c.units, c).to(self.device)
        # fused multi-tensor Adam on the GPU: the update is launch-bound (tiny MLPs), one kernel instead of ~60.
        # Two parameter groups = RL-Games' two optimisers: the actor's learning rate follows the KL schedule, the
        # central value network keeps central_value_config.lr; moments are per parameter, so nothing else is shared.
        fused = self.device.type == "cuda"
        lr_v = c.lr_value if self.net.central else c.lr
        groups = [{"params": self.net.actor_parameters(), "lr": c.lr},
                  {"params": self.net.critic_parameters(), "lr": lr_v}]
        kw = {"fused": True} if fused else {}
        self.opt = torch.optim.Adam(groups, eps=1e-8, **kw)
        self.lr = c.lr
        self.fused_loss = fused and c.fused_kernels        # hand-written objective kernel (GPU only)
        self.net.actor.mfma = self.net.critic.mfma = bool(fused and c.fused_kernels)   # ... and the MFMA linear layers
        self.dist_on = False
        self.n_grad_allreduce = self.n_kl_allreduce = 0       # collectives issued so far (what a test of the distributed path counts)
        rank = 0
        try:
            import torch.distributed as dist
            self.dist = dist
            # a process group that exists is used - also a world of one (`torch.distributed.run --nproc-per-node 1`): the collectives then run exactly
            # as in a multi-GPU job (that is how the RCCL path is exercised on a one-GPU box); without a group nothing is exchanged
            self.dist_on = dist.is_available() and dist.is_initialized()
        except Exception:
            self.dist = None
        if self.dist_on:                       # identical initial weights on every rank
            for p in self.net.parameters():
                self.dist.broadcast(p.data, src=0, group=group)
            rank = self.dist.get_rank(group)
        # GPU + fused kernels: truncation + Adam of both groups as two hand-written launches over one flat buffer
        self.flat_opt = None
        if fused and c.fused_kernels:
            from .ppo_kernels import FlatClipAdam
            if self.net.central:
                self.flat_opt = FlatClipAdam(self.net.actor_parameters(), self.net.critic_parameters(), c.lr, lr_v, c.grad_norm, c.value_grad_norm)
            else:
                self.flat_opt = FlatClipAdam(list(self.net.parameters()), [], c.lr, c.lr, c.grad_norm, c.grad_norm)
            for net in (self.net.actor, self.net.critic):      # the MFMA layers write dW / db straight into the flat gradient buffer
                for m in net:
                    if isinstance(m, nn.Linear):
                        m._grad_out = (self.flat_opt.grad_view(m.weight), self.flat_opt.grad_view(m.bias))
        # exploration noise and minibatch order must differ between ranks (the same env index of two shards would
        # otherwise receive the same noise sequence): re-seed with the rank after the weights are in place
        torch.manual_seed(c.seed + 7919 * rank)
        self.last = self._unpack(env.reset())
        self.frames = 0
        self.epoch = 0
        self.last_info = {}
        self.best_reward = -float("inf")

    # ---- checkpoints (what RL-Games' save / restore / `args.checkpoint` give the reference launcher) ----
    def _optimizer_state(self):
        """Adam state in ONE format whichever optimiser runs (the flat hand-written one on a GPU, torch's otherwise): first and
        second moments per parameter NAME, the step counter, the two learning rates - so that a checkpoint written by one path
        restores into the other (CPU <-> GPU, fused_kernels on <-> off)."""
        names = {id(p): n for n, p in self.net.named_parameters()}
        m, v, step = {}, {}, 0.0
        if self.flat_opt is not None:
            fo = self.flat_opt
            for p, o in zip(fo.params, fo.offsets):
                m[names[id(p)]] = fo.m[o:o + p.numel()].view_as(p).clone()
                v[names[id(p)]] = fo.v[o:o + p.numel()].view_as(p).clone()
            step = float(fo.step_count[1].item())             # completed steps
            lrs = [float(x) for x in fo.lr.tolist()]
        else:
            for g in self.opt.param_groups:
                for p in g["params"]:
                    st = self.opt.state.get(p, {})
                    m[names[id(p)]] = st["exp_avg"].detach().clone() if "exp_avg" in st else torch.zeros_like(p)
                    v[names[id(p)]] = st["exp_avg_sq"].detach().clone() if "exp_avg_sq" in st else torch.zeros_like(p)
                    step = max(step, float(st["step"]) if "step" in st else 0.0)
            lrs = [float(g["lr"]) for g in self.opt.param_groups]
        return {"kind": "adam_per_parameter", "exp_avg": m, "exp_avg_sq": v, "step": step, "lrs": lrs}

    def _parse_optimizer_state(self, sd):
        """Any of the three optimiser-state formats -> (first moments by parameter name, second moments by name, step, learning rates),
        VALIDATED against this network (unknown names, wrong shapes, wrong total size raise ValueError) and without touching the
        trainer: restore() calls this before it overwrites a single weight.  Formats: 'adam_per_parameter' (what `_optimizer_state`
        writes), 'flat_clip_adam' (round 2, GPU path: the flat buffers of FlatClipAdam, every parameter padded to 4 floats, actor
        group then critic group for a central-value net, net.parameters() order otherwise) and torch.optim.Adam.state_dict() (round 2,
        CPU path: moments by parameter index in the order of the optimiser's groups)."""
        named = dict(self.net.named_parameters())
        names = {id(p): n for n, p in named.items()}
        kind = sd.get("kind")
        if kind == "adam_per_parameter":
            m, v, step, lrs = dict(sd["exp_avg"]), dict(sd["exp_avg_sq"]), float(sd["step"]), list(sd.get("lrs", []))
        elif kind == "flat_clip_adam":
            order = self.net.actor_parameters() + self.net.critic_parameters() if self.net.central else list(self.net.parameters())
            if sum((p.numel() + 3) & ~3 for p in order) != sd["m"].numel() or sd["v"].numel() != sd["m"].numel():
                raise ValueError("checkpoint optimizer state belongs to a network of another size")
            m, v, off = {}, {}, 0
            for p in order:
                m[names[id(p)]] = sd["m"][off:off + p.numel()].view_as(p)
                v[names[id(p)]] = sd["v"][off:off + p.numel()].view_as(p)
                off += (p.numel() + 3) & ~3
            step, lrs = float(sd["step"].item()), [float(x) for x in sd["lr"].tolist()]
        elif "state" in sd and "param_groups" in sd:
            order = [p for g in self.opt.param_groups for p in g["params"]]     # actor group (log_std last), then critic group
            n_ck = sum(len(g["params"]) for g in sd["param_groups"])
            if n_ck != len(order):
                raise ValueError(f"checkpoint optimizer state holds {n_ck} parameters, this network {len(order)}")
            st = sd["state"]
            m = {names[id(p)]: st[i]["exp_avg"] for i, p in enumerate(order) if i in st}
            v = {names[id(p)]: st[i]["exp_avg_sq"] for i, p in enumerate(order) if i in st}
            step = max([float(x["step"]) for x in st.values()] or [0.0])
            lrs = [float(g["lr"]) for g in sd["param_groups"]]
        else:
            raise ValueError(f"unknown optimizer state format (kind = {kind!r})")
        for which in (m, v):
            for n, t in which.items():
                if n not in named:
                    raise ValueError(f"checkpoint optimizer state names a parameter this network does not have: {n}")
                if tuple(t.shape) != tuple(named[n].shape):
                    raise ValueError(f"checkpoint optimizer moment of '{n}': shape {tuple(t.shape)}, parameter {tuple(named[n].shape)}")
        return m, v, step, lrs

    def _load_optimizer_state(self, sd, lr, parsed=None):
        """inverse of `_optimizer_state` (and of the two round-2 formats: `_parse_optimizer_state`)"""
        named = dict(self.net.named_parameters())
        m, v, step, lrs = parsed if parsed is not None else self._parse_optimizer_state(sd)
        lr_actor = float(lr) if lr is not None else (lrs[0] if lrs else self.lr)
        lr_value = lrs[1] if len(lrs) > 1 else None
        if self.flat_opt is not None:
            fo = self.flat_opt
            names = {id(p): n for n, p in named.items()}
            fo.m.zero_(); fo.v.zero_()
            for p, o in zip(fo.params, fo.offsets):                  # every moment into its own (padded) slot
                n = names[id(p)]
                if n in m:
                    fo.m[o:o + p.numel()].copy_(m[n].reshape(-1).to(fo.m.device))
                    fo.v[o:o + p.numel()].copy_(v[n].reshape(-1).to(fo.v.device))
            fo._set_step(torch.tensor([float(step)]))
            fo.set_lr(0, lr_actor)
            if lr_value is not None:
                fo.set_lr(1, lr_value if self.net.central else lr_actor)
        else:
            fused = self.device.type == "cuda"
            for gi, g in enumerate(self.opt.param_groups):
                for p in g["params"]:
                    n = next(k for k, q in named.items() if q is p)
                    self.opt.state[p] = {
                        "step": torch.tensor(step, dtype=torch.float32, device=p.device if fused else "cpu"),
                        "exp_avg": (m[n].to(p.device).clone().view_as(p) if n in m else torch.zeros_like(p)),
                        "exp_avg_sq": (v[n].to(p.device).clone().view_as(p) if n in v else torch.zeros_like(p))}
                g["lr"] = lr_actor if (gi == 0 or not self.net.central) else (lr_value if lr_value is not None else g["lr"])

    def state_dict(self):
        return {"model": self.net.state_dict(), "optimizer": self._optimizer_state(), "lr": self.lr, "frames": self.frames,
                "epoch": self.epoch, "best_reward": self.best_reward, "config": dict(self.cfg.__dict__)}

    def save(self, path: str):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        torch.save(self.state_dict(), path)
        return path

    def restore(self, path: str):
        ck = torch.load(path, map_location=self.device, weights_only=False)
        parsed = None
        mine = self.net.state_dict()                                 # validate EVERYTHING before anything is overwritten
        for k, t in mine.items():
            if k not in ck["model"] or tuple(ck["model"][k].shape) != tuple(t.shape):
                raise ValueError(f"checkpoint model entry '{k}' missing or of another shape")
        if "optimizer" in ck:
            parsed = self._parse_optimizer_state(ck["optimizer"])
        with torch.no_grad():                                        # in place: the parameters may be views of a flat buffer
            for k, v in self.net.state_dict().items():
                v.copy_(ck["model"][k])
            if "optimizer" in ck:
                self._load_optimizer_state(ck["optimizer"], ck.get("lr"), parsed)
        self.lr = float(ck.get("lr", self.lr))
        self.frames, self.epoch = int(ck.get("frames", 0)), int(ck.get("epoch", 0))
        self.best_reward = float(ck.get("best_reward", -float("inf")))
        return ck

    @torch.no_grad()
    def act(self, obs, deterministic=True):
        mu, ls = self.net.dist(obs)
        return mu if deterministic else mu + ls.exp() * torch.randn_like(mu)

    @torch.no_grad()
    def play(self, steps: int, deterministic=True):
        """`args.play`: roll the policy without learning; returns the mean reward per step and the last info dict."""
        obs, _ = self.last
        total, info = 0.0, {}
        for _ in range(steps):
            out, r, _, extra = self.env.step(self.act(obs, deterministic))
            obs, states = self._unpack(out)
            total += float(r.mean())
            if isinstance(extra, (list, tuple)) and len(extra) > 1 and isinstance(extra[1], dict):
                info = extra[1]
        self.last = (obs, states)
        return total / max(steps, 1), info

    @staticmethod
    def _unpack(o):
        if isinstance(o, dict):
            return o["obs"].clone(), o["states"].clone()
        return o.clone(), None

    @torch.no_grad()
    def rollout(self):
        c, T = self.cfg, self.cfg.horizon
        obs, states = self.last
        n = obs.shape[0]
        dev = self.device
        # with the hand-written kernels the bookkeeping of a step is two launches (ppo_kernels.rollout_record / rollout_reward) and the advantage
        # estimate one (ppo_kernels.gae) instead of ~25 + 8 T elementwise launches; same arithmetic, same random numbers (torch.randn_like)
        fused = self.fused_loss and obs.is_cuda and obs.dtype == torch.float32
        if fused:
            from . import ppo_kernels as pk
        A = self.net.log_std.numel()
        sigma = self.net.log_std.detach().exp()              # constant over the rollout
        buf = dict(obs=torch.zeros(T, n, obs.shape[1], device=dev),
                   states=torch.zeros(T, n, states.shape[1], device=dev) if states is not None else None,
                   act=torch.zeros(T, n, A, device=dev), nlp=torch.zeros(T, n, device=dev), val=torch.zeros(T + 1, n, device=dev),
                   rew=torch.zeros(T, n, device=dev), done=torch.zeros(T, n, device=dev), mu=torch.zeros(T, n, A, device=dev))
        for t in range(T):
            mu, ls, val_t = self.net.dist_and_value(obs, states)
            if fused:
                a = pk.rollout_record(obs, states, mu, self.net.log_std, sigma, torch.randn_like(mu), val_t, buf, t)
            else:
                a = mu + ls.exp() * torch.randn_like(mu)
                buf["obs"][t], buf["act"][t], buf["mu"][t] = obs, a, mu
                if states is not None:
                    buf["states"][t] = states
                buf["nlp"][t] = neglogp(a, mu, ls)
                buf["val"][t] = val_t
            out, r, d, extra = self.env.step(a)
            last_step = t == T - 1
            # an env that DECLARES its buffers stable until its next step (`buffers_stable_until_next_step`: RlGamesGpuEnvAdapter hands out the
            # engine's own tensors) is read in place - they are filed above before the next step overwrites them, only what outlives the loop is cloned;
            # any other env gets a snapshot per step (it may refresh its observation asynchronously or on another stream)
            live = fused and not last_step and getattr(self.env, "buffers_stable_until_next_step", False)
            obs, states = ((out["obs"], out["states"]) if isinstance(out, dict) else (out, None)) if live else self._unpack(out)
            if isinstance(extra, (list, tuple)) and len(extra) > 1 and isinstance(extra[1], dict):
                self.last_info = extra[1]                   # RL-Games convention: [[], info] (direct logging from the env)
            if fused and r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and d.is_cuda and d.dtype in (torch.bool, torch.uint8) and d.is_contiguous():
                pk.rollout_reward(r, d, c.reward_scale, buf["rew"][t], buf["done"][t])
            else:
                buf["rew"][t] = r.to(dev) * c.reward_scale
                buf["done"][t] = d.to(dev).float()
        buf["val"][T] = self.net.value(obs, states)
        self.last = (obs, states)
        if fused:
            buf["adv"], buf["ret"] = pk.gae(buf["rew"], buf["done"], buf["val"], c.gamma, c.tau)
        else:
            adv = torch.zeros(T, n, device=dev)
            last = torch.zeros(n, device=dev)
            for t in reversed(range(T)):
                nd = 1.0 - buf["done"][t]
                delta = buf["rew"][t] + c.gamma * buf["val"][t + 1] * nd - buf["val"][t]
                last = delta + c.gamma * c.tau * nd * last
                adv[t] = last
            buf["ret"] = adv + buf["val"][:T]
            buf["adv"] = adv
        self.frames += T * n
        return buf

    # ---- one minibatch step, split at the gradient exchange ----
    def _mb_backward(self, d, idx, acc):
        """gather the minibatch, forward, losses, backward; returns nothing - gradients sit in p.grad, the running
        sums of the logged quantities in `acc` (device tensors, no host sync)"""
        c = self.cfg
        if self.fused_loss:
            return self._mb_backward_fused(d, idx, acc, None)
        obs = d["obs"][idx]
        mu, ls = self.net.dist(obs)
        nlp = neglogp(d["act"][idx], mu, ls)
        ratio = (d["old_nlp"][idx] - nlp).exp()
        a = d["adv"][idx]
        a_loss = torch.max(-a * ratio, -a * ratio.clamp(1 - c.e_clip, 1 + c.e_clip)).mean()
        v = self.net.value(obs, d["states"][idx] if d["states"] is not None else None)
        c_loss = (v - d["ret"][idx]).pow(2).mean()
        b_loss = ((mu - 1.1).clamp(min=0).pow(2) + (-1.1 - mu).clamp(min=0).pow(2)).sum(-1).mean()
        ent = (ls + 0.5 + 0.5 * math.log(2 * math.pi)).sum(-1).mean()
        # with a central value network RL-Games trains it on its own unweighted MSE and drops the critic term from the
        # actor loss; the two gradients do not overlap (separate parameters), so one backward serves both
        v_coef = 1.0 if self.net.central else 0.5 * c.critic_coef
        loss = a_loss + v_coef * c_loss - c.entropy_coef * ent + c.bounds_loss_coef * b_loss
        for p in self.net.parameters():
            p.grad = None
        loss.backward()
        with torch.no_grad():      # KL between the old and new diagonal Gaussians (same sigma)
            kl = (0.5 * ((mu - d["old_mu"][idx]) / ls.exp()).pow(2)).sum(-1).mean()
            acc["kl"] += kl
            acc["loss"] += loss.detach(); acc["a_loss"] += a_loss.detach(); acc["c_loss"] += c_loss.detach()

    def _mb_backward_fused(self, d, idx, acc, obs):
        """the same step on the hand-written kernels: one gather launch for the seven minibatch arrays, the MFMA layers, the objective
        and its gradients in ONE launch (ppo_kernels.ppo_loss_and_grads), the backward pass started at the network outputs with those
        gradients (no loss node), the chunk sums of all weight gradients in one launch.  Every parameter gradient lands in its slot of
        the flat gradient buffer directly (the log-std gradient too), so that the optimiser reads nothing but that buffer; the
        statistics accumulate on the device in `acc["_fused"]` = (loss, a_loss, c_loss, kl)"""
        from . import ppo_kernels as pk
        c = self.cfg
        srcs = [d["obs"], d["act"], d["old_nlp"], d["adv"], d["ret"], d["old_mu"]] + ([d["states"]] if d["states"] is not None else [])
        g = pk.gather_rows(srcs, idx)
        obs, act, old_nlp, adv, ret, old_mu = g[:6]
        states = g[6] if d["states"] is not None else None
        for p in self.net.parameters():
            p.grad = None
        # no autograd: the structure is fixed (two Linear / ELU stacks), so the step is a straight sequence of kernel launches from this
        # thread - forward of both networks, the objective with its gradients, the two backward walks, one launch for all chunk sums
        with torch.no_grad():
            la, lc = self.net.actor.layer_list(), self.net.critic.layer_list()
            xc = states if self.net.central else obs
            ya, yc = pk.mlp_forward_pair(obs, la, xc, lc)        # layer k of both networks in one launch
            mu, v = ya[-1], yc[-1].squeeze(-1)
            v_coef = 1.0 if self.net.central else 0.5 * c.critic_coef
            _, d_mu, d_v, _ = pk.ppo_loss_and_grads(mu, self.net.log_std, v, act, old_nlp, adv, ret, old_mu, acc["_fused"], c.e_clip, v_coef,
                                                    c.entropy_coef, c.bounds_loss_coef, d_ls_out=self.flat_opt.grad_view(self.net.log_std))
            try:
                pk.mlp_backward_pair(obs, ya, d_mu, la, xc, yc, d_v.unsqueeze(-1), lc)
                pk.flush_partial_sums()
            finally:
                pk.discard_partial_sums()                  # nothing stale survives a launch that raised

    @staticmethod
    def _new_acc(dev):
        """running sums of the logged quantities, device side: views of ONE buffer (loss, a_loss, c_loss, kl), which is also
        what the fused objective kernel accumulates into"""
        buf = torch.zeros(4, device=dev)
        return {"loss": buf[0], "a_loss": buf[1], "c_loss": buf[2], "kl": buf[3], "_fused": buf}

    def _mb_apply(self, gathered=False):
        if self.flat_opt is not None:
            self.flat_opt.step(gathered)
            return
        if self.net.central:                   # truncate_grads of each optimiser on its own network
            nn.utils.clip_grad_norm_(self.net.actor_parameters(), self.cfg.grad_norm, foreach=True)
            nn.utils.clip_grad_norm_(self.net.critic_parameters(), self.cfg.value_grad_norm, foreach=True)
        else:
            nn.utils.clip_grad_norm_(self.net.parameters(), self.cfg.grad_norm, foreach=True)
        self.opt.step()

    def _flatten_grads(self, out=None):
        if self.flat_opt is not None:
            return self.flat_opt.gather_grads()
        grads = [p.grad for p in self.net.parameters()]
        return torch.cat([g.reshape(-1) for g in grads], out=out)

    def _unflatten_grads(self, flat):
        if self.flat_opt is not None:
            return                                 # the optimiser reads the flat buffer itself
        off = 0
        for p in self.net.parameters():
            p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
            off += p.numel()

    def _exchange(self, flat):
        self.n_grad_allreduce += 1
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        flat /= self.dist.get_world_size(self.group)

    def update(self, buf):
        c = self.cfg
        T, n = buf["nlp"].shape
        flat = lambda x: x.reshape(T * n, *x.shape[2:]) if x is not None else None  # noqa: E731
        adv = flat(buf["adv"])
        if c.normalize_advantage:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        src = dict(obs=flat(buf["obs"]), states=flat(buf["states"]), act=flat(buf["act"]), old_nlp=flat(buf["nlp"]),
                   ret=flat(buf["ret"]), adv=adv, old_mu=flat(buf["mu"]))
        total = T * n
        mb = max(1, total // c.minibatches)
        dev = src["obs"].device
        d = src
        acc = self._new_acc(dev)
        for v in acc.values():
            v.zero_()
        stats = {"kl": 0.0}
        count = 0
        for _ in range(c.mini_epochs):
            perm = torch.randperm(total, device=dev)
            acc["kl"].zero_()
            nmb = 0
            for s in range(0, total - mb + 1, mb):
                self._mb_backward(d, perm[s:s + mb], acc)
                if self.dist_on:
                    fl = self._flatten_grads()
                    self._exchange(fl)
                    self._unflatten_grads(fl)
                self._mb_apply(gathered=self.dist_on)
                nmb += 1
            count += nmb
            kl = acc["kl"] / max(nmb, 1)
            if self.dist_on:
                self.n_kl_allreduce += 1
                self.dist.all_reduce(kl, op=self.dist.ReduceOp.SUM, group=self.group)
                kl /= self.dist.get_world_size(self.group)
            kl = float(kl)                     # the one host sync per mini-epoch (adaptive learning rate)
            stats["kl"] = kl
            if kl > 2.0 * c.kl_threshold:      # rl_games AdaptiveScheduler
                self.lr = max(self.lr / 1.5, 1e-6)
            elif kl < 0.5 * c.kl_threshold:
                self.lr = min(self.lr * 1.5, 1e-2)
            if self.flat_opt is not None:
                self.flat_opt.set_lr(0, self.lr)
                if not self.net.central:
                    self.flat_opt.set_lr(1, self.lr)
            for g in (self.opt.param_groups[:1] if self.net.central else self.opt.param_groups):
                g["lr"] = self.lr
        for k in ("loss", "a_loss", "c_loss"):
            stats[k] = float(acc[k]) / max(count, 1)
        stats["lr"] = self.lr
        stats["mean_reward"] = float(buf["rew"].mean() / c.reward_scale)
        return stats

    def train(self, epochs, log=None, checkpoint_dir=None):
        """`epochs` PPO iterations.  With `checkpoint_dir` (rank 0 only): `<name>.pth` every `save_frequency` epochs and
        at the end, `<name>_best.pth` whenever the mean reward improves after `save_best_after` epochs."""
        import os
        out = []
        for _ in range(epochs):
            st = self.update(self.rollout())
            self.epoch += 1
            st["epoch"], st["frames"] = self.epoch - 1, self.frames
            out.append(st)
            if log:
                log(st)
            if checkpoint_dir:
                if self.epoch % max(self.cfg.save_frequency, 1) == 0:
                    self.save(os.path.join(checkpoint_dir, f"{self.cfg.name}.pth"))
                if self.epoch >= self.cfg.save_best_after and st["mean_reward"] > self.best_reward:
                    self.best_reward = st["mean_reward"]
                    self.save(os.path.join(checkpoint_dir, f"{self.cfg.name}_best.pth"))
            if self.epoch >= self.cfg.max_epochs:
                break
        if checkpoint_dir:
            self.save(os.path.join(checkpoint_dir, f"{self.cfg.name}.pth"))
        return out
