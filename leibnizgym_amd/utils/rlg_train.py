"""RL-Games glue and the training launcher (counterpart of reference leibnizgym/utils/rlg_train.py).

What the reference's module does, and where it is here:

* `parse_vec_task` (:33-69) - build the task named by `args.task` and wrap it in `VecTaskPython`.
* the env creator + `RlGamesGpuEnvAdapter` (:72-162) - RL-Games' `IVecEnv` view of that wrapper: `reset()` / `step()`
  return the SAME `{"obs", "states"}` dict object on every call when the env has global states (asymmetric
  actor-critic) and the per-step info travels as `[[], info]`; registered under 'RLGPU' / 'rlgpu' when `rl_games` is
  installed.
* `LeibnizAlgoObserver` (:165-206) -> `EnvInfoObserver`: the env's `info` scalars and the episode scores go to a scalar
  sink every time the algorithm prints its statistics (the algorithm's TensorBoard writer under RL-Games; a
  `ScalarSink` - TensorBoard when importable, CSV otherwise - under the in-repo trainer).
* `run_rlg_hydra` / `run_rlg` (:208-252): time-stamped run directory, `agent_config.yaml` and `env_config.yaml` dumps,
  seeding, then `Runner(observer).load(rlg) / reset() / run(args)`.  The runner is RL-Games' `torch_runner.Runner` when
  the package is installed, `NativeRunner` (the in-repo PPO of leibnizgym_amd/ppo.py behind the same three calls, with
  `args.checkpoint` / `args.play` / `args.train`) otherwise.
"""
import csv
import os
import random
import time
from collections import deque
from datetime import datetime
from types import SimpleNamespace

import numpy as np
import torch
import yaml

from ..envs.trifinger import TrifingerEnv as Trifinger
from ..wrappers.vec_task import VecTaskPython
from .errors import InvalidTaskNameError
from .helpers import print_info, print_notify

try:  # pragma: no cover - depends on the host image
    from rl_games.common import env_configurations, vecenv
    from rl_games.common.algo_observer import AlgoObserver as _AlgoObserver
    _IVecEnv = vecenv.IVecEnv
    HAVE_RL_GAMES = True
except Exception:
    env_configurations = vecenv = None
    _IVecEnv = _AlgoObserver = object
    HAVE_RL_GAMES = False

_TASKS = {"Trifinger": Trifinger}

# module-level launcher state, as in the reference (rlg_train.py:208-216): the env creator RL-Games calls takes no
# arguments of ours, so what it needs is parked here by `configure` / `run_rlg_hydra`
task_cfg = None
agent_cfg_train = None
cli_args = None
logdir = None
vargs = None


def parse_vec_task(args, cfg: dict, **env_kwargs) -> VecTaskPython:
    """args: namespace with task, task_type, device, ppo_device, headless, verbose (scripts/rlg_hydra.py:193-233)."""
    if args.task_type != "Python":
        raise ValueError(f"No task of type `{args.task_type}` in leibnizgym.")
    if args.device == "CPU":
        raise RuntimeError("device='CPU' selects the reference's IsaacGym CPU pipeline, which has no counterpart "
                           "here: the environment runs as HIP kernels on an MI355X (use device='GPU').")
    print_info("Running using python GPU...")
    sim_device = env_kwargs.pop("sim_device", "cuda:0")     # 'cuda:0' is the HIP device on PyTorch-ROCm
    ppo_device = env_kwargs.pop("ppo_device", sim_device)
    try:
        task_cls = _TASKS[args.task]
    except KeyError:
        raise InvalidTaskNameError(args.task)
    task = task_cls(config=cfg, device=sim_device, visualize=not args.headless, verbose=args.verbose, **env_kwargs)
    return VecTaskPython(task, rl_device=ppo_device, clip_obs=5, clip_actions=1)


# ---- the RL-Games view of the env -----------------------------------------------------------------------------------
_env_kwargs = {}          # test / multi-rank hook: extra constructor arguments of the task (lib=, sim_device=, env_id_offset=)


def create_rlgpu_env(frames: int = 1, **_unused):
    """The 'rlgpu' env creator: the configured task behind `VecTaskPython`; dumps `env_config.yaml` into the run
    directory; optional frame stacking through RL-Games' wrapper (reference :72-86)."""
    vec_env = parse_vec_task(cli_args, task_cfg, **dict(_env_kwargs))
    print_info(vec_env)
    if logdir:
        vec_env.dump_config(os.path.join(logdir, "env_config.yaml"))
    if frames > 1:
        from rl_games.common import wrappers
        vec_env = wrappers.FrameStack(vec_env, frames, False)
    return vec_env


class RlGamesGpuEnvAdapter(_IVecEnv):
    """`IVecEnv` over a `VecTaskPython`.  The observation handed to RL-Games is ONE dict object for the adapter's
    lifetime (RL-Games keeps a reference to it between calls); only its entries are replaced."""

    # The tensors this adapter hands out are the env's OWN buffers: they hold the values of a step until the next `step` / `reset` call overwrites them, on
    # the stream that call runs on.  A consumer that reads them before its next call (on that stream) needs no copy - leibnizgym_amd.ppo.PPOTrainer checks
    # this attribute and files obs / states of a rollout step straight out of them; an env without it (buffers refreshed asynchronously or on a side stream)
    # gets a clone per step.
    buffers_stable_until_next_step = True

    def __init__(self, config_name: str, num_actors: int, env=None, **kwargs):
        self.env = env if env is not None else env_configurations.configurations[config_name]["env_creator"](**kwargs)
        self.use_global_obs = self.env.num_states > 0
        self.full_state = {}
        self._refresh(self.env.reset())

    def _refresh(self, obs):
        self.full_state["obs"] = obs
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state
        return obs

    def reset(self):
        return self._refresh(self.env.reset())

    def step(self, action):
        obs, reward, is_done, info = self.env.step(action)
        return self._refresh(obs), reward, is_done, [[], info]

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        spaces = {"num_envs": self.env.num_envs, "action_space": self.env.action_space,
                  "observation_space": self.env.observation_space}
        if self.use_global_obs:
            spaces["state_space"] = self.env.state_space
        return spaces


if HAVE_RL_GAMES:  # pragma: no cover
    vecenv.register("RLGPU", lambda config_name, num_actors, **kw: RlGamesGpuEnvAdapter(config_name, num_actors, **kw))
    env_configurations.register("rlgpu", {"vecenv_type": "RLGPU", "env_creator": create_rlgpu_env})


# ---- scalar logging --------------------------------------------------------------------------------------------------
class ScalarSink:
    """`add_scalar(tag, value, step)` into TensorBoard when it is importable, into `<logdir>/scalars.csv` otherwise."""

    def __init__(self, log_dir: str):
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self._tb = self._csv = self._file = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir)
        except Exception:
            self._file = open(os.path.join(log_dir, "scalars.csv"), "w", newline="")
            self._csv = csv.writer(self._file)
            self._csv.writerow(["tag", "step", "value"])

    def add_scalar(self, tag, value, step):
        value = float(value)
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        else:
            self._csv.writerow([tag, step, repr(value)])
            self._file.flush()

    def close(self):
        if self._tb is not None:
            self._tb.close()
        elif self._file is not None:
            self._file.close()


class EnvInfoObserver(_AlgoObserver):
    """Logs what the env reports next to the algorithm's own statistics (reference `LeibnizAlgoObserver`, :165-206).
    The algorithm calls `process_infos(infos, done_indices)` after every env step - `infos` is the `[[], info]` pair of
    the adapter, so `infos[1]` is the env's scalar dict ('env/rewards/...', 'env/current_position_goal/...', ...) - and
    `after_print_stats(frame, epoch, time)` once per epoch; scores of finished games ('scores' / 'battle_won' entries
    of per-agent info dicts) feed a running mean over the last `games_to_track` games."""

    def __init__(self):
        self.algo = self.writer = None
        self.direct_info = {}
        self._scores = deque()

    def after_init(self, algo):
        self.algo = algo
        self.writer = algo.writer
        self._scores = deque(maxlen=int(getattr(algo, "games_to_track", 100)))
        self.direct_info = {}

    def process_infos(self, infos, done_indices):
        if not infos:
            return
        first = infos[0] if len(infos) > 0 else None
        if isinstance(first, dict):                       # per-agent dicts (not produced by this env; RL-Games convention)
            agents = max(int(getattr(self.algo, "num_agents", 1)), 1)
            for ind in done_indices:
                slot = int(ind) // agents
                if slot < len(infos):
                    for key in ("battle_won", "scores"):
                        if key in infos[slot]:
                            self._scores.append(float(np.asarray(infos[slot][key]).mean()))
        if len(infos) > 1 and isinstance(infos[1], dict):   # direct logging from the env
            self.direct_info = infos[1]

    def after_clear_stats(self):
        self._scores.clear()

    def after_print_stats(self, frame, epoch_num, total_time):
        if self.writer is None:
            return
        if len(self._scores) > 0:
            mean_scores = sum(self._scores) / len(self._scores)
            for axis, x in (("scores/mean", frame), ("scores/iter", epoch_num), ("scores/time", total_time)):
                self.writer.add_scalar(axis, mean_scores, x)
        for tag, value in self.direct_info.items():
            self.writer.add_scalar(tag, value, frame)


LeibnizAlgoObserver = EnvInfoObserver            # the reference's name, for code written against it


# ---- the in-repo runner behind RL-Games' Runner interface ------------------------------------------------------------------
class NativeRunner:
    """`load(rlg) / reset() / run(args)` like `rl_games.torch_runner.Runner`, driving leibnizgym_amd.ppo.PPOTrainer on the
    'rlgpu' env.  `args` is the launcher's `args` dict: `train` / `play`, `checkpoint` (restored before either),
    `logdir`; `max_epochs` comes from the agent tree (override with the `TF_MAX_EPOCHS` environment variable)."""

    def __init__(self, algo_observer=None):
        self.observer = algo_observer
        self.params = None
        self.trainer = None
        self.writer = None
        self.games_to_track, self.num_agents, self.ppo_device = 100, 1, "cuda:0"

    def load(self, rlg: dict):
        self.params = rlg
        return self

    def reset(self):
        self.trainer = None

    def _build(self, args):
        from ..ppo import PPOConfig, PPOTrainer
        conf = self.params["params"]["config"]
        vec_env = create_rlgpu_env()
        self.ppo_device = str(vec_env.rl_device)
        adapter = RlGamesGpuEnvAdapter(conf.get("env_name", "rlgpu"), vec_env.num_envs, env=vec_env)
        cfg = PPOConfig.from_rlg(self.params, num_envs=vec_env.num_envs)
        if os.environ.get("TF_MAX_EPOCHS"):
            cfg.max_epochs = int(os.environ["TF_MAX_EPOCHS"])
        self.trainer = PPOTrainer(adapter, vec_env.num_obs, vec_env.num_states, vec_env.num_actions, cfg,
                                  device=self.ppo_device)
        ckpt = args.get("checkpoint") or (self.params["params"].get("load_path") if self.params["params"].get("load_checkpoint") else "")
        if ckpt:
            print_notify(f"Restoring checkpoint: {ckpt}")
            self.trainer.restore(ckpt)
        self.writer = ScalarSink(os.path.join(args.get("logdir") or "runs", "summaries"))
        if self.observer is not None:
            self.observer.after_init(self)
        return cfg

    def run(self, args):
        args = dict(args) if not isinstance(args, dict) else args
        cfg = self._build(args)
        tr, t0 = self.trainer, time.perf_counter()
        if args.get("play") or not args.get("train", True):
            steps = int(os.environ.get("TF_PLAY_STEPS", "750"))
            mean_reward, info = tr.play(steps)
            if self.observer is not None:
                self.observer.process_infos([[], info], [])
                self.observer.after_print_stats(tr.frames, tr.epoch, time.perf_counter() - t0)
            print_notify(f"play: {steps} steps, mean reward per step {mean_reward:.4f}")
            self.writer.close()
            return {"mean_reward": mean_reward}

        def log(st):
            total_time = time.perf_counter() - t0
            for k in ("loss", "a_loss", "c_loss", "kl", "lr", "mean_reward"):
                self.writer.add_scalar(f"losses/{k}" if k.endswith("loss") else f"info/{k}", st[k], st["frames"])
            if self.observer is not None:
                self.observer.process_infos([[], tr.last_info], [])
                self.observer.after_print_stats(st["frames"], st["epoch"], total_time)
            if conf_print:
                print(f"epoch {st['epoch']:5d} frames {st['frames']:11d} reward/step {st['mean_reward']:9.3f} kl {st['kl']:.4f} "
                      f"lr {st['lr']:.2e}  {st['frames'] / total_time:.3e} frames/s", flush=True)
        conf_print = bool(self.params["params"]["config"].get("print_stats", True))
        stats = tr.train(cfg.max_epochs, log, checkpoint_dir=os.path.join(args.get("logdir") or ".", "nn"))
        self.writer.close()
        return stats


# ---- launcher ------------------------------------------------------------------------------------------------------------
def configure(gym_cfg: dict, args, log_dir: str = None, agent_cfg: dict = None, **env_kwargs):
    """Set the module-level configuration the env creator reads (what run_rlg_hydra does in the reference)."""
    global task_cfg, cli_args, logdir, agent_cfg_train, vargs, _env_kwargs
    task_cfg, logdir, agent_cfg_train = gym_cfg, log_dir, agent_cfg
    vargs = dict(args) if isinstance(args, dict) else dict(vars(args))
    cli_args = SimpleNamespace(**vargs)
    _env_kwargs = dict(env_kwargs)


def set_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def run_rlg_hydra(cfg: dict, runner_factory=None, **env_kwargs):
    """`cfg` = {gym, rlg, args} as composed by leibnizgym_amd.config.compose (reference :208-216 takes the OmegaConf tree)."""
    configure(cfg["gym"], cfg["args"], cfg["args"].get("logdir", "logs/"), cfg["rlg"], **env_kwargs)
    return run_rlg(runner_factory)


def run_rlg(runner_factory=None):
    """Reference :219-252: default directories, time-stamped run directory, agent configuration dump, seed, runner."""
    global logdir
    os.makedirs("nn", exist_ok=True)
    os.makedirs("runs", exist_ok=True)
    np.set_printoptions(edgeitems=30, infstr="inf", linewidth=4000, nanstr="nan", precision=2, suppress=False, threshold=10000)
    logdir = os.path.join(logdir or "logs/", datetime.now().strftime("%m-%d-%Y-%H-%M-%S"))
    os.makedirs(logdir, exist_ok=True)
    print_notify(f"Saving logs at: {logdir}")
    print_notify(f"Verbosity     : {cli_args.verbose}")
    print_notify(f"Seed          : {agent_cfg_train['seed']}")
    cli_args.logdir = logdir
    vargs["logdir"] = logdir
    set_seed(agent_cfg_train["seed"])
    if cli_args.verbose:
        print_info("Agent training configuration: ")
        print(yaml.dump(agent_cfg_train))
        print(40 * "-")
    with open(os.path.join(logdir, "agent_config.yaml"), "w") as f:
        yaml.dump(agent_cfg_train, f)
    if runner_factory is None:
        if HAVE_RL_GAMES:  # pragma: no cover
            from rl_games.torch_runner import Runner as runner_factory
        else:
            runner_factory = NativeRunner
    runner = runner_factory(EnvInfoObserver())
    runner.load(agent_cfg_train)
    runner.reset()
    return runner.run(vargs)
