"""RL-Games glue (counterpart of reference leibnizgym/utils/rlg_train.py:33-162).

`RlGamesGpuEnvAdapter` gives RL-Games' `IVecEnv` view of a `VecTaskPython`: `reset()`/`step()` return the
SAME `{"obs", "states"}` dict object every call when the env has global states (asymmetric actor-critic),
and the info travels as `[[], info]` (rlg_train.py:144-154).  `rl_games` is optional: when it is installed the
adapter derives from `vecenv.IVecEnv` and registers itself under 'RLGPU' / 'rlgpu' exactly as the reference does.
"""
import os
from types import SimpleNamespace

from ..envs.trifinger import TrifingerEnv as Trifinger
from ..wrappers.vec_task import VecTaskPython
from .errors import InvalidTaskNameError
from .helpers import print_info

try:  # pragma: no cover - depends on the host image
    from rl_games.common import env_configurations, vecenv
    _IVecEnv = vecenv.IVecEnv
    HAVE_RL_GAMES = True
except Exception:
    env_configurations = vecenv = None
    _IVecEnv = object
    HAVE_RL_GAMES = False

_TASKS = {"Trifinger": Trifinger}

# module-level state filled by the launcher, as in the reference (rlg_train.py:208-216)
task_cfg = None
cli_args = None
logdir = None


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


def create_rlgpu_env(**kwargs):
    env = parse_vec_task(cli_args, task_cfg)
    print_info(env)
    if logdir:
        env.dump_config(os.path.join(logdir, 'env_config.yaml'))
    frames = kwargs.pop('frames', 1)
    if frames > 1:
        from rl_games.common import wrappers
        env = wrappers.FrameStack(env, frames, False)
    return env


class RlGamesGpuEnvAdapter(_IVecEnv):
    def __init__(self, config_name: str, num_actors: int, env=None, **kwargs):
        if env is not None:
            self.env = env
        else:
            self.env = env_configurations.configurations[config_name]['env_creator'](**kwargs)
        self.use_global_obs = (self.env.num_states > 0)
        self.full_state = {"obs": self.env.reset()}
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {'num_envs': self.env.num_envs, 'action_space': self.env.action_space,
                'observation_space': self.env.observation_space}
        if self.use_global_obs:
            info['state_space'] = self.env.state_space
        return info

    def reset(self):
        self.full_state["obs"] = self.env.reset()
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state
        return self.full_state["obs"]

    def step(self, action):
        next_obs, reward, is_done, info = self.env.step(action)
        self.full_state["obs"] = next_obs
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state, reward, is_done, [[], info]
        return self.full_state["obs"], reward, is_done, [[], info]


if HAVE_RL_GAMES:  # pragma: no cover
    vecenv.register('RLGPU', lambda config_name, num_actors, **kwargs: RlGamesGpuEnvAdapter(config_name, num_actors, **kwargs))
    env_configurations.register('rlgpu', {'vecenv_type': 'RLGPU', 'env_creator': lambda **kwargs: create_rlgpu_env(**kwargs)})


def configure(gym_cfg: dict, args, log_dir: str = None):
    """Set the module-level configuration the env creator reads (what run_rlg_hydra does in the reference)."""
    global task_cfg, cli_args, logdir
    task_cfg, cli_args, logdir = gym_cfg, args if not isinstance(args, dict) else SimpleNamespace(**args), log_dir
