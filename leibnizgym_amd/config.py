"""Hydra-schema-compatible configuration without hydra/omegaconf.

Reproduces the structured configs of the reference launcher (scripts/rlg_hydra.py:15-249): group `gym` with
`trifinger_difficulty_{1..4}`, the `args` block, and `update_cfg` (:251-286).  `compose(["gym=trifinger_difficulty_4",
"args.num_envs=65536", "args.headless=True"])` returns plain nested dicts shaped like
`OmegaConf.to_container(cfg)`.

Explicit fix of a reference hazard: the difficulty subclasses assign `task_difficulty = N` WITHOUT a type
annotation (scripts/rlg_hydra.py:123,128,133,138), so under plain `dataclasses` the field keeps its MISSING
default; here the value is set explicitly.
"""
import copy
import os

import yaml

MISSING = "???"

SIM_CONFIG = {      # SimConfig, scripts/rlg_hydra.py:15-41
    "dt": 0.02, "substeps": 4, "up_axis": "z", "use_gpu_pipeline": MISSING, "num_client_threads": 0,
    "gravity": [0.0, 0.0, -9.81],
    "physx": {"num_threads": 4, "solver_type": 1, "use_gpu": False, "num_position_iterations": 8,
              "num_velocity_iterations": 0, "contact_offset": 0.002, "rest_offset": 0.0,
              "bounce_threshold_velocity": 0.5, "max_depenetration_velocity": 1000.0,
              "default_buffer_size_multiplier": 5.0},
    "flex": {"num_outer_iterations": 5, "num_inner_iterations": 20, "warm_start": 0.8, "relaxation": 0.75},
}

TRIFINGER = {       # EnvConfig + Trifinger, scripts/rlg_hydra.py:43-118
    "env_name": "Trifinger", "num_instances": MISSING, "seed": MISSING, "spacing": 1.0, "aggregate_mode": True,
    "control_decimation": 1, "physics_engine": MISSING, "sim": SIM_CONFIG,
    "episode_length": 750, "task_difficulty": MISSING, "enable_ft_sensors": False,
    "asymmetric_obs": False, "normalize_obs": True, "apply_safety_damping": True,
    "command_mode": "torque", "normalize_action": True,
    "reset_distribution": {"object_initial_state": {"type": "random"},
                           "robot_initial_state": {"dof_pos_stddev": 0.4, "dof_vel_stddev": 0.2, "type": "default"}},
    "reward_terms": {
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -750},
        "object_dist": {"activate": True, "weight": 2000},
        "object_rot": {"activate": False, "weight": 300},
        "object_rot_delta": {"activate": False, "weight": -250},
        "object_move": {"activate": False, "weight": -750}},
    "termination_conditions": {"success": {"activate": False, "bonus": 5000.0, "orientation_tolerance": 0.1,
                                           "position_tolerance": 0.01}},
}

_D4_OVERRIDES = {   # TrifingerDifficulty4, scripts/rlg_hydra.py:135-182
    "episode_length": 750,
    "reward_terms": {
        "finger_move_penalty": {"activate": True, "weight": -0.1},
        "finger_reach_object_rate": {"activate": True, "norm_p": 2, "weight": -250,
                                     "thresh_sched_start": 0, "thresh_sched_end": 1e7},
        "object_dist": {"activate": True, "weight": 2000, "thresh_sched_start": 0, "thresh_sched_end": 10e10},
        "object_rot": {"activate": True, "weight": 2000, "epsilon": 0.01, "scale": 3.0,
                       "thresh_sched_start": 1e7, "thresh_sched_end": 1e10},
        "object_rot_delta": {"activate": False, "weight": -250},
        "object_move": {"activate": False, "weight": -750}},
    "termination_conditions": {"success": {"activate": False, "bonus": 5000.0, "orientation_tolerance": 0.25,
                                           "position_tolerance": 0.02}},
}

ARGS = {            # Args, scripts/rlg_hydra.py:193-233
    "cfg_env": "Base", "cfg_train": "Base", "task": "Trifinger", "task_type": "Python", "experiment_name": "Base",
    "num_envs": 256, "randomize": False, "seed": 7, "verbose": False, "logdir": "logs/",
    "physics_engine": "physx", "device": "GPU", "ppo_device": "GPU", "play": False, "train": MISSING,
    "checkpoint": "", "headless": False, "compute_device_id": 0, "graphics_deice_id": 0,
    "wandb_project_name": "trifinger-manip", "wandb_log": True,
}

GYM_GROUP = ("trifinger_difficulty_1", "trifinger_difficulty_2", "trifinger_difficulty_3", "trifinger_difficulty_4")

def _mlp_block(init_name: str, init_scale: float) -> dict:
    return {"units": [400, 200, 100], "activation": "elu", "d2rl": False,
            "initializer": {"name": init_name, "scale": init_scale}, "regularizer": {"name": "None"}}


# The agent group `rlg=asymm` (reference resources/config/rlg/asymm.yaml:1-90) as data: asymmetric actor-critic for
# RL-Games' a2c_continuous - actor on `obs`, central value network on `states`.  RL-Games consumes this tree as is
# (`Runner.load`); the in-repo trainer (leibnizgym_amd/ppo.py) reads its hyper-parameters from the same tree
# (`PPOConfig.from_rlg`).  YAML spells None as the string 'None' in two places; kept.
RLG_ASYMM = {
    "asymmetric_obs": True,
    "params": {
        "algo": {"name": "a2c_continuous"},
        "model": {"name": "continuous_a2c_logstd"},
        "network": {
            "separate": True, "name": "actor_critic",
            "space": {"continuous": {
                "mu_activation": "None", "sigma_activation": "None",
                "mu_init": {"name": "variance_scaling_initializer", "scale": 0.02},
                "sigma_init": {"name": "const_initializer", "val": 0},
                "fixed_sigma": True}},
            "mlp": _mlp_block("default", 2),
        },
        "load_checkpoint": False, "load_path": "nn/weights.pth",
        "config": {
            "name": "trifinger", "env_name": "rlgpu", "ppo": True, "normalize_input": False,
            "reward_shaper": {"scale_value": 0.01}, "normalize_advantage": True,
            "gamma": 0.99, "tau": 0.95, "learning_rate": 3e-4, "lr_schedule": "adaptive", "lr_threshold": 0.008,
            "score_to_win": 1000000, "max_epochs": 100000, "save_best_after": 500, "save_frequency": 100,
            "preemption_checkpoint_freq": 500, "print_stats": True, "grad_norm": 1.0, "entropy_coef": 0.0,
            "truncate_grads": True, "e_clip": 0.2, "steps_num": 32, "minibatch_size": 8192, "mini_epochs": 4,
            "critic_coef": 4, "clip_value": False, "seq_len": 4, "bounds_loss_coef": 0.0001,
            "central_value_config": {
                "seq_length": 4, "minibatch_size": 8192, "mini_epochs": 4, "lr": 5e-4, "clip_value": False,
                "normalize_input": False, "grad_norm": 1.0, "truncate_grads": True,
                "network": {"name": "actor_critic", "central_value": True,
                            "mlp": _mlp_block("variance_scaling_initializer", 2)},
            },
        },
    },
}


def gym_config(name: str) -> dict:
    if name not in GYM_GROUP:
        raise KeyError(f"Could not find 'gym/{name}'. Available options in 'gym': {list(GYM_GROUP)}")
    cfg = copy.deepcopy(TRIFINGER)
    d = int(name.rsplit("_", 1)[1])
    cfg["task_difficulty"] = d
    if d == 4:
        cfg.update(copy.deepcopy(_D4_OVERRIDES))
    return cfg


def _parse_scalar(text: str):
    return yaml.safe_load(text)


def _set_path(d: dict, path: str, value):
    keys = path.split(".")
    for k in keys[:-1]:
        d = d.setdefault(k, {})
    d[keys[-1]] = value


def update_cfg(cfg: dict) -> dict:
    """scripts/rlg_hydra.py:251-286, on plain dicts."""
    a, g, r = cfg["args"], cfg["gym"], cfg["rlg"]
    a["train"] = not a["play"]
    g["num_instances"] = a["num_envs"]
    g["sim"]["use_gpu_pipeline"] = a["device"] == "GPU"
    g["sim"]["physx"]["use_gpu"] = a["device"] == "GPU"
    g["physics_engine"] = a["physics_engine"]
    g["asymmetric_obs"] = r["asymmetric_obs"]
    params = r.setdefault("params", {})
    conf = params.setdefault("config", {})
    if a["experiment_name"] != "Base":
        conf["name"] = (f"{a['experiment_name']}_{a['task_type']}_{a['device']}_"
                        f"{str(a['physics_engine']).split('_')[-1]}")
    params["load_checkpoint"] = a["checkpoint"] != ""
    params["load_path"] = a["checkpoint"]
    conf["minibatch_size"] = a["num_envs"]
    conf["num_actors"] = a["num_envs"]
    if "central_value_config" in conf:
        conf["central_value_config"]["minibatch_size"] = a["num_envs"]
    g["seed"] = a["seed"]
    r["seed"] = a["seed"]
    return cfg


def compose(overrides=(), rlg_yaml: str = None) -> dict:
    """Build {gym, rlg, args, output_root} from hydra-style overrides; defaults as resources/config/config.yaml:5-8
    (gym: trifinger_difficulty_1, rlg: asymm)."""
    gym_name, rlg_name, rest = "trifinger_difficulty_1", "asymm", []
    for ov in overrides:
        key, _, val = ov.partition("=")
        if key == "gym":
            gym_name = val
        elif key == "rlg":
            rlg_name = val
        else:
            rest.append((key, val))
    if rlg_yaml is not None and os.path.isfile(rlg_yaml):
        with open(rlg_yaml) as f:
            rlg = yaml.safe_load(f)
    elif rlg_name == "asymm":
        rlg = copy.deepcopy(RLG_ASYMM)
    else:
        raise KeyError(f"Could not find 'rlg/{rlg_name}'")
    cfg = {"gym": gym_config(gym_name), "rlg": rlg, "args": copy.deepcopy(ARGS), "output_root": "./output"}
    for key, val in rest:
        _set_path(cfg, key, _parse_scalar(val))
    return update_cfg(cfg)
