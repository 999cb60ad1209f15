"""Host-side API and step sequencing (T1-T3, T14, T15 and the quirk list of SURVEY.md 8a-Q).  Every test takes the
`backend` fixture (tests/conftest.py): the CPU suite runs it with the oracle library injected through the test hook
`lib=` (the product never loads it by itself), `-m gpu` runs the same test on the HIP library on cuda:0.

Known answers are hand-derived from the cited reference lines; nothing here depends on the physics details.
"""
import os

import numpy as np
import pytest
import torch

from leibnizgym_amd.config import compose, gym_config
from leibnizgym_amd.envs import IsaacEnvBase, TrifingerEnv
from leibnizgym_amd.utils.errors import InvalidTaskNameError
from leibnizgym_amd.utils.rlg_train import RlGamesGpuEnvAdapter
from leibnizgym_amd.wrappers import VecTaskPython


def make_env(backend, **cfg):
    base = {"num_instances": 4, "command_mode": "torque"}
    base.update(cfg)
    return TrifingerEnv(config=base, device=backend[1], verbose=False, lib=backend[0])


def test_shapes_specs_and_getters(backend):
    env = make_env(backend, asymmetric_obs=True)
    assert isinstance(env, IsaacEnvBase)
    assert env.get_num_instances() == 4
    assert env.get_obs_dim() == 41 and env.get_state_dim() == 113 and env.get_action_dim() == 9
    assert tuple(env.get_obs_shape()) == (4, 41) and tuple(env.get_action_shape()) == (4, 9)
    assert sum(env.obs_spec.values()) == 41 and sum(env.state_spec.values()) == 113
    assert env.config["enable_ft_sensors"] is True                       # forced by asymmetric_obs (:272-273)
    env2 = make_env(backend, command_mode="position_impedance")
    assert env2.get_action_dim() == 18 and env2.get_obs_dim() == 50 and env2.get_state_dim() == 0
    # scale tables (trifinger_env.py:663-710)
    assert env._observations_scale.low.shape[0] == 41 and env._states_scale.high.shape[0] == 113
    assert torch.equal(env._action_scale.high.cpu(), torch.full((9,), 0.36))
    # buffer dtypes of the reference (env_base.py:560-572): float obs / states / reward, torch.long step counter
    assert env._steps_count_buf.dtype == torch.long and env._steps_count_buf.shape == (4,)
    assert env._obs_buf.dtype == torch.float32 and env._reward_buf.dtype == torch.float32


def test_reset_and_step_contract(backend):
    env = make_env(backend)
    assert env.env_steps_count == 0
    obs = env.reset()
    assert obs.shape == (4, 41) and obs.data_ptr() != env.obs_buf.data_ptr()          # reset returns a clone
    assert env.env_steps_count == 4                                                   # one simulate x 4 instances
    assert int(env._steps_count_buf.sum()) == 0
    out = env.step(torch.zeros(4, 9))
    obs2, rew, dones, info = out
    assert obs2.data_ptr() == env.obs_buf.data_ptr()                                  # step returns the live buffer
    assert rew.shape == (4,) and dones.dtype == torch.bool and dones.shape == (4,)
    assert env.env_steps_count == 8 and env._steps_count_buf.tolist() == [1, 1, 1, 1]
    keys = set(info)
    assert {"env/current_position_goal/count", "env/current_orientation_goal/count",
            "env/average_consecutive_success", "env/rewards/object_dist"} <= keys
    # numpy actions are accepted (env_base.py:361-362)
    env.step(np.zeros((4, 9), dtype=np.float32))
    assert env._steps_count_buf.tolist() == [2, 2, 2, 2]


def test_invalid_inputs_raise_like_the_reference(backend):
    env = make_env(backend)
    env.reset()
    with pytest.raises(ValueError, match="Invalid shape for tensor `action`"):
        env.step(torch.zeros(4, 8))
    with pytest.raises(ValueError, match="Invalid command mode"):
        make_env(backend, command_mode="velocity")
    with pytest.raises(ValueError, match="Invalid difficulty index"):
        make_env(backend, task_difficulty=7)
    with pytest.raises(ValueError, match="Invalid robot initial state distribution"):
        make_env(backend, reset_distribution={"robot_initial_state": {"type": "gaussian"}})
    with pytest.raises(ValueError, match="Invalid object initial state distribution"):
        make_env(backend, reset_distribution={"object_initial_state": {"type": "grid"}})
    with pytest.raises(ValueError, match="Invalid physics engine backend"):
        make_env(backend, physics_engine="bullet")
    with pytest.raises(ValueError, match="Invalid physics up-axis"):
        make_env(backend, sim={"up_axis": "x"})
    with pytest.raises(RuntimeError, match="MI355X"):
        TrifingerEnv(config={"num_instances": 2, "command_mode": "torque"}, device="cpu", verbose=False)


def test_timeout_reset_sequencing(backend):
    """Quirks 1-3: time-out uses steps >= episode_length after the increment; the reset happens at the START of
    the next step, before physics, and zeroes that env's action for the step (env_base.py:370-395)."""
    env = make_env(backend, episode_length=3, termination_conditions={"success": {"activate": False}})
    env.reset()
    act = torch.full((4, 9), 0.5)
    for k in range(1, 4):
        _, _, dones, _ = env.step(act)
        assert env._steps_count_buf.tolist() == [k] * 4
        assert env._reset_buf.tolist() == [k >= 3] * 4
        assert not dones.any()                       # quirk 2: dones = reset & goal_reset, never true here
        assert torch.equal(env.action_buf.cpu(), act)
    counts_before = env._engine.reset_count.clone()
    env.step(act)                                    # reset applied first, then one physics step
    assert env._steps_count_buf.tolist() == [1] * 4 and not env._reset_buf.any()
    assert torch.equal(env.action_buf.cpu(), torch.zeros(4, 9))      # trifinger_env.py:387
    assert torch.equal(env.obs_buf[:, 32:41].cpu(), torch.zeros(4, 9))   # normalised zero action in the observation
    assert (env._engine.reset_count == counts_before + 1).all()
    q = env._dof_position.cpu()
    assert (q - torch.tensor([0.0, 0.9, -1.7] * 3)).abs().max() < 0.15   # default pose + one step of sag
    assert env.dones_buf.data_ptr() == env._reset_buf.data_ptr()    # quirk: dones_buf IS the reset buffer (:281-284)


def test_partial_reset_only_touches_flagged_envs(backend):
    env = make_env(backend, episode_length=0)
    env.reset()
    env.step(torch.zeros(4, 9))
    before = env._engine.state.clone()
    env._reset_buf[2] = True
    goal_before = env._object_goal_poses_buf.clone()
    env.step(torch.full((4, 9), 0.3))
    assert env.action_buf[2].abs().max() == 0 and (env.action_buf[[0, 1, 3]] == 0.3).all()
    g = env._object_goal_poses_buf
    assert torch.equal(g[[0, 1, 3]], goal_before[[0, 1, 3]]) and not torch.equal(g[2], goal_before[2])
    assert env._engine.reset_count.tolist() == [1, 1, 2, 1]
    assert env._steps_count_buf.tolist() == [2, 2, 1, 2]
    assert before.shape == env._engine.state.shape


def test_success_goal_reset_and_dones(backend):
    """Success termination on: goal_reset_buf = hit, bonus added, successes |= hit, goal resampled next step
    (trifinger_env.py:1088-1094, 425-440); dones only when a time-out coincides with a hit."""
    env = make_env(backend, episode_length=2, task_difficulty=1,
                   termination_conditions={"success": {"activate": True, "bonus": 123.0, "position_tolerance": 10.0,
                                                       "orientation_tolerance": 10.0}},
                   reward_terms={k: {"activate": False} for k in
                                 ("finger_reach_object_rate", "finger_move_penalty", "object_dist", "object_rot",
                                  "object_rot_delta", "object_move")})
    env.reset()
    _, rew, dones, info = env.step(torch.zeros(4, 9))
    assert torch.equal(rew.cpu(), torch.full((4,), 123.0))                    # huge tolerance: every env hits
    assert env._goal_reset_buf.all() and env._successes.all() and not dones.any()
    assert float(info["env/average_consecutive_success"]) == 1.0
    assert float(info["env/current_position_goal/count"]) == 4.0
    goal0 = env._object_goal_poses_buf.clone()
    _, _, dones, _ = env.step(torch.zeros(4, 9))                        # goal resampled at the start of this step
    assert not torch.equal(env._object_goal_poses_buf, goal0)
    assert dones.all()                                                  # steps == 2 -> time-out AND goal hit
    assert env._engine.reset_count.tolist() == [2] * 4                  # reset() + one goal reset


def test_reward_schedule_follows_env_steps_count(backend):
    """object_rot switches on when env_steps_count >= 1e7 (difficulty-4 schedule, scripts/rlg_hydra.py:160-167)."""
    cfg = gym_config("trifinger_difficulty_4")
    cfg.update(num_instances=4, seed=1, physics_engine="physx")
    cfg["sim"]["use_gpu_pipeline"] = True
    env = TrifingerEnv(config=cfg, device=backend[1], verbose=False, lib=backend[0])
    env.reset()
    _, _, _, info = env.step(torch.zeros(4, 9))
    assert float(info["env/rewards/object_rot"]) == 0.0
    assert float(info["env/rewards/finger_reach_object_rate"]) != 0.0
    env._engine.frame_count = 2_500_000                                 # 2.5e6 frames x 4 envs = 1e7
    _, _, _, info = env.step(torch.zeros(4, 9))
    assert float(info["env/rewards/object_rot"]) > 0.0
    assert float(info["env/rewards/finger_reach_object_rate"]) == 0.0   # its window [0, 1e7] just closed
    assert env.env_steps_count == 4 * 2_500_001


def test_control_decimation_repeats_simulate(backend):
    # safety damping off: the torque of a zero action is then exactly zero, whatever the joint velocity
    a = make_env(backend, control_decimation=1, episode_length=0, apply_safety_damping=False)
    b = make_env(backend, control_decimation=5, episode_length=0, apply_safety_damping=False)   # reference tests use 5
    a.reset(), b.reset()
    b.step(torch.zeros(4, 9))
    assert b.env_steps_count == (1 + 5) * 4
    for _ in range(5):
        a.step(torch.zeros(4, 9))
    # same physics sequence (zero torque) -> same cube state; fingers too
    assert torch.allclose(a._engine.cube, b._engine.cube, atol=1e-6)
    assert torch.allclose(a._engine.q, b._engine.q, atol=1e-6)


def test_vec_task_clamps_and_spaces(backend):
    env = make_env(backend, asymmetric_obs=True)
    vec = VecTaskPython(env, rl_device="cpu", clip_obs=5.0, clip_actions=1.0)
    assert (vec.num_envs, vec.num_obs, vec.num_states, vec.num_actions) == (4, 41, 113, 9)
    assert vec.observation_space.shape == (41,) and float(vec.observation_space.high[0]) == 5.0
    assert vec.action_space.shape == (9,) and float(vec.action_space.low[0]) == -1.0
    obs = vec.reset()
    assert obs.abs().max() <= 5.0
    big = torch.full((4, 9), 7.0)
    obs, rew, done, info = vec.step(big)
    assert torch.equal(env.action_buf.cpu(), torch.ones(4, 9))                # clipped before the task sees it
    assert obs.abs().max() <= 5.0 and vec.get_state().abs().max() <= 5.0
    assert "Number of observations: 41" in str(vec)
    with pytest.raises(AssertionError):
        VecTaskPython(object(), rl_device="cpu")


def test_fused_clipping_equals_wrapper_clamps(backend):
    """The clamps fused into the native step give what the literal three-operation wrapper gives (vec_task.py:146-170),
    also when they bite: tiny bounds, actions outside +-clip_actions."""
    kw = dict(asymmetric_obs=True, command_mode="torque", num_instances=64)
    outs = []
    for fused in (True, False):
        env = make_env(backend, **kw)
        vec = VecTaskPython(env, rl_device="cpu", clip_obs=0.3, clip_actions=0.6, fuse_clipping=fused)
        assert vec._fused is fused
        g = torch.Generator().manual_seed(4)
        o = [vec.reset().clone()]
        for _ in range(12):
            obs, rew, done, _ = vec.step(torch.rand(64, 9, generator=g) * 4 - 2)
            o += [obs.clone(), vec.get_state().clone(), rew.clone(), env.action_buf.cpu().clone()]
        outs.append(o)
        env.close()
    assert any(bool((t.abs() == 0.3).any()) for t in outs[0][1::4])      # the observation clamp did bite
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_rl_games_adapter_contract(backend):
    env = make_env(backend, asymmetric_obs=True)
    ad = RlGamesGpuEnvAdapter("rlgpu", 4, env=VecTaskPython(env, rl_device="cpu"))
    assert ad.use_global_obs and set(ad.full_state) == {"obs", "states"}
    info = ad.get_env_info()
    assert set(info) == {"num_envs", "action_space", "observation_space", "state_space"} and info["num_envs"] == 4
    first = ad.reset()
    out, rew, done, extra = ad.step(torch.zeros(4, 9))
    assert out is first is ad.full_state                                 # the SAME dict object every call
    assert out["obs"].shape == (4, 41) and out["states"].shape == (4, 113)
    assert isinstance(extra, list) and extra[0] == [] and "env/average_consecutive_success" in extra[1]
    sym = RlGamesGpuEnvAdapter("rlgpu", 4, env=VecTaskPython(make_env(backend), rl_device="cpu"))
    assert not sym.use_global_obs and torch.is_tensor(sym.reset())
    assert ad.get_number_of_agents() == 1


def test_dump_config_and_seed(backend, tmp_path):
    env = make_env(backend)
    path = os.path.join(tmp_path, "sub", "env_config")
    env.dump_config(path)
    import yaml
    d = yaml.safe_load(open(path + ".yaml"))
    assert d["num_instances"] == 4 and d["command_mode"] == "torque" and d["sim"]["dt"] == 0.02
    TrifingerEnv.seed(3)
    a = torch.rand(2)
    TrifingerEnv.seed(3)
    assert torch.equal(a, torch.rand(2))
    env.render(), env.close()


def test_same_seed_same_trajectory_and_two_instances_coexist(backend):
    a, b = make_env(backend, seed=11), make_env(backend, seed=11)
    c = make_env(backend, seed=12)
    oa, ob, oc = a.reset(), b.reset(), c.reset()
    assert torch.equal(oa, ob) and not torch.equal(oa, oc)
    g = torch.Generator().manual_seed(0)
    for _ in range(5):
        act = torch.rand(4, 9, generator=g) * 2 - 1
        ra = a.step(act)[1].clone()
        rb = b.step(act)[1].clone()
        assert torch.equal(ra, rb)


def test_hydra_schema_loader():
    cfg = compose(["gym=trifinger_difficulty_4", "args.num_envs=8192", "args.headless=True", "args.seed=3"])
    g = cfg["gym"]
    assert g["task_difficulty"] == 4 and g["num_instances"] == 8192 and g["seed"] == 3
    assert g["asymmetric_obs"] is True                                   # copied from rlg.asymmetric_obs (:266)
    assert g["sim"]["use_gpu_pipeline"] is True and g["physics_engine"] == "physx"
    assert g["reward_terms"]["object_rot"]["thresh_sched_start"] == 1e7
    assert g["termination_conditions"]["success"]["position_tolerance"] == 0.02
    assert cfg["rlg"]["params"]["config"]["num_actors"] == 8192 and cfg["args"]["train"] is True
    d1 = compose([])["gym"]
    assert d1["task_difficulty"] == 1 and d1["num_instances"] == 256 and d1["command_mode"] == "torque"
    with pytest.raises(KeyError):
        gym_config("trifinger_difficulty_9")
    with pytest.raises(InvalidTaskNameError):
        raise InvalidTaskNameError("Foo")


def test_fingertip_history_keeps_the_pre_reset_tips_for_one_step(backend):
    """Quirk 4 of SURVEY 8a-Q: `_reset_impl` zeroes only the OLDER fingertip history entry, which the next fill shifts out
    (trifinger_env.py:1146-1147, :974), so on the first step after a reset the fingertip terms difference against the
    PRE-reset fingertips - a spurious jump - while the object history is rewritten with the reset pose (:1183-1187) and the
    object terms are clean.  Checked with one term active at a time against the host-side reward classes."""
    from leibnizgym_amd import _capi as capi
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    from leibnizgym_amd.envs.trifinger import rewards as rw
    n, dt = 64, 0.02
    only = lambda name, **kw: {k: dict(activate=(k == name), **(kw if k == name else {})) for k in capi.REWARD_TERM_ORDER}  # noqa: E731
    for term in ("finger_move_penalty", "object_move"):
        weight = -0.1 if term == "finger_move_penalty" else -750.0
        cfg = make_config(backend[0], n, seed=6, command_mode="torque", robot_reset="random", episode_length=5,
                          reward_terms=only(term, weight=weight), success={"activate": False}, task_difficulty=1)
        eng = TrifingerEngine(cfg, device=backend[1], lib=backend[0])
        eng.reset()
        g = torch.Generator().manual_seed(1)
        for t in range(7):
            tips_before = eng.state[capi.S_TIP_P:capi.S_TIP_P + 9].T.cpu().clone().view(n, 3, 3)      # history[0] going in
            was_flagged = eng.reset_buf.clone().bool()
            eng.step((torch.rand(n, 9, generator=g) * 2 - 1).to(eng.device))
            tips_now = eng.state[capi.S_TIP_P:capi.S_TIP_P + 9].T.cpu().reshape(n, 3, 3)
            if t != 5:
                assert not was_flagged.any()
                continue
            assert was_flagged.all()                       # every env timed out at step 5 and was reset inside this step
            if term == "finger_move_penalty":
                pad = lambda x: torch.cat([x, torch.zeros(n, 3, 10)], dim=-1)   # noqa: E731
                want = rw.FingertipMovementPenalty(activate=True, weight=weight).compute(dt, pad(tips_now), pad(tips_before))
                assert torch.allclose(eng.reward.cpu(), want, rtol=1e-4, atol=1e-5)
                # and it IS a jump: the pre-reset tips are far from where the freshly reset fingers are
                assert float(eng.reward.abs().median()) > 5 * 0.1 * (0.01 / dt) ** 2
            else:
                obj_prev = eng.state[capi.S_PREV_OBJ_P:capi.S_PREV_OBJ_P + 3].T.cpu()       # the reset pose the physics started from
                obj_now = eng.state[capi.S_CUBE_P:capi.S_CUBE_P + 3].T.cpu()
                goal = eng.state[capi.S_GOAL_P:capi.S_GOAL_P + 3].T.cpu()
                pad13 = lambda p: torch.cat([p, torch.zeros(n, 10)], dim=-1)          # noqa: E731
                want = rw.ObjectMoveReward(activate=True, weight=weight).compute(pad13(obj_now), pad13(obj_prev),
                                                                                 torch.cat([goal, torch.zeros(n, 4)], dim=-1))
                assert torch.allclose(eng.reward.cpu(), want, rtol=1e-4, atol=2e-4)
                assert float(eng.reward.abs().max()) < 750 * 0.01     # one step of a resting cube: no jump
        eng.close()


def test_finger_reach_norm_p_values(backend):
    """FingerReachObjectRatePenalty takes any p of torch.norm (rewards.py:190-226); the native step builds the integer ones
    up to 16 and the maximum norm, and says so for the rest."""
    for p in (1, 2, 3, float("inf"), "inf"):
        env = make_env(backend, reward_terms={"finger_reach_object_rate": {"activate": True, "norm_p": p}})
        env.reset()
        _, rew, _, _ = env.step(torch.zeros(4, 9))
        assert torch.isfinite(rew).all()
        env.close()
    for p in (2.5, 0, 17):
        with pytest.raises(ValueError, match="norm_p"):
            make_env(backend, reward_terms={"finger_reach_object_rate": {"activate": True, "norm_p": p}})


def test_native_solver_option(backend):
    """`native.solver`: 'pgs' (default: 2 sub-steps x num_position_iterations sweeps) or 'tgs' (num_position_iterations
    sub-steps of one sweep, PhysX's temporal Gauss-Seidel that `sim.physx.solver_type = 1` asks for, env_base.py:62-63)."""
    a = make_env(backend, native={"solver": "tgs"}, sim={"physx": {"num_position_iterations": 8}})
    b = make_env(backend, sim={"physx": {"num_position_iterations": 8}})
    assert (a._engine.cfg.substeps, a._engine.cfg.solver_iterations) == (8, 1)
    assert (b._engine.cfg.substeps, b._engine.cfg.solver_iterations) == (2, 8)
    a.reset(), b.reset()
    for _ in range(30):
        a.step(torch.zeros(4, 9)), b.step(torch.zeros(4, 9))
    za, zb = a._engine.cube[2].cpu(), b._engine.cube[2].cpu()
    assert (za - 0.0325).abs().max() < 1e-3 and (zb - 0.0325).abs().max() < 1e-3       # the cube rests on the table either way
    assert torch.isfinite(a.obs_buf).all() and torch.isfinite(a._engine.state).all()     # (undriven fingers sag onto some cubes)
    with pytest.raises(ValueError, match="native.solver"):
        make_env(backend, native={"solver": "jacobi"})


def test_moving_goal_is_observed_before_it_advances(backend):
    """goal_movement.rotation: observations, rewards and termination of step t use the goal pose the step STARTED with;
    `__update_goal_movement_post` refreshes the pose buffer from the rotated goal actor only afterwards
    (trifinger_env.py:500-559, 1278-1284).  The goal actor turns by |w| dt per step about its angular velocity."""
    from leibnizgym_amd import _capi as capi
    env = make_env(backend, num_instances=64, task_difficulty=4, normalize_obs=False, episode_length=0,
                   goal_movement={"rotation": {"activate": True, "rate_magnitude": 0.5}})
    env.reset()
    eng = env._engine
    gq0 = eng.state[capi.S_GOAL_Q:capi.S_GOAL_Q + 4].T.clone()
    gw = eng.state[capi.S_GOAL_W:capi.S_GOAL_W + 3].T.clone()
    assert gw.norm(dim=1).min() > 1e-3                                  # every goal does rotate
    obs, _, _, _ = env.step(torch.zeros(64, 9))
    assert torch.equal(obs[:, 28:32], gq0)                              # observed: the pose the step started with
    gq1 = eng.state[capi.S_GOAL_Q:capi.S_GOAL_Q + 4].T.clone()
    chord = (gq1.double() - gq0.double()).norm(dim=1)                   # |q1 - q0| = 2 sin(angle / 4) for unit quaternions
    angle = 4.0 * torch.asin(0.5 * chord)
    assert torch.allclose(angle, gw.double().norm(dim=1) * 0.02, atol=2e-5, rtol=1e-3)      # advanced afterwards by |w| dt
    assert torch.allclose(gq1.norm(dim=1), torch.ones(64, device=gq1.device), atol=1e-6)
    obs2, _, _, _ = env.step(torch.zeros(64, 9))
    assert torch.equal(obs2[:, 28:32], gq1)
    assert torch.equal(env._object_goal_poses_buf[:, 3:7], eng.state[capi.S_GOAL_Q:capi.S_GOAL_Q + 4].T)   # buffer = refreshed pose


def test_fused_random_action_source(backend):
    """tf_step_random: the step with the reference demo driver's action source (2 * rand - 1, scripts/trifinger_random_action.py:33)
    fused in.  The draws are uniform in [-1, 1), differ between envs, dimensions and steps, do not depend on the shard layout, and
    the step is otherwise the ordinary one: stepping with the reported actions gives the same state bit for bit."""
    from leibnizgym_amd import _capi as capi
    from leibnizgym_amd.engine import TrifingerEngine, make_config
    from scipy import stats
    lib, dev = backend
    kw = dict(seed=4, command_mode="torque", asymmetric_obs=True, task_difficulty=4, success={"activate": False})
    n = 4096
    a = TrifingerEngine(make_config(lib, n, **kw), device=dev, lib=lib)
    b = TrifingerEngine(make_config(lib, n, **kw), device=dev, lib=lib)
    shard = TrifingerEngine(make_config(lib, 1024, env_id_offset=1024, global_num_envs=n, **kw), device=dev, lib=lib)
    for e in (a, b, shard):
        e.reset()
    prev = None
    for _ in range(3):
        a.step_random(), shard.step_random()
        act = a.action_buf.clone()
        b.step(act)                                                        # the ordinary step with the same actions
        assert torch.equal(a.state, b.state) and torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward)
        assert torch.equal(shard.action_buf, act[1024:2048]) and torch.equal(shard.state, a.state[:, 1024:2048])
        u = (act.cpu().numpy() + 1) / 2
        assert u.min() >= 0 and u.max() < 1 and stats.kstest(u.ravel(), "uniform").pvalue > 1e-3
        assert abs(np.corrcoef(u[:, 0], u[:, 5])[0, 1]) < 0.06 and abs(np.corrcoef(u[:-1, 3], u[1:, 3])[0, 1]) < 0.06
        assert prev is None or not np.array_equal(prev, u)
        prev = u
    for e in (a, b, shard):
        e.close()
    imp = TrifingerEngine(make_config(lib, 64, seed=1, command_mode="position_impedance"), device=dev, lib=lib)   # 18 dimensions
    imp.reset()
    imp.step_random()
    assert imp.action_buf.shape == (64, 18) and imp.action_buf.abs().max() <= 1 and imp.action_buf[:, 17].std() > 0.3
    imp.close()


def test_env_state_checkpoint_continues_bit_for_bit(backend, tmp_path):
    """a state_dict of the SoA buffers + counters; a rollout continued from it - in the same env or in a
    fresh one, through torch.save / torch.load - is the original rollout bit for bit (resets, goal resets, reward schedule, the
    observation noise and action repeats of the domain randomisation and the solver's warm start included)."""
    cfg = dict(num_instances=48, seed=5, episode_length=12, asymmetric_obs=True, task_difficulty=4,
               domain_randomization=dict(activate=True, obs_noise=0.002, action_repeat_prob=0.2))
    env = make_env(backend, **cfg)
    g = torch.Generator().manual_seed(1)
    acts = [(torch.rand(48, 9, generator=g) * 2 - 1).to(backend[1]) for _ in range(40)]
    env.reset()
    for a in acts[:15]:
        env.step(a)
    ck = env.state_dict()
    assert ck["frame_count"] == 16 and ck["state"].data_ptr() != env._engine.state.data_ptr()
    path = os.path.join(tmp_path, "env.pt")
    torch.save(ck, path)

    def run(e):
        out = []
        for a in acts[15:]:
            obs, rew, dones, info = e.step(a)
            out.append((obs.clone(), rew.clone(), e._reset_buf.clone(), e.states_buf.clone(), e._engine.state.clone(), dict(info)))
        return out
    first = run(env)
    assert any(bool(o[2].any()) for o in first), "the continuation must cross a time-out"
    env.load_state_dict(ck)                                              # back in time, same env
    fresh = make_env(backend, **cfg)                                     # and a new env that never ran the first 15 steps
    fresh.load_state_dict(torch.load(path, map_location=backend[1]))
    assert fresh.env_steps_count == env.env_steps_count == 16 * 48
    for other in (run(env), run(fresh)):
        for x, y in zip(first, other):
            for k in range(5):
                assert torch.equal(x[k].view(torch.uint8) if x[k].dtype == torch.bool else x[k].view(torch.int32), y[k].view(torch.uint8) if y[k].dtype == torch.bool else y[k].view(torch.int32))
            assert {k: float(v) for k, v in x[5].items()} == {k: float(v) for k, v in y[5].items()}
    # a checkpoint of another layout is refused
    small = make_env(backend, **dict(cfg, num_instances=16))
    with pytest.raises(ValueError, match="checkpoint of another engine"):
        small.load_state_dict(ck)
