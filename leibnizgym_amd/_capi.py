"""ctypes binding of the C ABI declared in include/trifinger.h.

This is the stub a maintainer of the reference would add to reach the native step from
Python (INTEGRATION.md shows it next to the `gymapi` calls it replaces).  It knows nothing
about torch: buffers cross the boundary as integer device addresses.

The product loads exactly one library through it: ``leibnizgym_amd/csrc/libtrifinger_hip.so``
(see :func:`load_hip_library`).  The class takes a path so that the test-suite can point the
same binding at the CPU oracle; nothing in this package ever does.
"""
import ctypes as C
import os

TF_API_VERSION = 8
TF_NUM_REWARD_TERMS = 6
TF_NUM_INFO = 16
TF_STATE_ROWS = 172
TF_NUM_DR = 14
DR_BASE_POS, DR_STAGE_POS, DR_FRICTION_ROBOT, DR_FRICTION_OBJECT, DR_FRICTION_STAGE = 6, 9, 11, 12, 13

# status codes (include/trifinger.h: TfStatus)
TF_OK = 0
TF_ERR_INVALID_ARG = -1
TF_ERR_COMMAND_MODE = -2
TF_ERR_ROBOT_RESET = -3
TF_ERR_OBJECT_RESET = -4
TF_ERR_DIFFICULTY = -5
TF_ERR_NOT_BOUND = -6
TF_ERR_DEVICE = -7
TF_ERR_UNSUPPORTED = -8

COMMAND_MODES = {"torque": 0, "position": 1, "position_impedance": 2}
RESET_TYPES = {"none": 0, "default": 1, "random": 2}
REWARD_TERM_ORDER = (
    "finger_reach_object_rate",
    "finger_move_penalty",
    "object_dist",
    "object_rot",
    "object_rot_delta",
    "object_move",
)

# rows of the SoA state matrix (include/trifinger.h)
S_Q, S_QD, S_CUBE_P, S_CUBE_Q, S_CUBE_V, S_CUBE_W = 0, 9, 18, 21, 25, 28
S_GOAL_P, S_GOAL_Q, S_GOAL_W, S_TIP_P, S_TAU = 31, 34, 38, 41, 50
S_PREV_OBJ_P, S_PREV_OBJ_Q, S_FT, S_DR = 59, 62, 66, 84
# warm-start rows of the contact solver
S_LAM_FC, S_FC_LINK, S_LAM_TF, S_LAM_TW, S_LAM_CF, S_CF_FACE, S_LAM_CW, S_CW_FACE = 98, 110, 113, 122, 131, 143, 144, 156
S_NEXT_OBJ, S_NEXT_GOAL, S_NEXT_TAG = 157, 161, 171

INFO_POS_COUNT, INFO_ORI_COUNT, INFO_SUCCESS_MEAN, INFO_NUM_RESETS, INFO_NUM_NONFINITE = 6, 7, 8, 9, 10


class TfRewardTerm(C.Structure):
    _fields_ = [("activate", C.c_int32), ("weight", C.c_float),
                ("sched_start", C.c_double), ("sched_end", C.c_double)]


class TfLinkShape(C.Structure):
    _fields_ = [("a", C.c_float * 3), ("b", C.c_float * 3), ("w1", C.c_float * 2), ("w2", C.c_float * 2), ("rho", C.c_float * 2),
                ("o1", C.c_float * 2), ("o2", C.c_float * 2)]


class TfSphere(C.Structure):
    _fields_ = [("c", C.c_float * 3), ("radius", C.c_float)]


class TfModel(C.Structure):
    _fields_ = [
        ("base_height", C.c_float),
        ("base_yaw_cos", C.c_float * 3), ("base_yaw_sin", C.c_float * 3),
        ("base_half_yaw_cos", C.c_float * 3), ("base_half_yaw_sin", C.c_float * 3),
        ("j2_origin", C.c_float * 3), ("j3_origin", C.c_float * 3), ("tip_origin", C.c_float * 3),
        ("link_mass", C.c_float * 3), ("link_com", (C.c_float * 3) * 3), ("link_inertia", (C.c_float * 6) * 3),
        ("q_lo", C.c_float * 3), ("q_hi", C.c_float * 3),
        ("qd_max", C.c_float), ("tau_max", C.c_float), ("link_angular_damping", C.c_float),
        ("q_default", C.c_float * 3),
        ("cap_a", C.c_float * 3), ("cap_b", C.c_float * 3), ("cap_radius", C.c_float),
        ("shape3", TfLinkShape), ("sph3", TfSphere * 1), ("shape2", TfLinkShape), ("sph2", TfSphere * 2), ("shape1", TfLinkShape),
        ("upper_check_z", C.c_float), ("middle_check_z", C.c_float),
        ("cube_half", C.c_float), ("cube_mass", C.c_float), ("cube_inertia", C.c_float),
        ("cube_linear_damping", C.c_float), ("cube_angular_damping", C.c_float),
        ("wall_r", C.c_float * 4), ("wall_z", C.c_float * 4),
        ("mu_finger_cube", C.c_float), ("mu_cube_floor", C.c_float),
        ("mu_tip_floor", C.c_float), ("mu_cube_wall", C.c_float),
        ("mu_tip_wall", C.c_float), ("mu_finger_finger", C.c_float),
        ("mu_robot", C.c_float), ("mu_object", C.c_float), ("mu_floor", C.c_float), ("mu_stage", C.c_float),
        ("restitution_finger", C.c_float), ("restitution_ff", C.c_float), ("bounce_threshold", C.c_float),
        ("contact_margin", C.c_float), ("contact_slack", C.c_float), ("contact_offset", C.c_float), ("erp", C.c_float),
        ("max_depenetration_velocity", C.c_float), ("warm_start", C.c_float),
        ("box", C.c_int32), ("box_gyroscopic", C.c_int32), ("box_half", C.c_float * 3), ("box_inertia", C.c_float * 3),
        ("obj_radius_3d", C.c_float), ("obj_max_com_dist", C.c_float), ("obj_min_height", C.c_float),
        ("obj_span_min_height", C.c_float), ("obj_span_radius", C.c_float),
        ("ff_middle_pairs", C.c_int32),
    ]


class TfConfig(C.Structure):
    _fields_ = [
        ("api_version", C.c_int32), ("num_envs", C.c_int32), ("env_id_offset", C.c_int32),
        ("global_num_envs", C.c_int32), ("seed", C.c_uint64),
        ("command_mode", C.c_int32),
        ("normalize_action", C.c_int32), ("normalize_obs", C.c_int32), ("apply_safety_damping", C.c_int32),
        ("asymmetric_obs", C.c_int32), ("enable_ft_sensors", C.c_int32),
        ("task_difficulty", C.c_int32), ("episode_length", C.c_int32), ("control_decimation", C.c_int32),
        ("robot_reset_type", C.c_int32), ("dof_pos_stddev", C.c_float), ("dof_vel_stddev", C.c_float),
        ("object_reset_type", C.c_int32),
        ("goal_rotation_activate", C.c_int32), ("goal_rotation_rate_magnitude", C.c_float),
        ("reward", TfRewardTerm * TF_NUM_REWARD_TERMS),
        ("finger_reach_norm_p", C.c_int32), ("object_rot_scale", C.c_float),
        ("success_activate", C.c_int32), ("success_bonus", C.c_float),
        ("position_tolerance", C.c_float), ("orientation_tolerance", C.c_float),
        ("dt", C.c_float), ("substeps", C.c_int32), ("solver_iterations", C.c_int32), ("solver_inner", C.c_int32),
        ("gravity", C.c_float * 3),
        ("dr_enable", C.c_int32),
        ("dr_cube_mass", C.c_float * 2), ("dr_cube_size", C.c_float * 2),
        ("dr_friction", C.c_float * 2), ("dr_motor", C.c_float * 2),
        ("dr_link_mass", C.c_float * 2), ("dr_restitution", C.c_float * 2),
        ("dr_base_pos", C.c_float * 3), ("dr_stage_pos", C.c_float * 2), ("dr_friction_robot", C.c_float * 2),
        ("dr_friction_object", C.c_float * 2), ("dr_friction_stage", C.c_float * 2),
        ("dr_obs_noise", C.c_float), ("dr_action_repeat", C.c_float),
        ("model", TfModel),
    ]


class TfBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "state", "action_buf", "obs", "states", "reward", "reset_buf", "goal_reset_buf", "successes",
        "dones", "steps", "reset_count", "info", "scratch")]


_P = C.c_void_p

# every symbol include/trifinger.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tf_api_version": (C.c_int, []),
    "tf_backend_name": (C.c_char_p, []),
    "tf_last_error_string": (C.c_char_p, []),
    "tf_default_model": (None, [C.POINTER(TfModel)]),
    "tf_model_set_box": (None, [C.POINTER(TfModel), C.POINTER(C.c_float), C.c_float]),
    "tf_action_dim": (C.c_int, [C.c_int32]),
    "tf_scratch_floats": (C.c_int64, [C.c_int32]),
    "tf_create": (C.c_int, [C.POINTER(TfConfig), C.POINTER(_P)]),
    "tf_destroy": (C.c_int, [_P]),
    "tf_bind": (C.c_int, [_P, C.POINTER(TfBuffers)]),
    "tf_set_gravity": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "tf_set_clipping": (C.c_int, [_P, C.c_float, C.c_float]),
    "tf_frame_count": (C.c_int64, [_P]),
    "tf_set_frame_count": (C.c_int, [_P, C.c_int64]),
    "tf_set_kernel_variant": (C.c_int, [_P, C.c_int32]),
    "tf_kernel_variant": (C.c_int, [_P]),
    "tf_kernel_occupancy": (C.c_int, [_P]),
    "tf_step": (C.c_int, [_P, _P, _P]),
    "tf_step_random": (C.c_int, [_P, C.c_void_p]),
    "tf_reset": (C.c_int, [_P, _P]),
    "tf_enable_kernel_timing": (C.c_int, [_P, C.c_int32]),
    "tf_set_kernel_timing_window": (C.c_int, [_P, C.c_int32]),
    "tf_kernel_time_ms": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "tf_apply_resets": (C.c_int, [_P, _P]),
    "tf_pre_step": (C.c_int, [_P, _P]),
    "tf_simulate": (C.c_int, [_P, _P]),
    "tf_post_step": (C.c_int, [_P, _P]),
    "tf_finish_step": (C.c_int, [_P, _P]),
    "tf_test_quat_diff_rad": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "tf_test_quat_mul": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "tf_test_lgsk": (C.c_int, [_P, C.c_float, _P, C.c_int32, _P]),
    "tf_test_sample_xy": (C.c_int, [_P, _P, C.c_float, _P, _P, C.c_int32, _P]),
    "tf_test_sample_yaw_quat": (C.c_int, [_P, _P, C.c_int32, _P]),
    "tf_test_normalize_quat": (C.c_int, [_P, _P, C.c_int32, _P]),
    "tf_test_philox": (C.c_int, [C.c_uint64, _P, _P, C.c_uint32, _P, C.c_int32, _P]),
    "tf_test_finger_dynamics": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, _P]),
}


class TfLibraryError(RuntimeError):
    pass


class TfLib:
    """A loaded shared library exporting the trifinger C ABI."""

    def __init__(self, path):
        if not os.path.isfile(path):
            raise TfLibraryError(
                f"native library not found: {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C leibnizgym_amd/csrc`). There is no fallback path.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(self.dll, name)
            except AttributeError as exc:
                raise TfLibraryError(f"{path} does not export `{name}` declared in include/trifinger.h") from exc
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        ver = self.tf_api_version()
        if ver != TF_API_VERSION:
            raise TfLibraryError(f"{path}: API version {ver} != {TF_API_VERSION}")

    @property
    def backend(self):
        return self.tf_backend_name().decode()

    def default_model(self):
        m = TfModel()
        self.tf_default_model(C.byref(m))
        return m

    def box_model(self, size, density):
        """default model with the object replaced by a box of `size` (x, y, z in metres) and `density` (kg / m^3)"""
        m = self.default_model()
        self.tf_model_set_box(C.byref(m), (C.c_float * 3)(*[float(x) for x in size]), float(density))
        return m

    def last_error(self):
        return (self.tf_last_error_string() or b"").decode()


def hip_library_path():
    """the in-tree product library; nothing (no environment variable, no search path) redirects it"""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtrifinger_hip.so")


_HIP_LIB = None


def load_hip_library():
    """Load the HIP product library (cached).  Fails loudly when it has not been built."""
    global _HIP_LIB
    if _HIP_LIB is None:
        lib = TfLib(hip_library_path())
        if lib.backend != "hip-gfx950":       # the product path runs the HIP kernels or nothing
            raise TfLibraryError(f"{lib.path} reports backend `{lib.backend}`, not the gfx950 HIP build of csrc/trifinger_hip.hip")
        _HIP_LIB = lib
    return _HIP_LIB
