"""ctypes mirror of include/urgym.h (struct layouts and constants only; no library is loaded here)."""
import ctypes as C

ABI_VERSION = 3

ENV_ORI, ENV_OBS, ENV_DYN, ENV_STA = 0, 1, 2, 3
ENV_IDS = {"UR5OriReach-v1": ENV_ORI, "UR5ObsReach-v1": ENV_OBS, "UR5DynReach-v1": ENV_DYN, "UR5StaReach-v1": ENV_STA}

OK, ERR_ARG, ERR_HIP, ERR_STATE = 0, -1, -2, -3

STATUS_NAN = 1
STATUS_RESET_EXHAUSTED = 2
STATUS_RESET_COLLISION = 4
STATUS_PENETRATION = 8
STATUS_GJK_ITER = 16
STATUS_JOINT_LIMIT = 32
STATUS_STALE_RECORD = 64

OBS_DIMS = {ENV_ORI: (18, 6), ENV_OBS: (26, 3), ENV_DYN: (35, 6), ENV_STA: (29, 6)}  # (observation, goal) — core.py:241-247

GJK_START_BULLET, GJK_START_GUIDED = 0, 1  # urgym_config.gjk_start (include/urgym.h)
LINK_DIST_OBSTACLE, LINK_DIST_WORKBENCH = 0, 1  # urgym_config.link_dist_scope (include/urgym.h)
KEEP_SEED = 0xFFFFFFFFFFFFFFFF


class Config(C.Structure):
    _fields_ = [
        ("env_kind", C.c_int32),
        ("num_envs", C.c_int32),
        ("max_episode_steps", C.c_int32),
        ("auto_reset", C.c_int32),
        ("check_collision", C.c_int32),
        ("max_reset_tries", C.c_int32),
        ("dyn_motion_steps", C.c_int32),
        ("gjk_start", C.c_int32),
        ("link_dist_scope", C.c_int32),
        ("reserved0", C.c_int32),
        ("action_scale", C.c_double),
        ("dt", C.c_double),
        ("distance_threshold", C.c_double),
        ("ori_threshold", C.c_double),
        ("w_collision", C.c_double),
        ("w_success", C.c_double),
        ("w_distance", C.c_double),
        ("w_orientation", C.c_double),
        ("w_link", C.c_double * 5),
        ("near_threshold", C.c_double),
        ("collision_margin", C.c_double),
        ("target_clearance", C.c_double),
        ("min_travel", C.c_double),
        ("dyn_time_duration", C.c_double),
        ("goal_low", C.c_double * 3),
        ("goal_high", C.c_double * 3),
        ("obst_low", C.c_double * 3),
        ("obst_high", C.c_double * 3),
        ("neutral_q", C.c_double * 6),
    ]


# name -> (ctype of element, leading shape given (N, obs_dim, goal_dim), is_state)
BUFFER_FIELDS = [
    ("q", C.c_double, lambda N, od, gd: (6, N)),
    ("goal", C.c_double, lambda N, od, gd: (6, N)),
    ("obst_start", C.c_double, lambda N, od, gd: (6, N)),
    ("obst_end", C.c_double, lambda N, od, gd: (6, N)),
    ("obst_pos", C.c_double, lambda N, od, gd: (3, N)),
    ("obst_quat", C.c_double, lambda N, od, gd: (4, N)),
    ("obst_vel", C.c_double, lambda N, od, gd: (9, N)),
    ("link_dist", C.c_double, lambda N, od, gd: (5, N)),
    ("step_count", C.c_int32, lambda N, od, gd: (N,)),
    ("episode_id", C.c_int32, lambda N, od, gd: (N,)),
    ("observation", C.c_float, lambda N, od, gd: (N, od)),
    ("achieved_goal", C.c_float, lambda N, od, gd: (N, gd)),
    ("desired_goal", C.c_float, lambda N, od, gd: (N, gd)),
    ("reward", C.c_float, lambda N, od, gd: (N,)),
    ("terminated", C.c_uint8, lambda N, od, gd: (N,)),
    ("truncated", C.c_uint8, lambda N, od, gd: (N,)),
    ("is_success", C.c_uint8, lambda N, od, gd: (N,)),
    ("collision", C.c_uint8, lambda N, od, gd: (N,)),
    ("final_observation", C.c_float, lambda N, od, gd: (N, od)),
    ("final_achieved_goal", C.c_float, lambda N, od, gd: (N, gd)),
    ("final_desired_goal", C.c_float, lambda N, od, gd: (N, gd)),
    ("status", C.c_int32, lambda N, od, gd: (N,)),
    ("done_list", C.c_int32, lambda N, od, gd: (N,)),
    ("done_count", C.c_int32, lambda N, od, gd: (2,)),
]


class Buffers(C.Structure):
    _fields_ = [(name, C.POINTER(ct)) for name, ct, _ in BUFFER_FIELDS]


# Every symbol include/urgym.h declares (tests check that the built library exports each of them).
EXPORTED_SYMBOLS = [
    "urgym_abi_version",
    "urgym_config_default",
    "urgym_obs_dims",
    "urgym_create",
    "urgym_destroy",
    "urgym_bind",
    "urgym_reset",
    "urgym_step",
    "urgym_rollout",
    "urgym_refresh",
    "urgym_invalidate_records",
    "urgym_derive_obstacle_motion",
    "urgym_probe_closest",
    "urgym_probe_pose_distance",
    "urgym_enable_timing",
    "urgym_query_timing", "urgym_query_refill_timing",
    "urgym_last_error",
]
