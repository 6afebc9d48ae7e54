/*
 * urgym.h — C-ABI of the MI355X-native vectorised UR5e reach environment (liburgym_hip.so).
 *
 * The reference (WanqingXia/UR-gym) is pure Python on top of pybullet and has no FFI seam of its own
 * (SURVEY.md §8b).  The drop-in boundary is the Gymnasium Env surface of RobotTaskEnv, vectorised over N
 * environments; this header is what a ctypes binding of that surface calls.  Each entry point names the
 * reference interface it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - plain C, no torch types: the caller (PyTorch-ROCm tensors in ur_gym_amd/vector_env.py) allocates and
 *     owns EVERY buffer; the library borrows the raw device pointers registered with urgym_bind() and keeps
 *     only its constant tables (hull vertices, chain constants) uploaded at urgym_create().
 *   - state is float64 like the reference's internal state, struct-of-arrays [field][N] so that one lane per
 *     environment reads/writes coalesced; observations are float32 row-major [N][dim] exactly as
 *     RobotTaskEnv._get_obs casts them (UR_gym/envs/core.py:252-261).
 *   - all launches are asynchronous on the caller-supplied HIP stream (hipStream_t passed as void*).
 *   - return value: 0 = ok, <0 = error (urgym_last_error() gives the text).  No C++ exception crosses the ABI.
 *   - one handle per (process, device); calls on one handle are not re-entrant.
 */
#ifndef URGYM_H
#define URGYM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URGYM_ABI_VERSION 3

/* env kinds = the reference's registered ids (UR_gym/__init__.py:19-42, UR_gym/envs/ur_tasks.py:37-90) */
enum {
  URGYM_ENV_ORI = 0, /* UR5OriReach-v1 : ReachOri, reach.py:141-236 */
  URGYM_ENV_OBS = 1, /* UR5ObsReach-v1 : ReachObs, reach.py:239-374 */
  URGYM_ENV_DYN = 2, /* UR5DynReach-v1 : ReachDyn, reach.py:576-785 */
  URGYM_ENV_STA = 3, /* UR5StaReach-v1 : ReachSta, reach.py:377-573 (static obstacle; moves only when obst_end != 0) */
};

/* error codes */
enum {
  URGYM_OK = 0,
  URGYM_ERR_ARG = -1,
  URGYM_ERR_HIP = -2,
  URGYM_ERR_STATE = -3,
};

/* First separating axis of the GJK closest-point queries of the links (obstacle, table, track, self pairs).
 * BULLET: the world +Y axis, as btGjkPairDetector is entered by btConvexConvexAlgorithm -- the search then visits the
 *         same simplices as the reference's Bullet build (default; what every parity test pins).
 * GUIDED: the line from the other shape's centre to the mid point of the link's bounding capsule -- about half the
 *         iterations (a few % of step rate since the hull search starts from a direction map).  NOT parity-grade: Bullet's answer is not path-independent (its
 *         degenerate-simplex / no-progress exits return the current iterate), so on these finely faceted hulls the two
 *         modes differ by > 1e-6 m on ~1.5 % and > 1e-5 m on ~0.15 % of the queries, worst seen ~1e-4 m.  Opt-in for
 *         training runs that do not need the reference's exact numbers; see DESIGN.md "GJK start". */
enum {
  URGYM_GJK_START_BULLET = 0,
  URGYM_GJK_START_GUIDED = 1,
};

/* What task.link_dist (the five "link distance" slots of the Obs/Sta/Dyn observation and the distance-change reward) measures.
 * OBSTACLE : links 2..6 vs the obstacle -- PyBullet.get_link_distances as it stands (pyb_setup.py:439-456).  Default.
 * WORKBENCH: per link the minimum over obstacle, table and track -- what the docstring of get_link_distances still says
 *            ("the distance between workbench, obstacle and UR5") and what the reference's code evidently did when its
 *            UR5ObsReach-v1 / UR5StaReach-v1 checkpoints were trained (Sep 2023): the observations stored inside
 *            Trained_Models/Trained_{Obs,Sta}/best_model.zip are reproduced to 1e-7 by this rule and not by OBSTACLE
 *            (tests/test_reference_pins.py).  Needed to replay those two checkpoints (tests/test_closed_loop.py). */
enum {
  URGYM_LINK_DIST_OBSTACLE = 0,
  URGYM_LINK_DIST_WORKBENCH = 1,
};

/* bits of the per-env status word (device-side anomalies; SURVEY.md §5 "failure detection") */
enum {
  URGYM_STATUS_NAN = 1,              /* a NaN reached the reward/obs (utils.py:65-67 prints in the reference) */
  URGYM_STATUS_RESET_EXHAUSTED = 2,  /* rejection sampling hit max_reset_tries (reach.py:668-675 loops forever) */
  URGYM_STATUS_RESET_COLLISION = 4,  /* "Collision after reset, this should not happen" (reach.py:682-683) */
  URGYM_STATUS_PENETRATION = 8,      /* informational: a link_dist that was consumed is a penetration depth (negative) */
  URGYM_STATUS_GJK_ITER = 16,        /* GJK / EPA hit its iteration cap */
  URGYM_STATUS_STALE_RECORD = 64,    /* an env finished while the episode counter the caller had edited no longer matched its
                                        prefetched episode record, and urgym_invalidate_records had not been called: the env was
                                        NOT reset */
  URGYM_STATUS_JOINT_LIMIT = 32,     /* a joint was commanded past its URDF limit (ur5e.urdf:237-277: elbow +-pi, others +-2pi).
                                        The reference teleports joints with resetJointState, which does not clamp, but Bullet's
                                        limit constraints then act during stepSimulation: from here on the kinematic model of this
                                        build is outside the regime it was checked in. */
};

/* One POD config struct: every constant the reference hard-codes in its task constructors. */
typedef struct urgym_config {
  int32_t env_kind;          /* URGYM_ENV_* */
  int32_t num_envs;          /* N */
  int32_t max_episode_steps; /* TimeLimit, UR_gym/__init__.py:41 -> 100 */
  int32_t auto_reset;        /* 1: finished envs are reset inside urgym_step (gymnasium VectorEnv semantics) */
  int32_t check_collision;   /* 1: reference behaviour; 0: BASELINE.json configs[1] "FK + reward only" */
  int32_t max_reset_tries;   /* bound on the reference's unbounded rejection loop */
  int32_t dyn_motion_steps;  /* reach.py:735 -> 25 */
  int32_t gjk_start;         /* URGYM_GJK_START_*: first separating axis of every link query (default BULLET) */
  int32_t link_dist_scope;   /* URGYM_LINK_DIST_*: what task.link_dist measures (default OBSTACLE = the reference as it stands) */
  int32_t reserved0;         /* keeps the doubles 8-byte aligned; must be 0 */
  double action_scale;       /* UR5.py:276,314: pi*0.1 is applied as two float32 products; kept for reporting */
  double dt;                 /* pyb_setup.py:40,47-50: 20 substeps / 500 Hz = 0.04 s */
  double distance_threshold; /* reach.py:148/246/590 -> 0.05 */
  double ori_threshold;      /* reach.py:149/591 -> 0.0873 */
  double w_collision;        /* -500 */
  double w_success;          /* +200 */
  double w_distance;         /* Ori/Dyn -70, Obs -100 */
  double w_orientation;      /* Ori/Dyn -30, Obs 0 */
  double w_link[5];          /* Dyn: [8,2.4,1.2,1.2,0.2]/13*50 (reach.py:596-597); Obs: 100 each (reach.py:255,371) */
  double near_threshold;     /* 0.2 (reach.py:371,783) */
  double collision_margin;   /* 0.01 (pyb_setup.py:402,411,422) */
  double target_clearance;   /* 0.1 (reach.py:322,675) */
  double min_travel;         /* Dyn: 1.0 (reach.py:675) */
  double dyn_time_duration;  /* Dyn: 2.0 (reach.py:736) */
  double goal_low[3], goal_high[3]; /* reach.py:151-152 / 248-249 / 584-585 */
  double obst_low[3], obst_high[3]; /* reach.py:250-251 / 586-587 */
  double neutral_q[6];       /* UR5.py:262 */
} urgym_config;

/* Device pointers, all owned by the caller.  SoA state: X[f][n] at X[f*N + n]. */
typedef struct urgym_buffers {
  /* ---- per-env state (float64) ---- */
  double* q;          /* [6][N] joint angles (pybullet joint state, UR5.py:346-351) */
  double* goal;       /* [6][N] goal xyz + rpy (Obs uses rows 0..2) */
  double* obst_start; /* [6][N] obstacle start xyz+rpy  (Obs: the static obstacle; reach.py:263,582) */
  double* obst_end;   /* [6][N] obstacle end xyz+rpy    (Dyn; Sta: all-zero = static obstacle, reach.py:306) */
  double* obst_pos;   /* [3][N] current obstacle position (Bullet base position) */
  double* obst_quat;  /* [4][N] current obstacle orientation xyzw */
  double* obst_vel;   /* [9][N] rows 0..5: per-episode twist (v, omega) that set_velocity (reach.py:728-753) re-applies while
                         step_count < dyn_motion_steps; rows 6..8: the base displacement that twist produces in ONE env step
                         (20 Bullet sub-steps in which the linear velocity drifts by h * omega x v) --
                         derived at reset / refresh; a caller that writes a twist of its own into rows 0..5 calls
                         urgym_derive_obstacle_motion afterwards (urgym_refresh would recompute the twist from start / end) */
  double* link_dist;  /* [5][N] task.link_dist == task.last_dist (reach.py:680-681,780-782) */
  int32_t* step_count;/* [N] ReachDyn.step_num == TimeLimit._elapsed_steps */
  int32_t* episode_id;/* [N] number of resets so far (RNG counter) */
  /* ---- outputs ---- */
  float* observation;   /* [N][obs_dim]  */
  float* achieved_goal; /* [N][goal_dim] */
  float* desired_goal;  /* [N][goal_dim] */
  float* reward;        /* [N] */
  uint8_t* terminated;  /* [N] */
  uint8_t* truncated;   /* [N] */
  uint8_t* is_success;  /* [N] info["is_success"] (core.py:272,315) */
  uint8_t* collision;   /* [N] task.collision */
  /* terminal observation of envs that were auto-reset in this step (valid where terminated|truncated) */
  float* final_observation;   /* [N][obs_dim]  */
  float* final_achieved_goal; /* [N][goal_dim] */
  float* final_desired_goal;  /* [N][goal_dim] */
  int32_t* status;      /* [N] URGYM_STATUS_* bits, sticky until cleared by the caller */
  /* ---- scratch ---- */
  int32_t* done_list;   /* [N] compacted ids of envs to reset */
  int32_t* done_count;  /* [2] ping-pong counters */
} urgym_buffers;

/* ABI version of the loaded library (== URGYM_ABI_VERSION). */
int urgym_abi_version(void);

/* Fill cfg with the reference's constants for env_kind (reach.py constructors). */
int urgym_config_default(int env_kind, int num_envs, urgym_config* cfg);

/* Observation layout: obs_dim = 18|26|35|29, goal_dim = 6|3|6|6 (core.py:241-247; reach.py:189,307,653,453-457). */
int urgym_obs_dims(int env_kind, int* obs_dim, int* goal_dim);

/* Replaces UR5*ReachEnv.__init__ (ur_tasks.py:37-90): builds the constant scene/robot tables on `device` (hull neighbour
 * records, direction maps), allocates the library's own scratch (link distances and set-up cache of the running step, episode
 * records) and fixes the launch geometry for cfg->num_envs.  Environment variables read here, all optional and none of them
 * changes a result (scheduling / tuning / tests): URGYM_STEP_ENVS (envs per workgroup of the step kernel, 1..128),
 * URGYM_STEP_TIERS="E1,B,E2" (B workgroups of E1 envs, then workgroups of E2), URGYM_RESET_ENVS (envs per workgroup of the
 * auto-reset kernel, 1..64), URGYM_PREFETCH (0: reset finished envs with a kernel after each step instead of inline from
 * prefetched episode records), URGYM_SETUP_CACHE (0: every draw of a query recomputes the joint sines / cosines instead of
 * reading them from the per-env cache; 1 = default; 2: the cache also carries the link frames, +576 B per env),
 * URGYM_STEP_TIERS=0 (uniform workgroups where the default would be two-tier), URGYM_VERBOSE (print the chosen geometry to
 * stderr). */
int urgym_create(const urgym_config* cfg, int device, void** handle);
int urgym_destroy(void* handle);

/* Register the caller-owned device buffers (all pointers must stay valid until the next bind/destroy). */
int urgym_bind(void* handle, const urgym_buffers* bufs);

/* Replaces RobotTaskEnv.reset (core.py:263-273) for every env whose mask byte is non-zero (NULL = all).
 * `seed` re-keys the counter-based RNG; pass UINT64_MAX to keep the current key. */
int urgym_reset(void* handle, const uint8_t* mask_dev, uint64_t seed, void* stream);

/* Replaces RobotTaskEnv.step (core.py:303-317) + TimeLimit (UR_gym/__init__.py:41) for all N envs.
 * actions_dev: float32 [N][6] row-major, clipped to [-1,1] inside (UR5.py:274-275). */
int urgym_step(void* handle, const float* actions_dev, void* stream);

/* K consecutive urgym_step calls enqueued back to back: actions_dev is [K][N][6]. */
int urgym_rollout(void* handle, const float* actions_dev, int num_steps, void* stream);

/* Replaces Reach*.set_goal / set_goal_and_obstacle (reach.py:202-204, 328-335, 702-713): the caller has
 * overwritten goal / obst_start / obst_end (and possibly q) for the masked envs; this recomputes obstacle pose,
 * velocity, collision, link_dist and the observation for them, leaving step_count untouched. */
int urgym_refresh(void* handle, const uint8_t* mask_dev, void* stream);

/* The caller has edited episode_id (or step_count) of some envs in the bound buffers: the episode records the library keeps ready
 * for the inline auto-reset may no longer match, so the next max_episode_steps + 1 steps carry the fallback
 * launches that reset such envs with a kernel.  Every record and every pending refill entry is discarded (each env takes the
 * fallback at its first finish and leaves it with fresh records), after a device synchronisation: call it between steps, like the
 * edit itself.  Not needed after urgym_bind / urgym_reset (they do it themselves). */
int urgym_invalidate_records(void* handle);

/* The caller has written an obstacle twist of its own into rows 0..5 of obst_vel (a set_state-style harness; the reference's
 * counterpart is assigning task.velocity before sim.step, reach.py:745-747): re-derives rows 6..8 -- the base displacement of one
 * env step under that twist (pyb_setup.py:52-55, 20 sub-steps) -- for every env.  Leaves everything else alone (urgym_refresh would
 * teleport the obstacle to obst_start and recompute the twist from start / end).  No-op for UR5OriReach-v1. */
int urgym_derive_obstacle_motion(void* handle, void* stream);

/* Unit probe of the device closest-distance routine (what p.getClosestPoints computes, pyb_setup.py:401-452): one query per
 * entry, all pointers are DEVICE pointers.  type: 0 hull (par[0] = PyBullet link 1..6), 1 cylinder-Z (radius, height),
 * 2 box (half extents), 3 sphere (radius); pose = xyz + quaternion xyzw; out_dist = signed distance incl. Bullet margins.
 * Used by the parity tests to reach the hull<->box and hull<->hull paths directly. */
int urgym_probe_closest(void* handle, int count, const int* type_a, const double* par_a, const double* pose_a, const int* type_b,
                        const double* par_b, const double* pose_b, double threshold, double* out_dist, int* out_info, void* stream);

/* Unit probe of the device pose distances (utils.distance, utils.py:5-31; utils.angular_distance, utils.py:34-69 -- what
 * is_success and compute_reward call, reach.py:212-236): a6 / b6 are [count][6] float64 poses xyz + rpy (DEVICE pointers),
 * out2[count][2] = {distance, angular distance}.  Lets the fixtures generated by the reference's own utils.py reach the
 * device code directly (tests/test_gpu_parity.py). */
int urgym_probe_pose_distance(void* handle, int count, const double* a6, const double* b6, double* out2, void* stream);

/* Average device time (microseconds) of the step launch over the calls since the last query, measured with hipEvents on the
 * launch stream; returns <0 if timing was not enabled.  enable = k > 1 times every k-th step only: a pair of events costs the
 * stream about 6 us per step, i.e. measuring every launch slows the thing measured by ~3 % (bench.py samples every 8th). */
int urgym_enable_timing(void* handle, int enable);
int urgym_query_timing(void* handle, double* step_kernel_us, double* reset_kernel_us, int* launches);

/* Kept for ABI v2 callers: always reports 0.  The search for the next episodes of the envs that finished used to run as a launch
 * of its own on a side stream; it now rides in the step launch (its workgroups follow the step workgroups in one grid), so
 * urgym_query_timing()'s step figure includes it. */
int urgym_query_refill_timing(void* handle, double* refill_us);

const char* urgym_last_error(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* URGYM_H */
