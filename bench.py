#!/usr/bin/env python3
"""bench.py — env-steps/sec of the fused UR5e reach step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one call of VectorEnv.step() (fused step kernel + device-side auto-reset of finished envs) over one
batch of synthetic actions for every environment of this rank.  Workload at N=1: BASELINE.json configs[3],
UR5DynReach-v1 with 65536 environments on one MI355X (the configuration the metric is quoted on).  For N>1 there is
one rank per GPU: either the driver launches them through torch.distributed.run, or -- when this file is started as
plain `python bench.py --gpus N` -- it starts that launcher itself as a CHILD process (the parent never touches the GPU)
and relays rank 0's JSON line and the exit code.  Environments shard across ranks with no data-path collective (weak
scaling: 65536 envs per GPU); the barrier + max-over-ranks timing uses RCCL; `--gather-obs` adds the one optional
exchange of the path, an all-gather of the observations (12.3 MB per GPU per step at 65536 envs).

Rank 0 prints ONE JSON line.  `roofline.achieved` prices the dominant kernel (env_kernel<Dyn, STEP>) with the
algorithmic bytes of SURVEY.md §8(d) (418 B per env-step) against the HBM peak; the kernel's duration is measured
live with HIP events on the launch stream (urgym_enable_timing).  `cpu_baseline` times the CPU oracle
(oracle/, kind "port": the reference's PyBullet path cannot run here) on rank 0's host cores over a bounded
sample of the same workload, and -- SURVEY.md section 8d "outputs compared at the end of the timed loop" -- steps a slice
of the GPU's final state once on both sides and reports the largest deviations (`parity_check`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"UR5OriReach-v1": 230, "UR5ObsReach-v1": 290, "UR5DynReach-v1": 418, "UR5StaReach-v1": 370}  # SURVEY.md §8(d); Sta = Dyn without the velocity slots
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r2", "pmc_summary.json")  # rocprofv3 --pmc passes of this same command (tools/gpu_round.sh prof)


def profiled_counters(env_id, n):
    """Counters of the step kernel from the committed rocprofv3 PMC summary; only valid for the configuration that was
    profiled (Dyn, N=65536).  traffic = HBM bytes per launch (FETCH_SIZE corrected x2 as the gfx950 guide prescribes + WRITE_SIZE);
    valu_* = the VALU-issue picture of the same launches (SQ_ACTIVE_INST_VALU in quad-cycles over the SIMD-cycles of the kernel)."""
    if env_id != "UR5DynReach-v1" or n != 65536 or not os.path.exists(PMC_SUMMARY):
        return None, None
    try:
        k = json.load(open(PMC_SUMMARY))["env_kernel<2, 0>"]
        traffic = float(k["hbm_read_bytes_per_launch_corrected"] + k["hbm_write_bytes_per_launch"])
        valu = {"busy_fraction_of_simd_cycles": k.get("valu_busy_fraction_of_simd_cycles"), "lane_utilisation": k.get("valu_lane_utilisation"),
                "wave_slot_occupancy": k.get("wave_slot_occupancy"), "wait_fraction_of_wave_cycles": k.get("wait_fraction_of_wave_cycles"),
                "valu_wave_instructions_per_launch": k.get("SQ_INSTS_VALU"), "scratch_bytes_per_lane": k.get("Scratch_Size"),
                "source": "profiles/r2/pmc_summary.json (static: separate rocprofv3 --pmc passes of this command, not measured in this run)"}
        return traffic, valu
    except Exception:
        return None, None


def cpu_baseline(env_id, seed, budget_s=12.0):
    """Oracle (CPU restatement) on all host cores of this process, same workload shape, bounded sample."""
    from oracle import binding as ob
    from ur_gym_amd import _abi

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kind = _abi.ENV_IDS[env_id]
    n = max(2048, 64 * cores)  # enough envs per host thread that the thread pool is not the thing measured
    env = ob.OracleEnv(kind, n, threads=cores)
    env.reset(seed=seed)
    rng = np.random.default_rng(seed)
    acts = rng.uniform(-1, 1, (8, n, 6)).astype(np.float32)
    # calibrate on 2 steps, then run as many steps as fit in the budget (at least 4)
    t0 = time.perf_counter()
    for k in range(2):
        env.step(acts[k])
    per_step = (time.perf_counter() - t0) / 2
    steps = int(max(4, min(400, budget_s / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % len(acts)])
    dt = time.perf_counter() - t0
    env.close()
    # the same restatement on ONE host thread, ~2 s
    n1 = 256
    env1 = ob.OracleEnv(kind, n1, threads=1)
    env1.reset(seed=seed)
    t1 = time.perf_counter()
    s1 = 0
    while time.perf_counter() - t1 < 2.0:
        env1.step(acts[s1 % len(acts)][:n1])
        s1 += 1
    d1 = time.perf_counter() - t1
    env1.close()
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {steps} steps of {env_id}, random actions, auto-reset on ({dt:.1f} s)",
            "single_thread_value": n1 * s1 / d1, "single_thread_sample": f"{n1} envs x {s1} steps ({d1:.1f} s)"}



STATUS_BITS = {1: "nan", 2: "reset_exhausted", 4: "reset_collision", 8: "penetration_depth_consumed", 16: "gjk_or_epa_iteration_cap",
               32: "joint_limit_passed", 64: "stale_episode_record"}


def parity_check(env, actions, m=512):
    """After the timed loop: the first m envs of the GPU's current state are stepped once more on the GPU and, from the very
    same state and actions, by the CPU oracle; every output is compared (the checker leg of cpu_baseline, never timed)."""
    from oracle import binding as ob
    from ur_gym_amd import _abi

    m = min(m, env.num_envs)
    state = {k: v[..., :m].copy() for k, v in env.get_state().items()}
    orc = ob.OracleEnv(env.env_kind, m, threads=min(16, os.cpu_count() or 1), auto_reset=0, check_collision=int(env.cfg.check_collision),
                       gjk_start=int(env.cfg.gjk_start), link_dist_scope=int(env.cfg.link_dist_scope))
    orc.load_state(state)
    orc.buf["observation"][...] = env.buf["observation"][:m].cpu().numpy()  # (carries the stale velocity slot of the Dyn observation)
    env.step(actions)
    torch.cuda.synchronize(env.device)
    orc.step(actions[:m].cpu().numpy())
    done = (orc.buf["terminated"] | orc.buf["truncated"]).astype(bool)
    # finished envs were auto-reset on the GPU: their terminal observation is what the oracle (auto-reset off) still shows
    obs_gpu = env.buf["observation"][:m].cpu().numpy().astype(np.float64)
    fin_gpu = env.buf["final_observation"][:m].cpu().numpy().astype(np.float64)
    obs_gpu[done] = fin_gpu[done]
    d = np.abs(obs_gpu - orc.buf["observation"])
    euler = [3, 4, 5] + ([21, 22, 23] if env.env_kind in (_abi.ENV_DYN, _abi.ENV_STA) else [])
    d[:, euler] = np.minimum(d[:, euler], np.abs(d[:, euler] - 2 * np.pi))  # atan2 branch cut at +-pi
    flags_equal = all(np.array_equal(env.buf[k][:m].cpu().numpy(), orc.buf[k]) for k in ("terminated", "truncated", "is_success", "collision"))
    out = {"envs": m, "max_abs_dev_observation": float(d.max()),
           "max_abs_dev_reward": float(np.abs(env.buf["reward"][:m].cpu().numpy().astype(np.float64) - orc.buf["reward"]).max()),
           "flags_equal": bool(flags_equal), "finished_in_slice": int(done.sum()), "tolerance": 1e-4}
    orc.close()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 outside a launcher: this process has not touched the GPU (torch is only imported);
    it starts one rank per GPU through torch.distributed.run as a child, relays rank 0's JSON line and the exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child_env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, env=child_env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    rc = p.returncode if p.returncode != 0 else (0 if lines else 1)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--env", default="UR5DynReach-v1")
    ap.add_argument("--num-envs", type=int, default=65536, help="environments PER GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gather-obs", action="store_true", help="all-gather observations over RCCL every step (optional)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gjk-start", choices=("bullet", "guided"), default="bullet",
                    help="bullet = the reference's search start (parity-grade, default); guided = opt-in, NOT parity-grade")
    ap.add_argument("--no-collision", action="store_true",
                    help="check_collision=0: BASELINE.json configs[1] 'FK + pose-distance reward kernel only' (not the reference's step)")
    ap.add_argument("--rollout", action="store_true", help="enqueue all K steps through urgym_rollout (no Python per step)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: no HIP events around the kernels (roofline.achieved is then 0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        spawn_ranks(args)  # does not return
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} does not match --gpus {args.gpus}")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # URGYM_BENCH_REHEARSE=1: rehearsal of the N>1 control flow on a ONE-GPU box (gloo, every rank on cuda:0); never a result
        rehearse = os.environ.get("URGYM_BENCH_REHEARSE") == "1"
        if rehearse:
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from ur_gym_amd import make_vec

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    n = args.num_envs
    env = make_vec(args.env, num_envs=n, device=dev, seed=args.seed + 1000 * rank, gjk_start=int(args.gjk_start == "guided"),
                   check_collision=not args.no_collision)
    env.reset(seed=args.seed + 1000 * rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + rank)
    # distinct action batches, cycled (random policy as demo.py:11).  64 is within 1 % of never repeating a batch (measured: 206.2 M
    # env-steps/s with 64, 208.8 M with 256 or 1000, N = 65536); a SHORT cycle is a different workload -- with 16 the joints drift,
    # the arms run into contact and 50 % more episodes end per step (170 M).  URGYM_BENCH_NACT overrides (diagnostic).
    n_act = min(args.steps + args.warmup, int(os.environ.get("URGYM_BENCH_NACT", "64")))
    actions = torch.rand((n_act, n, 6), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
    gathered = None
    gloo = dist is not None and dist.get_backend() == "gloo"
    if args.gather_obs and world > 1:
        # the optional exchange of the path: every rank ends up with all observations (RCCL all-gather over xGMI; in the
        # one-GPU rehearsal gloo gathers through a pinned host staging buffer)
        gathered = torch.empty((world * n, env.obs_dim), dtype=torch.float32, device="cpu" if gloo else dev)
        stage = torch.empty((n, env.obs_dim), dtype=torch.float32, pin_memory=True) if gloo else None

    def one_step(k):
        env.step(actions[k % n_act])
        if gathered is not None:
            if gloo:
                stage.copy_(env.buf["observation"])  # (synchronises with the step on the current stream)
                dist.all_gather(list(gathered.view(world, n, env.obs_dim).unbind(0)), stage)
            else:
                dist.all_gather_into_tensor(gathered, env.buf["observation"])

    for k in range(args.warmup):
        one_step(k)
    rollout_actions = None
    if args.rollout and gathered is None:  # resident before the timed region, like the per-step action batches
        rollout_actions = torch.stack([actions[(args.warmup + k) % n_act] for k in range(args.steps)])
    torch.cuda.synchronize(dev)
    env.enable_timing(not args.no_kernel_timing, every=8)  # HIP events around every 8th step launch (an event pair costs the stream ~6 us)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if rollout_actions is not None:
        env.rollout(rollout_actions)
    else:
        for k in range(args.steps):
            one_step(args.warmup + k)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    step_us, reset_us, launches = (0.0, 0.0, 0) if args.no_kernel_timing else env.query_timing()
    env.enable_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    status = env.buf["status"]
    anomalies = int((status != 0).sum().item())
    anomaly_bits = {name: int(((status & bit) != 0).sum().item()) for bit, name in STATUS_BITS.items()}
    episodes = int(env.buf["episode_id"].sum().item())
    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        algo = ALGO_BYTES[args.env] * n  # bytes one launch of the step kernel has to move, per rank
        traffic, valu = profiled_counters(args.env, n)
        achieved = algo / (step_us * 1e-6) / 1e9 if step_us > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node)",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.env} N={n} per GPU, random actions U(-1,1), auto-reset, seed {args.seed}"
                                   + (", collision checks OFF (FK + reward only)" if args.no_collision else ""),
                       "envs_total": total_envs, "gather_obs": bool(gathered is not None), "rollout_api": bool(args.rollout),
                       "gjk_start": args.gjk_start},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": "profiles/r2/pmc_summary.json (bytes per launch, separate rocprofv3 --pmc passes)",
                         "valu_issue": valu,  # what really bounds the kernel: float64 VALU issue + the latency of the GJK chains
                         "algorithmic_bytes_per_launch": algo,
                         "kernel": "env_step_fused<Dyn> (STEP workgroups + the refill of the previous step's episode records)" if args.env == "UR5DynReach-v1" else "step launch",
                         "kernel_us": step_us, "reset_kernel_us": reset_us, "launches_timed": launches,
                         "algorithmic_bytes_per_env_step": ALGO_BYTES[args.env],
                         "note": "bound by the dependent float64 chain of the GJK iterations (resident waves, then VALU issue), not by HBM (DESIGN.md section 4)"},
            "anomalous_envs": anomalies,
            "anomalous_envs_by_status_bit": anomaly_bits,  # informational bits included (include/urgym.h URGYM_STATUS_*)
            "episodes_started": episodes,
        }
        if gathered is not None:
            out["config"]["gather_bytes_per_gpu_per_step"] = n * env.obs_dim * 4
            ok = bool(torch.equal(gathered.view(world, n, env.obs_dim)[rank].to(dev), env.buf["observation"]))
            out["config"]["gather_own_shard_intact"] = ok
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.env, args.seed)
            out["cpu_baseline"]["parity_check"] = parity_check(env, actions[(args.warmup + args.steps) % n_act])
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
