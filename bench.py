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
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r3", "pmc_summary.json")  # rocprofv3 passes of this same command per config (tools/r3_prof.sh)


def config_label(env_id, n, no_collision=False, rollout=False):
    """Key of a profiled configuration in profiles/r3/pmc_summary.json (tools/r3_prof.sh uses the same)."""
    return f"{env_id} N={n}" + (" no-collision" if no_collision else "") + (" rollout" if rollout else "")


def profiled_counters(label):
    """Counters of the dominant kernel of a profiled configuration from the committed rocprofv3 summary (separate --pmc passes of this
    very command line; static figures of that profile, not measured in this run).  Returns (kernel name, traffic dict, valu dict).
    traffic: L2 <-> fabric bytes per launch.  FETCH_SIZE is reported by gfx950 at half the volume for WIDE coalesced reads, which is
    what the guide's x2 correction is for; this kernel reads mostly 8- / 16-byte gathers, so the true read volume lies between the
    uncorrected and the corrected figure -- both are given, `traffic` (the contract's field) carries the corrected (upper) one."""
    if not os.path.exists(PMC_SUMMARY):
        return None, None, None
    try:
        cfgs = json.load(open(PMC_SUMMARY))
        kernels = cfgs.get(label)
        if not kernels:
            return None, None, None
        name, k = max(kernels.items(), key=lambda kv: kv[1].get("calls", 0) * kv[1].get("avg_ns", 0.0))
        traffic = None
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            rd, wr = k["FETCH_SIZE"] * 1024.0, k["WRITE_SIZE"] * 1024.0
            traffic = {"corrected": 2.0 * rd + wr, "uncorrected": rd + wr, "read_uncorrected": rd, "write": wr}
        busy, lanes = k.get("valu_busy_fraction_of_simd_cycles"), k.get("valu_lane_utilisation")
        valu = {"busy_fraction_of_simd_cycles": busy, "lane_utilisation": lanes,
                "valu_frac": (busy * lanes) if busy is not None and lanes is not None else None,  # useful lanes / peak lane-issue capacity
                "wave_slot_occupancy": k.get("wave_slot_occupancy"), "wait_fraction_of_wave_cycles": k.get("wait_fraction_of_wave_cycles"),
                "valu_wave_instructions_per_launch": k.get("SQ_INSTS_VALU"), "scratch_bytes_per_lane": k.get("Scratch_Size"),
                "vgprs_elf": k.get("vgprs_elf"), "profiled_kernel_avg_us": (k.get("avg_ns") or 0.0) / 1e3 or None,
                "source": "profiles/r3/pmc_summary.json[%r][%r] (static: separate rocprofv3 --pmc passes of this command, not measured in this run)" % (label, name)}
        return name, traffic, valu
    except Exception:
        return None, None, None


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
INFORMATIONAL_BITS = 8 | 32  # include/urgym.h: a consumed penetration depth and a joint past its URDF limit are facts about the episode, not anomalies


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
        # The optional exchange of the path: every rank ends up with all observations (RCCL all-gather over xGMI).  It OVERLAPS the
        # next step: the step kernel starts rewriting the observation buffer at its very beginning (P1 parks the end-effector pose in
        # the rows, every workgroup stores its rows as it finishes), so the gather must not read that buffer while the next step runs.
        # After step t the rows are copied device-to-device into stage[t % 2] on the step's own stream (N x obs_dim x 4 B: a few
        # microseconds), the all-gather of stage[t % 2] into gathered[t % 2] runs on a SIDE stream behind an event, and the copy of
        # step t + 2 waits for that gather's event before it reuses the staging buffer.  A consumer reads gathered[t % 2].
        # (one-GPU rehearsal: gloo gathers through pinned host buffers; the host part of the gather of step t runs after step t + 1
        #  has been launched, so the same overlap is exercised)
        side = torch.cuda.Stream(device=dev)
        stage = [torch.empty((n, env.obs_dim), dtype=torch.float32, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * n, env.obs_dim), dtype=torch.float32, device="cpu" if gloo else dev) for _ in range(2)]
        host = [torch.empty((n, env.obs_dim), dtype=torch.float32, pin_memory=True) for _ in range(2)] if gloo else None
        ev_copy = [torch.cuda.Event() for _ in range(2)]
        ev_gather = [torch.cuda.Event() for _ in range(2)]
        pending = []          # rehearsal: (buffer index) of the gather whose host part is still to run
        check = {"step": None, "ref": None, "ok": None}  # one sampled step: its observations, cloned, against what the gather delivered

    def finish_host_gather(b):
        ev_gather[b].synchronize()  # the device-to-host copy of stage[b]
        dist.all_gather(list(gathered[b].view(world, n, env.obs_dim).unbind(0)), host[b])

    def one_step(k):
        env.step(actions[k % n_act])
        if gathered is not None:
            b = k & 1
            main = torch.cuda.current_stream(dev)
            main.wait_event(ev_gather[b])           # the gather that last read stage[b] (step k - 2) has finished
            stage[b].copy_(env.buf["observation"])  # on the step's stream: ordered after step k, before step k + 1
            ev_copy[b].record(main)
            if check["step"] == k:
                check["ref"] = env.buf["observation"].clone()
            with torch.cuda.stream(side):
                side.wait_event(ev_copy[b])
                if gloo:
                    host[b].copy_(stage[b], non_blocking=True)
                else:
                    dist.all_gather_into_tensor(gathered[b], stage[b])  # (c10d: the collective is ordered behind `side`, and `side` behind it)
                ev_gather[b].record(side)
            if gloo:
                # step k is on the device; now the host part of step k - 1's gather runs beside it
                if pending:
                    finish_host_gather(pending.pop())
                pending.append(b)
            if check["step"] is not None and k == check["step"] + 1 and check["ok"] is None:
                # step k (= sampled step + 1) has been launched; the gather of the sampled step must deliver the sampled step's rows
                pb = check["step"] & 1
                if not gloo:
                    ev_gather[pb].synchronize()
                own = gathered[pb].view(world, n, env.obs_dim)[rank]
                check["ok"] = bool(torch.equal(own.to(dev), check["ref"]))

    for k in range(args.warmup):
        one_step(k)
    rollout_actions = None
    if args.rollout and gathered is None:  # resident before the timed region, like the per-step action batches
        rollout_actions = torch.stack([actions[(args.warmup + k) % n_act] for k in range(args.steps)])
    if gathered is not None:
        check["step"] = args.warmup + args.steps // 2
    torch.cuda.synchronize(dev)
    # HIP events around every k-th step launch (an event pair costs the stream ~6 us, i.e. ~3 % of a 0.2 ms step if every launch carried
    # one): every 8th launch in the default 200-step run (25 launches timed), every 2nd in a short one (the driver's 20-step run: 10)
    every = 8 if args.steps >= 128 else (4 if args.steps >= 64 else 2)
    env.enable_timing(not args.no_kernel_timing, every=every)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if rollout_actions is not None:
        env.rollout(rollout_actions)
    else:
        for k in range(args.steps):
            one_step(args.warmup + k)
    if gathered is not None and gloo:
        while pending:
            finish_host_gather(pending.pop())
    torch.cuda.synchronize(dev)  # (every stream of the device: the last gather too)
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
    anomalies = int(((status & ~INFORMATIONAL_BITS) != 0).sum().item())   # nan, reset exhausted / collision, iteration cap, stale record
    informational = int(((status & INFORMATIONAL_BITS) != 0).sum().item())
    anomaly_bits = {name: int(((status & bit) != 0).sum().item()) for bit, name in STATUS_BITS.items()}
    episodes = int(env.buf["episode_id"].sum().item())
    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        algo = ALGO_BYTES[args.env] * n  # bytes one launch of the step kernel has to move, per rank
        label = config_label(args.env, n, args.no_collision, args.rollout)
        prof_kernel, traffic, valu = profiled_counters(label)
        achieved = algo / (step_us * 1e-6) / 1e9 if step_us > 0 else 0.0
        fused = args.env != "UR5OriReach-v1"
        kind_no = {"UR5OriReach-v1": 0, "UR5ObsReach-v1": 1, "UR5DynReach-v1": 2, "UR5StaReach-v1": 3}[args.env]
        epa = (args.env == "UR5ObsReach-v1") or args.no_collision
        kernel = (f"env_step_fused<{kind_no}, {str(epa).lower()}> (STEP workgroups + the refill of the previous step's episode records)" if fused
                  else f"env_kernel<{kind_no}, 0, false> (STEP; finished envs reset inside the launch)")
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
            # `achieved` / `peak` / `frac` price the kernel against the HBM roofline with the algorithmic bytes of SURVEY.md section 8(d), as
            # the measurement contract asks; `bound` names what really limits it: float64 VALU issue while the chip is full (three waves
            # per SIMD keep the vector pipes saturated), then the latency of the longest GJK chains in the tail of the launch
            # (DESIGN.md section 4).  valu_issue.valu_frac = VALU busy x lane utilisation = useful lanes / peak lane-issue capacity.
            "roofline": {"bound": "valu_issue/latency", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic["corrected"] if traffic else None,
                         "traffic_uncorrected": traffic["uncorrected"] if traffic else None,
                         "traffic_note": "L2 <-> fabric bytes per launch, FETCH_SIZE x 1024 (x 2: the gfx950 correction for wide coalesced reads; this "
                                         "kernel's reads are mostly 8- / 16-byte gathers, so the truth lies between the two figures) + WRITE_SIZE x 1024",
                         "traffic_source": "profiles/r3/pmc_summary.json (separate rocprofv3 --pmc passes of this command)" if traffic else None,
                         "valu_issue": valu,
                         "valu_frac": valu["valu_frac"] if valu else None,
                         "algorithmic_bytes_per_launch": algo,
                         "kernel": kernel, "profiled_kernel": prof_kernel,
                         "kernel_us": step_us, "reset_kernel_us": reset_us, "launches_timed": launches,
                         "algorithmic_bytes_per_env_step": ALGO_BYTES[args.env],
                         "note": "not HBM-bound: float64 VALU issue in the bulk of the launch, the dependent chain of the longest GJK searches in its tail (DESIGN.md section 4)"},
            "anomalous_envs": anomalies,                       # real anomalies only
            "informational_envs": informational,               # penetration depth consumed / joint past its URDF limit
            "envs_by_status_bit": anomaly_bits,                # include/urgym.h URGYM_STATUS_*
            "episodes_started": episodes,
        }
        if gathered is not None:
            out["config"]["gather_bytes_per_gpu_per_step"] = n * env.obs_dim * 4
            last = (args.warmup + args.steps - 1) & 1
            out["config"]["gather_overlapped_with_next_step"] = True
            out["config"]["gather_own_shard_intact"] = bool(torch.equal(gathered[last].view(world, n, env.obs_dim)[rank].to(dev), env.buf["observation"]))
            out["config"]["gather_delivered_step_t_while_step_t_plus_1_ran"] = check["ok"]
        if not args.no_cpu_baseline:  # rank 0 at any world size
            out["cpu_baseline"] = cpu_baseline(args.env, args.seed)
            out["cpu_baseline"]["parity_check"] = parity_check(env, actions[(args.warmup + args.steps) % n_act])
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
