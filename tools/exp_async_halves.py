#!/usr/bin/env python3
"""Experiment (GPU box): the headline workload (UR5DynReach-v1, 65536 envs, random actions, auto-reset) as ONE handle stepped on
one stream (what bench.py measures) against the same envs split over K handles stepped on K streams without synchronising in
between -- the half-batch / asynchronous vector-env pattern: the tail of one launch (the few workgroups with the longest GJK
chains) then overlaps the body of the next handle's launch.  Every handle still gets its actions per step."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ur_gym_amd import make_vec

N, STEPS, WARM = 65536, 200, 120  # (warm-up past the first episodes: the record fallbacks are gone by then)
out = []
for k in (1, 2, 3, 4):
    n = N // k
    if n * k != N:
        n = (N + k - 1) // k
    envs = [make_vec("UR5DynReach-v1", num_envs=n, seed=10 + i) for i in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    gen = torch.Generator(device="cuda").manual_seed(1)
    acts = [torch.rand((64, n, 6), device="cuda", generator=gen) * 2 - 1 for _ in range(k)]  # (64 distinct batches, like bench.py)
    for i, e in enumerate(envs):
        e.reset(seed=10 + i)
    torch.cuda.synchronize()
    def run(steps):
        for t in range(steps):
            for i, e in enumerate(envs):
                with torch.cuda.stream(streams[i]):
                    e.step(acts[i][t % 64])
    run(WARM)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.append({"handles": k, "envs_per_handle": n, "value_M": round(N * STEPS / dt / 1e6, 2), "ms_per_round_of_steps": round(dt / STEPS * 1e3, 4)})
    print(json.dumps(out[-1]), flush=True)
    for e in envs:
        e.close()
