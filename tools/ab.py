#!/usr/bin/env python3
"""A/B runner for kernel experiments on the GPU box: each variant is one `bench.py --no-cpu-baseline` child process with its own
library (URGYM_LIB) and environment switches; variants are interleaved over the repetitions so that clock drift hits all alike.

    python tools/ab.py --out gpurun_out/exp_x/results.jsonl --reps 2 \
        --variant base::                                  (name:library:ENV=V;ENV=V)
        --variant res2:ur_gym_amd/csrc/build/liburgym_res2.so:URGYM_STEP_ENVS=128 \
        -- --num-envs 65536                                (arguments after -- go to bench.py)

This process never touches the GPU itself.  One JSON line per run: variant, value (M env-steps/s), ms_per_step, kernel_us."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--variant", action="append", default=[])
ap.add_argument("--timeout", type=int, default=300)
ap.add_argument("bench_args", nargs="*")
args = ap.parse_args()
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
variants = []
for v in args.variant:
    name, lib, envs = (v.split(":") + ["", ""])[:3]
    env = dict(kv.split("=", 1) for kv in envs.split(";") if kv)  # ENV=V;ENV=V (values may hold commas)
    if lib:
        env["URGYM_LIB"] = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
    variants.append((name, env))
with open(args.out, "a") as f:
    for rep in range(args.reps):
        for name, env in variants:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args.bench_args
            try:
                p = subprocess.run(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=args.timeout)
                lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                if not lines:
                    rec = {"variant": name, "error": (p.stderr or "")[-400:], "rc": p.returncode}
                else:
                    d = json.loads(lines[-1])
                    rec = {"variant": name, "env": env, "args": " ".join(args.bench_args), "value_M": round(d["value"] / 1e6, 2),
                           "ms_per_step": round(d["ms_per_step"], 5), "kernel_us": round(d["roofline"]["kernel_us"], 2),
                           "anomalous_envs": d.get("anomalous_envs")}
            except subprocess.TimeoutExpired:
                rec = {"variant": name, "error": "timeout"}
                print(json.dumps(rec), flush=True)
                f.write(json.dumps(rec) + "\n")
                sys.exit(3)  # a run that hangs says something: stop, do not start the next GPU step
            print(json.dumps(rec), flush=True)
            f.write(json.dumps(rec) + "\n")
            f.flush()
