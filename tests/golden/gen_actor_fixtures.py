#!/usr/bin/env python3
"""Export the deterministic actor (latent_pi + mu) of the reference's shipped SAC checkpoints as plain float32 arrays.

    python tests/golden/gen_actor_fixtures.py [--reference /root/reference]

Source: Trained_Models/Trained_{Ori,Obs,Sta,Dyn}/best_model.zip -> policy.pth, loaded with torch.load(weights_only=True)
(nothing from the file is executed).  These are DATA files of the reference (SB3 MultiInputPolicy weights); the GPU box
has no /root/reference, so the closed-loop tests read these fixtures instead.  Also records the closed-loop results the
reference ships next to each checkpoint (best.txt / best_modeltest_result.txt): the two aggregate lines and statistics of the
per-trial rows "reward, success, last step" (model_test.py:59-60) -- early failures (an unsuccessful trial that ended before
step 99 = a collision), time-outs, percentiles of the last step and mean reward of the successful trials.  Round 3: the per-trial
rows themselves go to reference_trials.npz (three numbers per trial: episode reward, success, last step index -- data the reference
ships, not source), so that the tests can compare DISTRIBUTIONS (quantiles of the episode reward, standard errors of every mean)
instead of means with hand-picked tolerances.
"""
import argparse
import io
import json
import os
import re
import zipfile

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "actors")
    os.makedirs(out_dir, exist_ok=True)
    summary = {}
    trials = {}
    for name, result_file in (("Ori", "best.txt"), ("Obs", "best.txt"), ("Sta", "best.txt"), ("Dyn", "best_modeltest_result.txt")):
        d = os.path.join(args.reference, "Trained_Models", f"Trained_{name}")
        z = zipfile.ZipFile(os.path.join(d, "best_model.zip"))
        sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True, map_location="cpu")
        arrs = {k.replace("actor.", "").replace(".", "_"): v.numpy().astype(np.float32)
                for k, v in sd.items() if k.startswith("actor.latent_pi") or k.startswith("actor.mu")}
        np.savez_compressed(os.path.join(out_dir, f"actor_{name.lower()}.npz"), **arrs)
        lines = open(os.path.join(d, result_file)).read().splitlines()
        rate = float(re.search(r"([\d.]+)%", lines[0]).group(1))
        reward = float(re.search(r"(-?[\d.]+)\s*$", lines[1]).group(1))
        rows = [l.split(",") for l in lines[2:] if l.strip()]
        steps = np.array([float(r[2]) for r in rows])
        rew = np.array([float(r[0]) for r in rows])
        ok = np.array([float(r[1]) for r in rows]) > 0.5
        summary[name.lower()] = {"success_rate_percent": rate, "mean_episode_reward": reward, "trials": len(rows),
                                 "mean_last_step_index": float(steps.mean()), "in_features": int(arrs["latent_pi_0_weight"].shape[1]),
                                 "early_fail_percent": 100.0 * float((~ok & (steps < 99)).mean()),
                                 "timeout_percent": 100.0 * float((steps >= 99).mean()),
                                 "success_last_step_p5": float(np.percentile(steps[ok], 5)),
                                 "success_last_step_p50": float(np.percentile(steps[ok], 50)),
                                 "success_last_step_p95": float(np.percentile(steps[ok], 95)),
                                 "mean_success_reward": float(rew[ok].mean()), "mean_failure_reward": float(rew[~ok].mean()),
                                 "sd_episode_reward": float(rew.std(ddof=1)), "sd_success_reward": float(rew[ok].std(ddof=1)),
                                 "sd_failure_reward": float(rew[~ok].std(ddof=1)), "sd_last_step_index": float(steps.std(ddof=1)),
                                 "failures": int((~ok).sum()),
                                 "episode_reward_quantiles": {str(q): float(np.percentile(rew, q)) for q in (5, 25, 50, 75, 95)}}
        trials[f"{name.lower()}_reward"] = rew.astype(np.float64)
        trials[f"{name.lower()}_success"] = ok.astype(np.uint8)
        trials[f"{name.lower()}_last_step"] = steps.astype(np.uint8)
        print(name, {k: v.shape for k, v in arrs.items()}, summary[name.lower()])
    with open(os.path.join(out_dir, "reference_results.json"), "w") as f:
        json.dump(summary, f, indent=1)
    np.savez_compressed(os.path.join(out_dir, "reference_trials.npz"), **trials)


if __name__ == "__main__":
    main()
