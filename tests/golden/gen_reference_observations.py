#!/usr/bin/env python3
"""Extract the observations the reference's OWN PyBullet environments produced and its checkpoints still carry.

    python tests/golden/gen_reference_observations.py [--reference /root/reference]

Every ``Trained_Models/Trained_*/best_model.zip`` is a stable-baselines3 archive whose ``data`` member is a JSON document.
SB3 stores, for the off-policy algorithm it saved, ``_last_original_obs`` and ``_last_obs`` -- the observations BEFORE and
AFTER the last environment step of training -- and, next to the pickled blob, a plain-text ``repr`` of each array
("achieved_goal": "[[ 0.7157695 -0.2308369 ...]]").  Only that text is read here: ``zipfile`` + ``json`` + a number
regex; nothing is unpickled or executed, and the float32 reprs round-trip exactly.

These are the only state -> observation samples of the reference's PyBullet path that exist anywhere in its repository
(it ships no tests and no golden vectors).  Two CONSECUTIVE observations of one episode per env pin:

  * forward kinematics + Bullet's Euler read-out:   q (slots 6..11) -> ee position / rpy (slots 0..5)        [8 samples]
  * PyBullet.get_link_distances (getClosestPoints incl. collision margins): the link_dist slots of observation t are the
    distances evaluated in compute_reward of step t-1 (core.py:311 vs 316), i.e. at the joint vector and obstacle pose
    of observation t-1                                                                                       [15 values]
  * PyBullet.step's integration of the moving obstacle (UR5DynReach-v1): pose(t-1) + velocity slot -> pose(t)  [1 sample]

Output: tests/golden/reference_observations.json (data only).
"""
import argparse
import json
import os
import re
import zipfile

import numpy as np

NUM = re.compile(r"-?\d+\.?\d*(?:e[-+]?\d+)?")


def parse(text):
    return [float(np.float32(x)) for x in NUM.findall(text)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    out = {}
    for name in ("Ori", "Obs", "Sta", "Dyn"):
        path = os.path.join(args.reference, "Trained_Models", f"Trained_{name}", "best_model.zip")
        data = json.loads(zipfile.ZipFile(path).read("data"))
        entry = {"num_timesteps": data["num_timesteps"], "source": f"Trained_Models/Trained_{name}/best_model.zip:data"}
        for key, tag in (("_last_original_obs", "before"), ("_last_obs", "after")):
            entry[tag] = {k: parse(data[key][k]) for k in ("observation", "achieved_goal", "desired_goal")}
        out[name.lower()] = entry
        print(name, {t: len(entry[t]["observation"]) for t in ("before", "after")})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_observations.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
