"""First-contact GPU probe (not a pytest file): reset + a few steps of each env kind against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ur_gym_amd import make_vec, _abi
from oracle import binding as ob

def compare(kind_id, N=256, steps=30, seed=3):
    kind = _abi.ENV_IDS[kind_id]
    env = make_vec(kind_id, num_envs=N, seed=seed)
    orc = ob.OracleEnv(kind, N, threads=8)
    obs, info = env.reset(seed=seed)
    orc.reset(seed=seed)
    torch.cuda.synchronize()
    st = env.get_state()
    worst = {}
    def upd(name, a, b):
        d = float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))) if np.size(a) else 0.0
        worst[name] = max(worst.get(name, 0.0), d)
    for k in ("q", "goal", "obst_start", "obst_end", "obst_pos", "obst_quat", "obst_vel", "link_dist"):
        upd("reset." + k, st[k], orc.buf[k])
    upd("reset.obs", env.buf["observation"].cpu().numpy(), orc.buf["observation"])
    rng = np.random.default_rng(seed)
    nterm = 0
    for t in range(steps):
        a = rng.uniform(-1, 1, (N, 6)).astype(np.float32)
        env.step(torch.from_numpy(a).cuda())
        orc.step(a)
        torch.cuda.synchronize()
        for k in ("observation", "achieved_goal", "desired_goal", "reward", "final_observation"):
            if k == "final_observation":
                m = orc.buf["terminated"].astype(bool) | orc.buf["truncated"].astype(bool)
                upd(k, env.buf[k].cpu().numpy()[m], orc.buf[k][m])
            else:
                upd(k, env.buf[k].cpu().numpy(), orc.buf[k])
        for k in ("terminated", "truncated", "is_success", "collision"):
            mism = int((env.buf[k].cpu().numpy() != orc.buf[k]).sum())
            worst["mismatch." + k] = worst.get("mismatch." + k, 0) + mism
        st = env.get_state()
        for k in ("q", "goal", "obst_pos", "obst_quat", "link_dist"):
            upd("state." + k, st[k], orc.buf[k])
        upd("state.step", st["step_count"], orc.buf["step_count"])
        nterm += int(orc.buf["terminated"].sum())
        # teacher-force: keep both on the oracle's state so one discrepancy does not cascade
    print(kind_id, "terminated total", nterm, "status gpu", np.unique(env.buf["status"].cpu().numpy()), "oracle", np.unique(orc.buf["status"]))
    for k, v in worst.items():
        print(f"   {k:28s} {v:.3e}")
    env.close()

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for kid in ("UR5OriReach-v1", "UR5ObsReach-v1", "UR5DynReach-v1"):
        t0 = time.time()
        compare(kid)
        print("  took", time.time() - t0)
