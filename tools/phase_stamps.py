"""Diagnostic: per-wave phase timing of the step kernel from a -DURGYM_STAMPS build (ur_gym_amd/csrc/build/liburgym_stamps.so).

    make -C ur_gym_amd/csrc stamps && python tools/phase_stamps.py [--env UR5DynReach-v1] [--num-envs 65536]

Not part of the product: the stamped library is loaded in place of liburgym_hip.so only by this script."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ur_gym_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="UR5DynReach-v1")
ap.add_argument("--num-envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--reset-kernel", action="store_true", help="library built with -DURGYM_STAMP_MODE=1: stamps of the auto-reset kernel")
ap.add_argument("--lib", default="liburgym_stamps.so")
ap.add_argument("--envs-per-block", type=int, default=64, help="forces URGYM_STEP_ENVS so that the stamp layout is known")
ap.add_argument("--tiers", default="", help="E1,B,E2: two-tier geometry (URGYM_STEP_TIERS) instead of --envs-per-block")
ap.add_argument("--no-collision", action="store_true", help="check_collision=0 (BASELINE configs[1]: FK + reward only)")
args = ap.parse_args()
os.environ["URGYM_STEP_ENVS"] = str(args.envs_per_block)
if args.tiers:
    os.environ["URGYM_STEP_TIERS"] = args.tiers
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "build", args.lib)
from ur_gym_amd import make_vec

env = make_vec(args.env, num_envs=args.num_envs, seed=5, check_collision=not args.no_collision)
env.reset(seed=5)
gen = torch.Generator(device="cuda").manual_seed(5)
for _ in range(args.steps):
    env.step(torch.rand((args.num_envs, 6), device="cuda", generator=gen) * 2 - 1)
torch.cuda.synchronize()
groups = args.envs_per_block
blocks = min(8192, (args.num_envs + groups - 1) // groups)
if args.tiers:
    e1, nb, e2 = (int(x) for x in args.tiers.split(","))
    groups = f"{nb} x {e1}, then {e2}"
    blocks = min(8192, nb + (args.num_envs - nb * e1 + e2 - 1) // e2)
if args.reset_kernel:
    groups = int(os.environ.get('URGYM_RESET_ENVS', '4'))
    blocks = 2048  # upper bound; only the workgroups of the LAST launch are kept below
W, S = int(os.environ.get("URGYM_WAVES", "4")), 44
buf = np.zeros(blocks * W * S, dtype=np.uint64)
lib = env.lib
lib.urgym_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.urgym_debug_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(blocks, W, S).astype(np.int64)
if args.reset_kernel:
    t_last = st[:, 0, 8].max()
    st = st[(st[:, 0, 8] > t_last - 100000) & (st[:, 0, 9] >= st[:, 0, 8])]  # started within the last millisecond
    blocks = len(st)
t0 = st[:, :, 0].min(axis=1, keepdims=True)
# s_memtime counts shader cycles, s_memrealtime 100 MHz: calibrate one against the other over the block lifetimes
real = (st[:, :, 9] - st[:, :, 8]).astype(np.float64)
cyc = (st[:, :, 7] - st[:, :, 0]).astype(np.float64)
ghz = cyc.sum() / (real.sum() * 10.0) / 1e3 * 1e3 / 1e3
print(f"in-kernel clock ~ {cyc.sum() / real.sum() / 10.0 / 100.0:.2f} GHz")
tick = real.sum() / cyc.sum() / 100.0  # microseconds per shader cycle
def us(x): return x * tick
ph = {"start->P1 done": st[:, :, 1] - st[:, :, 0], "P1 barrier wait": st[:, :, 2] - st[:, :, 1], "first set-up": st[:, :, 3] - st[:, :, 2],
      "GJK loop": st[:, :, 4] - st[:, :, 3], "loop barrier wait": st[:, :, 6] - st[:, :, 4], "P4": st[:, :, 7] - st[:, :, 6]}
print(f"{args.env} N={args.num_envs} envs/block={groups} blocks={blocks}  (last step; mean over waves, microseconds)")
for k, v in ph.items():
    print(f"  {k:20s} mean {us(v.mean()):8.1f}  p50 {us(np.median(v)):8.1f}  max {us(v.max()):8.1f}")
trips = st[:, :, 5] & 0xFFFFFFFF
draws = st[:, :, 5] >> 32
loop = (st[:, :, 4] - st[:, :, 3])
print(f"  loop trips per wave: mean {trips.mean():.1f} max {trips.max()}   draws per wave: mean {draws.mean():.1f}")
print(f"  time per loop trip: {us(loop.sum()) / trips.sum():.2f} us")
if not args.reset_kernel:
    sect = np.concatenate([st[:, :, 12:18], st[:, :, 20:24]], axis=2).astype(np.float64)
    # section i ends at mark i (urgym_device.h URGYM_TRIP_MARK / the kernel's SECTION): 0 is never marked
    names = ("(unused)", "support of A: candidate record(s), point transform", "pose, cell code request, support of B | exits, vertex stored",
             "vertex reduction, convergence tests", "result handling", "polling, draw, set-up",
             "simplex front: segment case / plane tests", "simplex: face evaluations (triangle routine)", "(unused)", "(unused)")
    tot = sect.sum()
    print("  wave time inside the loop by section (microseconds per trip; share):")
    boxq, selfq, trips_self = (st[:, :, 18] & 0xFFFFFFFF), (st[:, :, 18] >> 32), st[:, :, 19]
    print(f"  pair queries per workgroup: {boxq.sum(axis=1).mean():.1f} link <-> table / track, {selfq.sum(axis=1).mean():.1f} link <-> link; "
          f"trips that carried a link <-> link query: {100.0 * trips_self.sum() / trips.sum():.1f} %")
    for i in (1, 2, 6, 7, 3, 4, 5):
        print(f"    {names[i]:52s} {us(sect[:, :, i].sum()) / trips.sum():6.2f}   {100 * sect[:, :, i].sum() / tot:5.1f} %")
    # lane counters: (executions by the wave, active lanes summed over them)
    cn = ("loop trip (lanes = busy lanes)", "hull support: first candidate record", "hull support: chained record", "simplex: segment case",
          "simplex: four plane tests (tetrahedron)", "simplex: one face evaluation", "  triangle exit: vertex A", "  triangle exit: vertex B",
          "  triangle exit: edge AB (division)", "  triangle exit: vertex C", "  triangle exit: edge AC (division)", "  triangle exit: edge BC (division)",
          "  triangle exit: face interior (division)", "vertex reduction", "draw + set-up", "result handling of a finished query",
          "cylinder support (sqrt + division)", "box support", "early exits (separating axis / duplicate / no progress)", "(unused)")
    cnt = st[:, :, 24:44]
    ex = (cnt & 0xFFFFFFFF).astype(np.float64).sum(axis=(0, 1)); ln = (cnt >> 32).astype(np.float64).sum(axis=(0, 1))
    T = float(trips.sum())
    busy = ln[0] / max(T, 1.0)  # (busy lanes summed over the trips / trips)
    print(f"  LANE TABLE (all waves of the last launch; a wave instruction is issued for 64 lanes whatever the exec mask holds)")
    print(f"    busy lanes per loop trip: {busy:.1f} of 64 = {100 * busy / 64:.1f} %   (idle-lane term: {100 * (1 - busy / 64):.1f} % of every issued lane is an idle lane)")
    print(f"    {'code':58s} {'executions per trip':>20s} {'active lanes per execution':>28s} {'of the busy lanes':>18s}")
    for i, nm in enumerate(cn[:19]):
        if ex[i] <= 0: continue
        a = ln[i] / ex[i]
        print(f"    {nm:58s} {ex[i] / T:20.2f} {a:28.1f} {100 * a / max(busy, 1e-9):17.1f} %")
    # time-weighted occupancy of the issued lanes: each section's time weighted with the lanes active in its dominant code
    occ = {1: ln[1] / max(ex[1], 1), 2: busy, 6: (ln[3] + ln[4]) / max(ex[3] + ex[4], 1), 7: (ln[6:13].sum() + ln[5]) / max(ex[6:13].sum() + ex[5], 1), 3: ln[13] / max(ex[13], 1),
           4: busy, 5: busy}
    tw = sum(sect[:, :, i].sum() * occ[i] / 64.0 for i in occ) / sum(sect[:, :, i].sum() for i in occ)
    print(f"    time-weighted share of active lanes over the loop (section time x lanes active in its dominant code): {100 * tw:.1f} %"
          f"  -> idle lanes {100 * (1 - busy / 64):.1f} %, divergence among the busy lanes {100 * (1 - tw / (busy / 64)):.1f} %")
if not args.reset_kernel:
    loop_end = st[:, :, 4] - st[:, :, 0].min(axis=1, keepdims=True)   # end of each wave's loop since its workgroup started
    last = loop_end.argmax(axis=1)
    print("  wave whose loop ends last (share of workgroups):", ", ".join(f"wave {w}: {100.0 * (last == w).mean():.0f} %" for w in range(W)),
          "| mean loop end per wave (us):", ", ".join(f"{us(loop_end[:, w].mean()):.0f}" for w in range(W)),
          "| trips per wave index:", ", ".join(f"{trips[:, w].mean():.1f}" for w in range(W)))
    slow = np.argsort(loop_end.max(axis=1))[-max(1, blocks // 20):]      # the slowest 5 % of the workgroups
    print("  slowest 5 % of the workgroups: last wave", ", ".join(f"{w}: {100.0 * (last[slow] == w).mean():.0f} %" for w in range(W)),
          f"| its trips mean {trips[slow, last[slow]].mean():.1f}, its loop end mean {us(loop_end[slow].max(axis=1).mean()):.0f} us")
blk = st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)
print(f"  block lifetime: mean {us(blk.mean()):.1f} us, max {us(blk.max()):.1f} us; kernel span {us(st[:, :, 7].max() - st[:, :, 0].min()):.1f} us")

occ = C.c_int(0)
lib.urgym_debug_occupancy.argtypes = [C.POINTER(C.c_int)]
print("  occupancy API: rc", lib.urgym_debug_occupancy(C.byref(occ)), "blocks per CU", occ.value)
b0 = st[:, :, 8].min(axis=1); b1 = st[:, :, 9].max(axis=1)
ts = np.linspace(b0.min(), b1.max(), 200)
conc = [(int(((b0 <= t) & (b1 >= t)).sum())) for t in ts]
print(f"  concurrent blocks over the kernel (100 MHz clock): max {max(conc)}, mean {np.mean(conc):.0f}; kernel span {(b1.max() - b0.min()) / 100.0:.1f} us")
print("  concurrency timeline (20 bins):", [conc[i] for i in range(5, 200, 10)])
print(f"  first block start spread: {(b0.max() - b0.min()) / 100.0:.1f} us; sum of block lifetimes / span = {((b1 - b0).sum()) / (b1.max() - b0.min()):.1f} blocks on average")
hw = st[:, 0, 10]; xcc = st[:, 0, 11]
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | ((xcc & 0xF) << 8)   # CU_ID | SE_ID | XCC
mid = 0.5 * (b0.min() + b1.max())
alive = (b0 <= mid) & (b1 >= mid)
u, cnt = np.unique(cu[alive], return_counts=True)
print(f"  at mid-kernel: {alive.sum()} blocks alive on {len(u)} distinct (xcc,se,cu) ids; blocks per id: max {cnt.max()}, mean {cnt.mean():.2f}")
