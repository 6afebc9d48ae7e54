"""CPU tests (-m "not gpu"): the C-ABI library loads and exports what include/urgym.h declares, the host logic fails
loudly without a GPU, and the N>1 sharding path works under gloo with world_size 2.  No compute call is made."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest
import torch

from ur_gym_amd import _abi, _native, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "urgym.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(urgym_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    assert int(re.search(r"#define URGYM_ABI_VERSION (\d+)", hdr).group(1)) == _abi.ABI_VERSION
    # struct field order of urgym_buffers == ctypes mirror
    body = hdr[hdr.index("typedef struct urgym_buffers"):hdr.index("} urgym_buffers;")]
    fields = re.findall(r"^\s*(?:double|float|int32_t|uint8_t)\*\s*(\w+);", body, flags=re.M)
    assert fields == [name for name, _, _ in _abi.BUFFER_FIELDS]
    cbody = hdr[hdr.index("typedef struct urgym_config"):hdr.index("} urgym_config;")]
    cfields = re.findall(r"^\s*(?:int32_t|double)\s+([^;]+);", cbody, flags=re.M)
    names = []
    for decl in cfields:
        for part in decl.split(","):
            names.append(re.sub(r"\[.*\]", "", part).strip())
    assert names == [f[0] for f in _abi.Config._fields_]


def test_library_loads_and_exports_every_symbol():
    lib = _native.lib()
    for sym in _abi.EXPORTED_SYMBOLS:
        assert hasattr(lib, sym), sym
    assert lib.urgym_abi_version() == _abi.ABI_VERSION


def test_config_defaults_mirror_reference_constants(oracle):
    lib = _native.lib()
    for kind in (0, 1, 2, 3):
        a, b = _abi.Config(), oracle.default_config(kind, 17)
        assert lib.urgym_config_default(kind, 17, C.byref(a)) == 0
        assert bytes(a) == bytes(b)  # the product and the oracle state the same constants independently
        od, gd = C.c_int(), C.c_int()
        assert lib.urgym_obs_dims(kind, C.byref(od), C.byref(gd)) == 0
        assert (od.value, gd.value) == _abi.OBS_DIMS[kind]
    c = _abi.Config()
    lib.urgym_config_default(_abi.ENV_DYN, 1, C.byref(c))
    assert c.max_episode_steps == 100 and c.dyn_motion_steps == 25 and abs(c.dt - 0.04) < 1e-15
    assert np.allclose(list(c.w_link), np.array([8, 2.4, 1.2, 1.2, 0.2]) / 13 * 50)  # reach.py:596-597
    assert list(c.neutral_q) == [0.0, -1.5708, 0.0, -1.5708, 0.0, 0.0]               # UR5.py:262
    assert (c.w_collision, c.w_success, c.w_distance, c.w_orientation) == (-500, 200, -70, -30)
    assert lib.urgym_config_default(7, 1, C.byref(c)) == _abi.ERR_ARG


def test_no_cpu_path():
    """device < 0 must be refused: the product has no CPU fallback (the oracle is test infrastructure only)."""
    lib = _native.lib()
    c = _abi.Config()
    lib.urgym_config_default(_abi.ENV_ORI, 4, C.byref(c))
    h = C.c_void_p()
    assert lib.urgym_create(C.byref(c), -1, C.byref(h)) == _abi.ERR_ARG
    assert b"no CPU path" in lib.urgym_last_error(None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine WITHOUT a GPU")
def test_vector_env_fails_loudly_without_gpu():
    from ur_gym_amd import make_vec

    with pytest.raises(_native.NativeError):
        make_vec("UR5DynReach-v1", num_envs=4)
    with pytest.raises(ValueError):
        make_vec("UR5Nope-v1", num_envs=4)


def test_product_never_imports_the_oracle():
    """The shipped path must not import, link, load or execute anything under oracle/ (comments may mention it)."""
    pkg = os.path.join(ROOT, "ur_gym_amd")
    bad = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|liburgym_oracle|urgym_oracle_|#include\s+\"[^\"]*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert not bad.search(txt), (dirpath, f)


def test_shard_range_partitions_exactly():
    for total in (1, 7, 64, 65536, 524288 + 3):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_range(total, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(8, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).repeat(1, 3)  # obs rows tagged with global id
    full = sharding.gather_observations(local)
    slowest = sharding.max_over_ranks(1.0 + rank)
    dist.barrier()
    q.put((rank, full[:, 0].tolist(), slowest, sharding.rank_seed(5, rank)))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_gather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ids, slowest, seed in res:
        assert ids == [float(i) for i in range(8)]  # every rank sees all shards in global order
        assert slowest == 2.0                       # max over ranks
        assert seed == 5 + 1000 * rank


def test_lazy_info_dict_semantics():
    """VectorEnv.step returns derived info entries lazily (no kernel launch unless read); it must still behave like a dict."""
    from ur_gym_amd.vector_env import _LazyInfo

    calls = []
    info = _LazyInfo({"is_success": 1}, {"_final_observation": lambda: (calls.append(1), 7)[1]})
    assert "_final_observation" in info and len(info) == 2 and not calls
    assert info.get("missing", 3) == 3 and not calls
    assert info["_final_observation"] == 7 and info["_final_observation"] == 7 and calls == [1]
    other = _LazyInfo({"a": 1}, {"b": lambda: 2})
    assert dict(other) == {"a": 1, "b": 2} and sorted(other) == ["a", "b"] and list(other.values()) == [1, 2]
    with pytest.raises(KeyError):
        info["nope"]


def test_bench_reads_the_committed_counters_by_configuration():
    """bench.py's roofline.traffic / valu_issue come from profiles/r3/pmc_summary.json, keyed by configuration and by the REAL kernel
    name (round 2 filed the fused launch under a legacy key): every BASELINE single-GPU configuration must be there, and the dominant
    kernel of the obstacle envs must be the fused step launch."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for env_id, n, nc, ro, kernel in (("UR5DynReach-v1", 65536, False, False, "env_step_fused<2, false>"),
                                      ("UR5ObsReach-v1", 16384, False, False, "env_step_fused<1, true>"),
                                      ("UR5OriReach-v1", 4096, False, False, "env_kernel<0, 0, false>"),
                                      ("UR5OriReach-v1", 4096, True, True, "env_kernel<0, 0, false>")):
        name, traffic, valu = bench.profiled_counters(bench.config_label(env_id, n, nc, ro))
        assert name == kernel, (env_id, name)
        assert traffic["corrected"] > traffic["uncorrected"] > 0
        assert 0.0 < valu["valu_frac"] < 1.0 and abs(valu["valu_frac"] - valu["busy_fraction_of_simd_cycles"] * valu["lane_utilisation"]) < 1e-12
        assert 100 <= valu["vgprs_elf"] <= 168   # three resident workgroups per CU: the 168-VGPR budget
    assert bench.profiled_counters("no such configuration") == (None, None, None)
    assert bench.INFORMATIONAL_BITS == 8 | 32
