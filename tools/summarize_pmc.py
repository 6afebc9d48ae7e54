#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/r3_prof.sh: one JSON object per profiled CONFIGURATION, keyed by the REAL kernel names
(`env_step_fused<2, false>`, `env_kernel<0, 0, false>`, ...), mean counter values per launch.

    python tools/summarize_pmc.py [--elf build/resource_usage.txt] "<label>::<dir>" ["<label>::<dir>" ...]

<dir> holds the passes of one configuration: trace/ (--kernel-trace --stats) and one directory per --pmc group.
Derived figures (MI355X_MICROARCH.md): SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES is summed over the 32 shader engines.
`VGPR_Count` as rocprofv3 prints it is the kernel descriptor's figure in 8-byte (two-dword) units on this wave64 target -- half the
per-lane count; `vgprs_elf` is the ELF metadata's count (hipcc -Rpass-analysis=kernel-resource-usage), the authoritative one."""
import glob, json, re, sys
import pandas as pd

args = sys.argv[1:]
elf = {}
if args and args[0] == "--elf":
    txt = open(args[1]).read()
    # remark blocks: "Function Name: <mangled>" followed by "VGPRs: n", "ScratchSize [bytes/lane]: n", "LDS Size [bytes/block]: n", "VGPRs Spill: n"
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", txt, re.S):
        elf[m.group(1)] = {"vgprs_elf": int(m.group(2)), "scratch_elf": int(m.group(3)), "vgpr_spills_elf": int(m.group(4)), "lds_elf": int(m.group(5))}
    args = args[2:]


def mangled_guess(name):
    """env_step_fused<2, false> -> env_step_fusedILi2ELb0E ; env_kernel<0, 0, false> -> env_kernelILi0ELi0ELb0E"""
    m = re.match(r"(env_step_fused|env_kernel)<([^>]*)>", name)
    if not m:
        return None
    parts = [p.strip() for p in m.group(2).split(",")]
    enc = "".join(("Lb1E" if p == "true" else "Lb0E" if p == "false" else f"Li{p}E") for p in parts)
    return f"{m.group(1)}I{enc}"


def short(name):
    m = re.search(r"(env_step_fused<[^>]*>|env_kernel<[^>]*>)", name)
    return m.group(1) if m else None


out = {}
for spec in args:
    label, d = spec.split("::", 1)
    cfg = {}
    for f in glob.glob(f"{d}/*/*/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"):
        df = pd.read_csv(f)
        df = df[df.Kernel_Name.str.contains("env_kernel|env_step_fused")]
        if df.empty:
            continue
        df["kernel"] = df.Kernel_Name.map(short)
        g = df.groupby(["kernel", "Counter_Name"]).Counter_Value.mean().unstack()
        for k, row in g.iterrows():
            cfg.setdefault(k, {}).update({c: float(v) for c, v in row.items()})
        meta = df.groupby("kernel")[["VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"]].first()
        for k, row in meta.iterrows():
            cfg[k].update({c: int(v) for c, v in row.items()})
    for f in glob.glob(f"{d}/trace/*/*kernel_stats.csv") + glob.glob(f"{d}/trace/*kernel_stats.csv"):
        ks = pd.read_csv(f)
        ks = ks[ks.Name.str.contains("env_kernel|env_step_fused")]
        for _, r in ks.iterrows():
            cfg.setdefault(short(r.Name), {}).update({"calls": int(r.Calls), "avg_ns": float(r.AverageNs), "min_ns": float(r.MinNs), "max_ns": float(r.MaxNs)})
    for k, v in cfg.items():
        if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
            v["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
        if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            v["wait_fraction_of_wave_cycles"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
        if v.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in v:
            v["kernel_shader_cycles"] = v["SQ_BUSY_CYCLES"] / 32.0
            v["valu_busy_fraction_of_simd_cycles"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (1024.0 * v["kernel_shader_cycles"])
            v["wave_slot_occupancy"] = 4.0 * v["SQ_WAVE_CYCLES"] / (256.0 * 12.0 * v["kernel_shader_cycles"]) if v.get("LDS_Block_Size", 0) else None
            v["valu_frac"] = v["valu_busy_fraction_of_simd_cycles"] * v.get("valu_lane_utilisation", 0.0)
        if "FETCH_SIZE" in v:
            v["hbm_read_bytes_per_launch_uncorrected"] = v["FETCH_SIZE"] * 1024
            v["hbm_read_bytes_per_launch_corrected"] = v["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE = 1/2 of WIDE reads (MI355X_MICROARCH.md)
        if "WRITE_SIZE" in v:
            v["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        mg = mangled_guess(k)
        for name, e in elf.items():
            if mg and mg in name:
                v.update(e)
    out[label] = cfg
print(json.dumps(out, indent=1))
