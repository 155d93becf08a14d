#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into one JSON/markdown summary under profiles/.

usage: tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<name>
Per kernel: calls, average duration (kernel trace) and the per-launch average of every PMC counter
collected.  HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced streaming reads and is uncalibrated for other widths, so both the raw and the doubled
figure are reported.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("crt::", "")
    return name.split("(")[0]


def isa_metadata(path):
    """per-kernel register / LDS / scratch figures from the code-object metadata in the compiler's assembly (make kernels.s)"""
    import re
    import subprocess
    meta, cur = {}, None
    if not os.path.exists(path):
        return meta

    def close(c):
        if c and "name" in c:
            try:
                dem = subprocess.check_output(["c++filt", c["name"]], text=True).strip()
            except Exception:
                dem = c["name"]
            meta[short(dem)] = {k: v for k, v in c.items() if k != "name"}
    fields = {".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds",
              ".agpr_count": "agpr"}
    for line in open(path):
        if line.startswith("  - .agpr_count:"):
            close(cur)
            cur = {"agpr": int(line.split()[-1])}
            continue
        if cur is None:
            continue
        m = re.match(r"    (\.\w+):\s+(\S+)\s*$", line)        # kernel-level keys sit at four spaces; argument fields deeper
        if not m:
            continue
        if m.group(1) == ".name":
            cur["name"] = m.group(2)
        elif m.group(1) in fields:
            cur[fields[m.group(1)]] = int(m.group(2))
    close(cur)
    return meta


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {"source": src, "kernels": {}}
    isa = isa_metadata(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cudaraytracer_amd", "csrc", "kernels.s"))
    for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                 "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
    for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            acc = defaultdict(lambda: defaultdict(list))
            meta = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                           "scratch": int(r["Scratch_Size"])}
            for k, cs in acc.items():
                e = out["kernels"].setdefault(k, {})
                e.update(meta[k])
                pm = e.setdefault("pmc_per_launch_avg", {})
                for c, v in cs.items():
                    pm[c] = sum(v) / len(v)
    for k, e in out["kernels"].items():
        pm = e.get("pmc_per_launch_avg", {})
        if "FETCH_SIZE" in pm:
            e["hbm_read_bytes_per_launch_raw"] = pm["FETCH_SIZE"] * 1024
            e["hbm_read_bytes_per_launch_x2_gfx950"] = pm["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in pm:
            e["hbm_write_bytes_per_launch"] = pm["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in pm and "TCC_MISS_sum" in pm and pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in pm and pm.get("SQ_WAVE_CYCLES", 0) > 0:
            if "SQ_ACTIVE_INST_VALU" in pm:
                e["valu_active_frac_of_wave_cycles"] = pm["SQ_ACTIVE_INST_VALU"] * 4 / pm["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_ANY" in pm:
                e["wait_any_frac_of_wave_cycles"] = pm["SQ_WAIT_ANY"] * 4 / pm["SQ_WAVE_CYCLES"] if False else pm["SQ_WAIT_ANY"] / pm["SQ_WAVE_CYCLES"]
        if "SQ_THREAD_CYCLES_VALU" in pm and pm.get("SQ_ACTIVE_INST_VALU", 0) > 0:
            e["valu_lane_utilisation"] = pm["SQ_THREAD_CYCLES_VALU"] / (pm["SQ_ACTIVE_INST_VALU"] * 64)
        if pm.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_ACTIVE_INST_VALU" in pm:
            e["active_inst_valu_over_busy_cycles"] = pm["SQ_ACTIVE_INST_VALU"] / pm["SQ_BUSY_CYCLES"]
        if pm.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_ACTIVE_INST_VALU" in pm:
            # rocprof's derived VALUBusy: 4 x SQ_ACTIVE_INST_VALU / SIMDs / clocks.  On gfx950 SQ_ACTIVE_INST_VALU advances once per
            # VALU wave-instruction (it equals SQ_INSTS_VALU within 1 %), so the formula prices every instruction at 4 cycles: a mix
            # of 2-cycle and 4-cycle opcodes that saturates the SIMDs reads above 1 here and as its true cycles/instruction x count
            # in bench.py's roofline.frac
            e["valu_busy_rocprof_formula"] = 4 * pm["SQ_ACTIVE_INST_VALU"] / 1024 / (pm["GRBM_GUI_ACTIVE"] / 8)
        if pm.get("GRBM_GUI_ACTIVE", 0) > 0 and e.get("avg_ms", 0) > 0:
            e["clock_ghz"] = pm["GRBM_GUI_ACTIVE"] / 8 / (e["avg_ms"] * 1e6)          # the counter sums the 8 XCDs
            if "SQ_WAVE_CYCLES" in pm:
                # SQ_WAVE_CYCLES advances once per 4 clocks per resident wave (5120 waves x launch clocks / 4 reproduces it for
                # the persistent k_trace8); 1024 SIMDs
                e["achieved_waves_per_simd"] = 4 * pm["SQ_WAVE_CYCLES"] / (pm["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if k in isa:
            e["isa"] = isa[k]
    for name in ("bench_kt.json",):
        p = os.path.join(src, name)
        if os.path.exists(p):
            try:
                out["bench_line_under_kernel_trace"] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception:
                pass
    khash = None
    try:
        khash = json.load(open(os.path.join(src, "isa_mix.json"))).get("kernel_hash")
    except Exception:
        pass
    out["k_trace8_kernel_hash"] = khash
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    json.dump(out, open(dst + ".json", "w"), indent=1, sort_keys=True)
    # HBM-side traffic of the dominant kernel for bench.py's roofline.traffic (it cannot collect PMC itself)
    for k, e in out["kernels"].items():
        if k.startswith("k_trace8<false") and "hbm_read_bytes_per_launch_raw" in e and "hbm_write_bytes_per_launch" in e:
            json.dump({"kernel": k, "fetch_size_kib_per_launch": e["pmc_per_launch_avg"]["FETCH_SIZE"],
                       "write_size_kib_per_launch": e["pmc_per_launch_avg"]["WRITE_SIZE"],
                       "hbm_bytes_per_launch": e["hbm_read_bytes_per_launch_x2_gfx950"] + e["hbm_write_bytes_per_launch"],
                       "valu_wave_instructions_per_launch": e["pmc_per_launch_avg"].get("SQ_INSTS_VALU"),
                       "valu_lane_utilisation": e.get("valu_lane_utilisation"),
                       "valu_busy_rocprof_formula": e.get("valu_busy_rocprof_formula"),
                       "kernel_hash": khash, "avg_launch_ms_under_kernel_trace": e.get("avg_ms"),
                       "rule": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts a 128-byte read request as 64 bytes) + WRITE_SIZE KiB x 1024, "
                               "MI355X_MICROARCH.md 'HBM'; the x2 is calibrated for 16-byte-per-lane reads, which these are",
                       "source": dst + ".json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, "
                                 "bench.py --steps 2 --warmup 0)"},
                      open(os.path.join(os.path.dirname(dst) or ".", "latest_traffic.json"), "w"), indent=1)
    with open(dst + ".md", "w") as f:
        f.write(f"# rocprofv3 summary ({src})\n\n"
                "VGPR: `rocprofv3` reports `VGPR_Count` in units of two registers on gfx950 (96 -> 48, 128 -> 64); the column shows the "
                "profiler's figure / the code object's `.vgpr_count` (+ bytes of scratch) from the compiler's assembly, which is what "
                "DESIGN.md quotes and what sets the occupancy.  Waves/SIMD = 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): "
                "resident waves per SIMD averaged over the launch (separate PMC passes of the same command).\n\n"
                "| kernel | calls | avg ms | % | VGPR prof / ISA (+scratch B) | LDS | waves/SIMD | clock GHz | L2 hit | HBM read/launch (raw, x2) | HBM write/launch | VALU lane util | VALUBusy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / clocks | SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("total_ms", 0)):
            f.write("| {} | {} | {:.4f} | {:.2f} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |\n".format(
                k, e.get("calls", ""), e.get("avg_ms", 0), e.get("pct", 0),
                "{} / {}".format(e.get("vgpr", ""), "{} (+{})".format(e["isa"].get("vgpr"), e["isa"].get("scratch")) if "isa" in e else "?"),
                e.get("lds", ""),
                "{:.2f}".format(e["achieved_waves_per_simd"]) if "achieved_waves_per_simd" in e else "",
                "{:.3f}".format(e["clock_ghz"]) if "clock_ghz" in e and e.get("avg_ms", 0) > 0.1 else "",
                "{:.3f}".format(e["l2_hit_rate"]) if "l2_hit_rate" in e else "",
                "{:.1f} MB, {:.1f} MB".format(e["hbm_read_bytes_per_launch_raw"] / 1e6, e["hbm_read_bytes_per_launch_x2_gfx950"] / 1e6) if "hbm_read_bytes_per_launch_raw" in e else "",
                "{:.1f} MB".format(e["hbm_write_bytes_per_launch"] / 1e6) if "hbm_write_bytes_per_launch" in e else "",
                "{:.3f}".format(e["valu_lane_utilisation"]) if "valu_lane_utilisation" in e else "",
                "{:.3f}".format(e["valu_busy_rocprof_formula"]) if "valu_busy_rocprof_formula" in e else "",
                "{:.2f}".format(e["active_inst_valu_over_busy_cycles"]) if "active_inst_valu_over_busy_cycles" in e else ""))
        f.write("\nVALUBusy: rocprof's derived formula prices every VALU wave-instruction at 4 cycles (on gfx950 SQ_ACTIVE_INST_VALU counts "
                "instructions: it equals SQ_INSTS_VALU within 1 %), so a saturated mix of 2- and 4-cycle opcodes reads above 1; bench.py's "
                "roofline.frac prices the same count with the kernel's own cycles per instruction.  SQ_BUSY_CYCLES sums 32 shader-engine "
                "counters (8 XCDs x 4), hence ratios of ~8 for a kernel that issues on every SIMD.  k_trace8 kernel hash of these passes: "
                + str(khash) + "\n")
    print("wrote", dst + ".json", dst + ".md")


if __name__ == "__main__":
    main()
