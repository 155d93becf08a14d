#!/usr/bin/env python3
"""tools/isa_mix.py -- what does one iteration of k_trace8 cost in VALU issue cycles?

Reads the gfx950 assembly of kernels.hip (hipcc -S --cuda-device-only), cuts k_trace8<false> into the regions its
`; CRT_MARK <name>` comment lines delimit (kernels.hip: refill, vote, node step, primitive step, finish), counts the
wave-level VALU / SALU / VMEM / LDS instructions of each region and prices every VALU opcode with the issue cost
MEASURED by tools/ubench/valu_issue.hip (profiles/r02_ubench_valu_issue.jsonl: cycles per wave64 instruction per
SIMD, the lowest over the occupancies measured: ~2.3 for the full-rate opcodes, ~4.1 for the half-rate ones, ~8.1 for
v_rcp/v_sqrt).

Output (JSON): per region the instruction counts and VALU issue cycles -- with the measured cost of every opcode
(`valu_issue_cycles`) and with the nominal cost of its rate class (`valu_issue_cycles_nominal`: 2 / 4 / 8 cycles; the
measurement decides the class, the nominal figure is the floor no stream can go below, which bench.py's roofline uses).  bench.py multiplies them by the iteration
counts the counting frame reports (node iterations, primitive iterations, refills, finishes) to get the launch's VALU
instruction count and its issue-cycle demand -- the numerator of the roofline -- and checks the instruction count
against SQ_INSTS_VALU of the committed PMC pass.

usage: tools/isa_mix.py kernels.s profiles/r02_ubench_valu_issue.jsonl out.json [waves_per_simd; 0 (default) = best over occupancies]
"""
import json
import os
import re
import sys

KERNEL = os.environ.get("CRT_ISA_KERNEL", "_ZN3crt8k_trace8ILb0ELi256ELb0EEEvNS_9TraceArgsE")
# opcodes the microbenchmark does not cover are priced by a measured sibling of the same encoding / unit
SIBLING = {
    "v_cndmask_b32_e64": "v_cndmask_b32 SGPR-pair mask (VOP3)",
    "v_cndmask_b32_e32": "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc",       # behind the v_cmp that wrote vcc (the back-to-back case is 23 cycles)
    "v_cndmask_b32": "v_cndmask_b32 SGPR-pair mask (VOP3)",
    "v_max_f32": "v_max_f32", "v_min_f32": "v_min_f32", "v_fma_mix_f32": "v_fma_mix_f32 (lo half)", "v_min_u32": "v_max_u32",
    "v_addc_co_u32": "v_cmp_le_f32 vcc + v_addc_co_u32", "v_subrev_u32": "v_sub_u32", "v_subrev_f32": "v_sub_f32",
    "v_mov_b64": "v_pk_mul_f32 / v_pk_add_f32", "v_pk_mov_b32": "v_pk_mul_f32 / v_pk_add_f32", "v_pk_mul_f32": "v_pk_mul_f32 / v_pk_add_f32",
    "v_pk_add_f32": "v_pk_mul_f32 / v_pk_add_f32", "v_lshl_add_u64": "v_lshl_add_u32", "v_lshlrev_b64": "v_lshl_add_u32",
    "v_bfe_i32": "v_bfe_u32", "v_bitop3_b32": "v_or3_b32", "v_div_scale_f32": "v_max3_f32", "v_div_fmas_f32": "v_max3_f32",
    "v_div_fixup_f32": "v_max3_f32", "v_rcp_f32": "v_rcp_f32", "v_sqrt_f32": "v_sqrt_f32", "v_ffbl_b32": "v_ffbh_u32",
    "v_readfirstlane_b32": "v_mov_b32_dpp quad_perm", "v_mbcnt_lo_u32_b32": "v_bcnt_u32_b32", "v_mbcnt_hi_u32_b32": "v_bcnt_u32_b32",
    "v_sub_co_u32": "v_cmp_le_f32 vcc + v_addc_co_u32", "v_add_co_u32": "v_cmp_le_f32 vcc + v_addc_co_u32",
    "v_mul_hi_u32": "v_mul_lo_u32", "v_fmac_f32": "v_fmac_f32", "v_mad_u64_u32": "v_mul_lo_u32",
}
CMP = re.compile(r"^v_cmp[x]?_")
# nominal issue cycles of the three rate classes on the 32-lane SIMD (a wave64 instruction takes 2 passes at full rate); the
# microbenchmark measures 2.2-2.4 / 4.1 / 8.1 including its own loop overhead.  The class of an opcode comes from its measurement.
NOMINAL = {"full": 2.0, "half": 4.0, "quarter": 8.0}


def load_costs(path, k):
    """issue cost per class: the LOWEST cycles per instruction per SIMD measured at any occupancy (k = 0) -- what the
    hardware can do at best, so that a sum of such costs is a lower bound of the issue cycles a stream needs and
    demand / supply cannot exceed 1 -- or the cost at exactly k waves per SIMD"""
    cost = {}
    for line in open(path):
        r = json.loads(line)
        if r.get("bench") != "valu_issue" or (k and r["waves_per_simd"] != k):
            continue
        c = r["cycles_per_instruction_simd"]
        cost[r["class"]] = min(c, cost.get(r["class"], c))
    return cost


def price(op, cost):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if CMP.match(op):
        return cost["v_cmp_l[et]_f32 -> SGPR pair (VOP3)"] if op.endswith("_e64") else cost["v_cmp_l[et]_f32 -> vcc"], "half"
    for key in (op, base):
        if key in SIBLING and SIBLING[key] in cost:
            c = cost[SIBLING[key]]
            return c, ("full" if c < 3.2 else "half" if c < 6 else "quarter")
    for key in (base, op):
        if key in cost:
            c = cost[key]
            return c, ("full" if c < 3.2 else "half" if c < 6 else "quarter")
    return None, "unknown"


def main():
    asm, ub, out = sys.argv[1], sys.argv[2], sys.argv[3]
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    cost = load_costs(ub, k)
    half_default = cost["v_perm_b32"]
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    region = "prologue"
    regions = {}
    unknown = {}
    import hashlib
    digest = hashlib.sha256()       # of the kernel's instruction stream: ties a PMC pass to the binary it profiled (bench.py pmc_matches_build)
    for line in lines[start + 1:]:
        t = line.strip()
        if t and t[0] not in ";." and not t.endswith(":"):
            digest.update((" ".join(t.split(";")[0].split()) + "\n").encode())
        if t.startswith("s_endpgm"):
            break
        m = re.match(r";\s*CRT_MARK\s+(\w+)", t)
        if m:
            name = m.group(1)
            region = name[:-6] if name.endswith("_begin") else "between_" + name[:-4] if name.endswith("_end") else name
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        r = regions.setdefault(region, {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "valu_issue_cycles": 0.0,
                                        "valu_full_rate": 0, "valu_half_rate": 0, "valu_quarter_rate": 0, "opcodes": {}})
        if op.startswith("v_"):
            c, cls = price(op, cost)
            if c is None:
                c, cls = half_default, "half"
                unknown[op] = unknown.get(op, 0) + 1
            r["valu"] += 1
            r["valu_issue_cycles"] += c
            r["valu_issue_cycles_nominal"] = r.get("valu_issue_cycles_nominal", 0.0) + NOMINAL[cls]
            r["valu_%s_rate" % cls] += 1
            r["opcodes"][op] = r["opcodes"].get(op, 0) + 1
        elif op.startswith("s_"):
            if op not in ("s_waitcnt", "s_nop"):
                r["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            r["vmem"] += 1
        elif op.startswith("ds_"):
            r["lds"] += 1
    for r in regions.values():
        r["valu_issue_cycles"] = round(r["valu_issue_cycles"], 2)
    res = {"kernel": "k_trace8<false, 256, false>", "kernel_hash": digest.hexdigest()[:16], "waves_per_simd": k, "cost_source": ub,
           "cycles_full_rate": cost["v_mov_b32"], "cycles_half_rate": half_default, "cycles_quarter_rate": cost["v_rcp_f32"],
           "regions": regions, "opcodes_priced_as_half_rate_by_default": unknown}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for name, r in regions.items():
        print(f"{name:22s} valu {r['valu']:4d} (full {r['valu_full_rate']:3d}, half {r['valu_half_rate']:3d}, quarter {r['valu_quarter_rate']:2d}) "
              f"= {r['valu_issue_cycles']:7.1f} issue cycles; salu {r['salu']:3d} vmem {r['vmem']:2d} lds {r['lds']:2d}")
    if unknown:
        print("priced as half-rate by default:", unknown)


if __name__ == "__main__":
    main()
