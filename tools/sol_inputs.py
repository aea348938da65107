#!/usr/bin/env python3
"""Builds the speed-of-light inputs that bench.py reads (profiles/<tag>_sol_inputs.json) from files kept under profiles/:

  <tag>_pmc_passes.txt            per-kernel counter sums of ONE 64-spp step (tools/profile_all.sh), five passes
  <tag>_kernel_stats_single_lane.md   kernel trace of one single-lane step (kernels one after the other)
  <tag>_isa_mix.json              static VALU instruction mix of each kernel (this script writes it from build/isa/kl_kernels.s)
  r02_valu_issue.txt              measured issue cost of each VALU instruction (tools/micro/valu_issue.hip)

VALU issue time of a kernel = sum over its executed wave-instructions of the clocks each holds its SIMD's vector issue:
  2 clk  v_add/sub/mul/fma/fmac_f32, v_mov_b32, v_and/or/xor_b32, v_add/sub_u32, v_lshrrev_b32   (measured 2.2-2.6 at 6-8 waves/SIMD)
  8 clk  v_rcp/rsq/sqrt_f32 (and the other transcendentals)
  4 clk  everything else measured (compares, selects, min/max, conversions, shifts, packed fp32, DPP moves, ...); unmeasured
         opcodes are priced at 4
The hardware counts executed instructions by class: SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 (2 clk each), _TRANS_F32 (8), _CVT (4); the
remainder (integer, logic, moves, compares, selects, min/max) has no class counter that separates 2- from 4-clock opcodes, so
it is priced at the average cost of those opcodes in the kernel's own code (static mix, `rest_cost_static`).
Every kernel's figures are stamped with the hash of that kernel's machine code in the library the counter passes ran on
(`code`, printed by tools/pmc_workload.py into the header line of each pass): bench.py compares it with the library it
loads and reports no issue / traffic fraction for a kernel whose code has changed since.
usage: tools/sol_inputs.py <tag> [isa.s]     (<tag>_pmc_passes.txt names the configuration and the traversal instance)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWO = {"v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32",
       "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32"}
EIGHT = {"v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}
CLASSED = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32"} | EIGHT   # + v_cvt_*: have class counters
N_SIMD, PEAK_CLOCK_HZ = 1024, 2.4e9
L2_BYTES_PER_S, FABRIC_BYTES_PER_S = 34.5e12, 6.3e12      # L2 aggregate; HBM achievable (hardware guide)


def kernel_names(types):
    """(rocprof name patterns, mangled prefixes) of a step's kernels when the tree holds the primitive types `types`"""
    kernels = {"intersect": r"k_intersect_bvh<(0|false), %du>" % types, "shade": r"k_shade<1>", "generate": r"k_generate", "accumulate": r"k_accumulate"}
    mangled = {"intersect": "_ZN2kl15k_intersect_bvhILi0ELj%dEE" % types, "shade": "_ZN2kl7k_shadeILi1EE", "generate": "_ZN2kl10k_generate",
               "accumulate": "_ZN2kl12k_accumulate"}
    return kernels, mangled


def base(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def cost(op):
    return 2 if op in TWO else 8 if op in EIGHT else 4


def isa_mix(path, mangled):
    out = {}
    for key, prefix in mangled.items():
        on, ops = False, {}
        for l in open(path):
            if not on:
                on = l.startswith(prefix)
                continue
            t = l.strip().split()
            if not t:
                continue
            if t[0].startswith("v_"):
                ops[base(t[0])] = ops.get(base(t[0]), 0) + 1
            if t[0] == "s_endpgm":
                break
        rest = {o: n for o, n in ops.items() if o not in CLASSED and not o.startswith("v_cvt_")}
        n_rest = sum(rest.values())
        out[key] = {"static_valu": sum(ops.values()), "static_rest": n_rest,
                    "rest_cost_static": round(sum(cost(o) * n for o, n in rest.items()) / max(n_rest, 1), 4),
                    "all_cost_static": round(sum(cost(o) * n for o, n in ops.items()) / max(sum(ops.values()), 1), 4),
                    "opcodes": dict(sorted(ops.items(), key=lambda kv: -kv[1]))}
    return out


def parse_passes(path):
    """{pass: {"step": {...}, kernels: {name: {counter: (sum, calls)}}}}"""
    res, cur_pass, cur_k = {}, None, None
    for l in open(path):
        if l.startswith("### pass"):
            m = re.match(r"### pass (\w+): (.*)", l)
            cur_pass = m.group(1)
            try:
                step = json.loads(m.group(2))
            except Exception:
                step = {}
            res[cur_pass] = {"step": step, "kernels": {}}
        elif not l.startswith(" "):
            m = re.match(r"(.*?)\s+calls=(\d+)", l)
            cur_k = m.group(1)
            res[cur_pass]["kernels"][cur_k] = {"calls": int(m.group(2))}
        else:
            t = l.split()
            res[cur_pass]["kernels"][cur_k][t[0]] = float(t[1])
    return res


def pick(kernels, pattern):
    for k, v in kernels.items():
        if re.fullmatch(pattern, k):
            return v
    return None


def single_lane(path):
    out = {}
    for l in open(path):
        c = [x.strip() for x in l.split("|")]
        if len(c) > 6 and c[2].isdigit():
            out[c[1]] = {"calls": int(c[2]), "total_ms": float(c[3]), "avg_us": float(c[4])}
    return out


def main(tag, isa):
    prof = os.path.join(ROOT, "profiles")
    P = parse_passes(os.path.join(prof, tag + "_pmc_passes.txt"))
    step = P["mix"]["step"]
    types, config = step.get("traversal_types", 4), step.get("config", 3)
    KERNELS, MANGLED = kernel_names(types)
    mix = isa_mix(isa, MANGLED)
    json.dump(mix, open(os.path.join(prof, tag + "_isa_mix.json"), "w"), indent=1)
    SL = single_lane(os.path.join(prof, tag + "_kernel_stats_single_lane.md"))
    rays = step["rays"]
    paths = step["paths"]
    codes = [P[p]["step"].get("code") for p in P]
    if any(c != codes[0] for c in codes):
        sys.exit("the counter passes of %s ran on different builds of the kernels: %r" % (tag, codes))
    out = {"tag": tag, "config": config, "traversal_types": types,
           "workload": "one %d-spp step of BASELINE configs[%d] (tools/pmc_workload.py): %d paths, %d rays" % (step.get("spp", 64), config - 1, paths, rays),
           "rays_per_step": rays, "paths_per_step": paths, "n_simd": N_SIMD, "peak_clock_hz": PEAK_CLOCK_HZ,
           "l2_bytes_per_s": L2_BYTES_PER_S, "fabric_bytes_per_s": FABRIC_BYTES_PER_S,
           "code": codes[0] or {}, "kernels": {}}
    for key, pat in KERNELS.items():
        m, t, l2, wr, fe = (pick(P[p]["kernels"], pat) for p in ("mix", "time", "l2", "write", "fetch"))
        if not m:
            continue
        valu = m["SQ_INSTS_VALU"]
        two = m["SQ_INSTS_VALU_ADD_F32"] + m["SQ_INSTS_VALU_MUL_F32"] + m["SQ_INSTS_VALU_FMA_F32"]
        trans, cvt = m["SQ_INSTS_VALU_TRANS_F32"], m["SQ_INSTS_VALU_CVT"]
        rest = valu - two - trans - cvt
        c_rest = mix[key]["rest_cost_static"]
        issue = 2 * two + 8 * trans + 4 * cvt + c_rest * rest
        unit = paths if key in ("generate", "accumulate") else rays
        k = {"unit": "path" if unit == paths else "ray", "launches_per_step": m["calls"], "kernel": pat.replace("(0|false)", "0"),
             "code": (codes[0] or {}).get(key),
             "valu_insts": valu, "valu_2clk_float": two, "valu_trans": trans, "valu_cvt": cvt, "valu_rest": rest, "rest_cost_static": c_rest,
             "issue_clocks": issue, "issue_clocks_per_unit": issue / unit, "avg_clocks_per_valu_inst": issue / valu,
             "issue_clocks_if_rest_all_2clk": 2 * two + 8 * trans + 4 * cvt + 2 * rest,
             "issue_clocks_if_rest_all_4clk": 2 * two + 8 * trans + 4 * cvt + 4 * rest}
        if l2:
            req = l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]
            k.update({"l2_requests": req, "l2_hit_rate": l2["TCC_HIT_sum"] / max(req, 1), "l2_bytes_per_unit": req * 128 / unit,
                      "fabric_read_requests": l2["TCC_EA0_RDREQ_sum"], "fabric_read_requests_32B": l2["TCC_EA0_RDREQ_32B_sum"]})
            rd = (l2["TCC_EA0_RDREQ_sum"] - l2["TCC_EA0_RDREQ_32B_sum"]) * 128 + l2["TCC_EA0_RDREQ_32B_sum"] * 32
            wb = wr["WRITE_SIZE"] * 1024 if wr else 0.0
            k.update({"fabric_read_bytes": rd, "fabric_write_bytes": wb, "fabric_bytes_per_unit": (rd + wb) / unit,
                      "fabric_bytes_per_launch": (rd + wb) / m["calls"]})
            if fe:
                k["FETCH_SIZE_KB_as_reported"] = fe["FETCH_SIZE"]     # tallies a 128-byte request as 64 bytes on gfx950
        if t:
            # SQ_* time counters are in units of 4 clocks, summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            k.update({"wave_cycles_quads": t["SQ_WAVE_CYCLES"], "wait_any_share": t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"],
                      "wait_inst_share": t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"], "grbm_gui_active": t["GRBM_GUI_ACTIVE"],
                      "lds_insts": t["SQ_INSTS_LDS"], "vmem_rd_insts": t["SQ_INSTS_VMEM_RD"]})
        for name, v in SL.items():
            if re.fullmatch(pat, name):
                k.update({"single_lane_total_ms": v["total_ms"], "single_lane_launches": v["calls"]})
                if t:
                    k["clock_hz_under_counters"] = None
                k["valu_issue_ms_at_peak_clock"] = issue / N_SIMD / PEAK_CLOCK_HZ * 1e3
                k["valu_issue_share_of_single_lane_time"] = k["valu_issue_ms_at_peak_clock"] / v["total_ms"]
        out["kernels"][key] = k
    json.dump(out, open(os.path.join(prof, tag + "_sol_inputs.json"), "w"), indent=1)
    for key, k in out["kernels"].items():
        print("%-10s %.2f G VALU  avg %.2f clk  issue %.1f ms at 2.4 GHz  single-lane %.1f ms  -> %.2f   L2 hit %.2f  fabric %.0f B/%s" % (
            key, k["valu_insts"] / 1e9, k["avg_clocks_per_valu_inst"], k.get("valu_issue_ms_at_peak_clock", 0), k.get("single_lane_total_ms", 0),
            k.get("valu_issue_share_of_single_lane_time", 0), k.get("l2_hit_rate", 0), k.get("fabric_bytes_per_unit", 0), k["unit"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "build", "isa", "kl_kernels.s"))
