"""CPU suite: the roofline record of bench.py and the code identity it rests on. Counter figures kept in files are used only
when the kernel's machine code in the loaded library hashes to the value the file was stamped with."""
import copy
import os
import sys
import types

import klrender as kl
from conftest import ROOT
from klrender import codeid

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_render_kernel_of_the_library_has_an_identity():
    h = codeid.kernel_hashes(kl.LIB_PATH)
    for types_ in (1, 4, 5, 7):
        ident = codeid.render_code_identity(kl.LIB_PATH, types_)
        assert set(ident) == {"intersect", "shade", "generate", "accumulate"}
        assert all(isinstance(v, str) and len(v) == 16 for v in ident.values()), ident
    # different instances are different code; the same file hashes the same twice
    assert codeid.pick(h, "_ZN2kl15k_intersect_bvhILi0ELj4EE") != codeid.pick(h, "_ZN2kl15k_intersect_bvhILi0ELj5EE")
    assert h == codeid.kernel_hashes(kl.LIB_PATH)
    assert codeid.pick(h, "_ZN2kl15k_intersect_bvh") is None          # ambiguous prefix: no identity


def _kernel_pass():
    return {"frames": 64, "rays": 575632860, "paths": 132710400, "intersect_ms": 44.7, "intersect_launches": 8, "shade_ms": 21.1, "shade_launches": 8,
            "generate_ms": 1.8, "accumulate_ms": 0.47, "render_ms": 68.2, "nodes_per_ray": 13.76, "tris_per_ray": 1.47, "spheres_per_ray": 0.0,
            "quads_per_ray": 0.0, "steps_per_ray": 15.23, "lane_active_frac": 0.85, "wave_steps": 161000000}


def _sol(code):
    def kern(unit, units, insts, clk, fabric):
        return {"unit": unit, "code": code[k], "valu_insts": insts, "issue_clocks": clk * units, "issue_clocks_per_unit": clk,
                "avg_clocks_per_valu_inst": clk * units / insts, "issue_clocks_if_rest_all_2clk": 0.72 * clk * units,
                "issue_clocks_if_rest_all_4clk": 1.1 * clk * units, "fabric_bytes_per_unit": fabric, "l2_bytes_per_unit": 600.0, "l2_hit_rate": 0.64}
    out = {"config": 3, "kernels": {}}
    for k, (unit, units, insts, clk, fabric) in {"intersect": ("ray", 575632860, 30.9e9, 166.8, 212.0), "shade": ("ray", 575632860, 11.3e9, 56.7, 119.0),
                                                 "generate": ("path", 132710400, 1e9, 20.0, 48.0), "accumulate": ("path", 132710400, 0.1e9, 2.0, 40.0)}.items():
        out["kernels"][k] = kern(unit, units, insts, clk, fabric)
    return out


def test_roofline_names_three_ceilings_and_the_one_that_binds():
    bvh = types.SimpleNamespace(flat_spheres=0, flat_quads=7, flat_tris=0)
    code = {"intersect": "a" * 16, "shade": "b" * 16, "generate": "c" * 16, "accumulate": "d" * 16}
    rl = bench.build_roofline(_kernel_pass(), bvh, _sol(code), code, 3, "x.json")
    # flops: (100 * 13.76 + 45 * 1.47) flop x rays / 44.7 ms / 157.3 TFLOP/s
    assert abs(rl["frac_fp32_flops"] - (1376 + 66.15) * 575632860 / 44.7e-3 / 157.3e12) < 2e-4
    assert abs(rl["frac_hbm_fabric"] - 212.0 * 575632860 / 44.7e-3 / 8e12) < 2e-4
    assert abs(rl["frac_valu_issue"] - 166.8 * 575632860 / 1024 / 2.4e9 / 44.7e-3) < 2e-4
    assert abs(rl["frac_valu_issue_if_all_2clk"] - 0.72 * rl["frac_valu_issue"]) < 2e-4
    assert rl["bound"] == "valu_issue" and rl["frac"] == rl["frac_valu_issue"] and rl["stale_sol_inputs"] is False
    assert rl["traffic"] == round(212.0 * 575632860 / 8) and rl["shade"]["frac_valu_issue"] > 0
    assert rl["hbm_algorithmic"]["ratio_to_hbm_peak"] > 1.0      # served by the caches: labelled, never `frac`


def test_a_touched_kernel_makes_the_line_say_stale():
    bvh = types.SimpleNamespace(flat_spheres=0, flat_quads=7, flat_tris=0)
    code = {"intersect": "a" * 16, "shade": "b" * 16, "generate": "c" * 16, "accumulate": "d" * 16}
    now = dict(code, intersect="e" * 16)                          # the traversal kernel was rebuilt from changed source
    rl = bench.build_roofline(_kernel_pass(), bvh, _sol(code), now, 3, "x.json")
    assert rl["stale_sol_inputs"] is True and rl["frac_valu_issue"] is None and rl["frac_hbm_fabric"] is None and rl["traffic"] is None
    assert rl["bound"] == "fp32_valu_flops" and rl["frac"] == rl["frac_fp32_flops"]     # what is still measured live
    assert rl["shade"]["frac_valu_issue"] > 0 and "valu_issue_ms_all_kernels" not in rl  # shade's own figures stay valid
    now = dict(code, shade="f" * 16)
    rl = bench.build_roofline(_kernel_pass(), bvh, _sol(code), now, 3, "x.json")
    assert rl["stale_sol_inputs"] is False and rl["bound"] == "valu_issue" and rl["shade"] == {"kernel": "k_shade", "measured_ms": 21.1, "stale_sol_inputs": True}
    # figures of another configuration, or no file at all: only the live ceiling
    other = copy.deepcopy(_sol(code)); other["config"] = 5
    for sol in (other, None):
        rl = bench.build_roofline(_kernel_pass(), bvh, sol, code, 3, "x.json")
        assert rl["bound"] == "fp32_valu_flops" and rl["stale_sol_inputs"] is None and rl["sol_inputs"] is None


def test_the_kept_counter_files_name_the_code_they_were_taken_on():
    """every speed-of-light file that bench.py may read carries a hash per kernel"""
    import json
    for cfg, name in bench.SOL_INPUTS.items():
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        sol = json.load(open(path))
        assert sol["config"] == cfg
        for k, v in sol["kernels"].items():
            assert isinstance(v.get("code"), str) and len(v["code"]) == 16, (name, k)
