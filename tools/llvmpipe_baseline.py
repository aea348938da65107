#!/usr/bin/env python3
"""Times the reference's own shaders on Mesa llvmpipe (oracle/_ref/glsl_ref --timing) for BASELINE configs 1 and 2 and writes
profiles/llvmpipe_baseline.json. Runs only where the reference checkout is (the build container): the reference cannot travel
to the GPU box in any form, so this container's figure is the kept record of the reference-shader CPU baseline.
Declared deviations of the harness (oracle/ref_harness.c): RGBA32F attachments, `vec3 col = vec3(0)`, ping-pong accumulation."""
import json
import os
import platform
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "glsl_ref")
REF = os.environ.get("KL_REFERENCE", "/root/reference")
TMP = os.environ.get("TMPDIR", "/tmp")


def cpu_model():
    for l in open("/proc/cpuinfo"):
        if l.startswith("model name"):
            return l.split(":", 1)[1].strip()
    return platform.processor()


def run(w, h, mode, frames, tex, repeat=3):
    best, gl = None, []
    for _ in range(repeat):
        cmd = [BIN, "--shaders", os.path.join(REF, "resources", "Shaders"), "--w", str(w), "--h", str(h), "--mode", str(mode), "--frames", str(frames),
               "--zero-col", "--timing", "--hdri", tex["hdri"][0], *map(str, tex["hdri"][1:]), "--albedo", tex["albedo"][0], *map(str, tex["albedo"][1:])]
        res = subprocess.run(cmd, check=True, capture_output=True, text=True)
        info = json.loads([l for l in (res.stdout + res.stderr).splitlines() if l.startswith("{")][-1])
        gl = [l for l in res.stderr.splitlines() if l.startswith("GL_")]
        if best is None or info["seconds"] < best["seconds"]:
            best = info
    return best, gl


def main():
    if not os.path.exists(BIN) or not os.path.isdir(REF):
        sys.exit("needs oracle/_ref/glsl_ref and the reference checkout")
    stb = os.path.join(ROOT, "oracle", "_ref", "stb_decode")
    tex = {}
    for key, name in (("hdri", "Background.png"), ("albedo", "Texture.png")):
        raw = os.path.join(TMP, "kl_base_%s.raw" % key)
        w, h, c = map(int, subprocess.run([stb, os.path.join(REF, "resources", name), raw], check=True, stdout=subprocess.PIPE, text=True).stdout.split())
        tex[key] = (raw, w, h, c)
    out = {"what": "reference GLSL shaders (resources/Shaders/*.glsl, read from the checkout) on Mesa llvmpipe through the DRI swrast driver",
           "harness": "oracle/ref_harness.c --timing: one untimed draw (shader JIT), then the frames bracketed by glFinish and CLOCK_MONOTONIC; best of 3 runs",
           "declared_patches": ["RGBA32F colour attachments (the application's unsized GL_RGBA is RGBA8)", "vec3 col = vec3(0) (as shipped AO renders all-zero on llvmpipe)",
                                "ping-pong accumulation textures", "MESA_GL_VERSION_OVERRIDE=4.6 MESA_GLSL_VERSION_OVERRIDE=460"],
           "host": {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cpu": cpu_model(),
                    "LP_NUM_THREADS": os.environ.get("LP_NUM_THREADS", "unset (llvmpipe uses every core, at most its built-in LP_MAX_THREADS)")},
           "textures": "reference Background.png (4096x2048 RGBA8) and Texture.png (1920x1080 RGB8, stb_image-decoded)", "runs": {}}
    for name, (w, h, mode, frames) in {"config1_pt_640x360_16frames": (640, 360, 1, 16), "pt_1920x1080_4frames": (1920, 1080, 1, 4),
                                       "config2_zdepth_1920x1080": (1920, 1080, 2, 4), "config2_ao_1920x1080": (1920, 1080, 0, 4)}.items():
        info, gl = run(w, h, mode, frames, tex)
        out["runs"][name] = info
        out["gl"] = gl
    json.dump(out, open(os.path.join(ROOT, "profiles", "llvmpipe_baseline.json"), "w"), indent=1)
    print(json.dumps(out["runs"], indent=1))


if __name__ == "__main__":
    main()
