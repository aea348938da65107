#!/bin/bash
# MI355X figures for the configurations the reference shader itself can run (BASELINE configs 1 and 2, built-in scene),
# next to profiles/llvmpipe_baseline.json: kl_bench over the C ABI, library defaults, real textures.
set -e
cd "$GRAFT_REPO_ROOT"
B=path-tracer_amd/bin/kl_bench
python3 - <<'P'
import numpy as np
np.load("tests/golden/assets/Texture_stb_rgb8.npz")["rgb"].tofile("/tmp/kl_albedo.raw")
P
T="--hdri tests/golden/assets/Background.png --albedo-raw /tmp/kl_albedo.raw 1920 1080 3"
{
timeout -k 10 60 $B --scene builtin --mode pt --w 640 --h 360 --frames 16 --warmup 2 $T
timeout -k 10 60 $B --scene builtin --mode pt --w 1920 --h 1080 --frames 64 --warmup 1 $T
timeout -k 10 60 $B --scene builtin --mode z --w 1920 --h 1080 --frames 64 --warmup 1 $T
timeout -k 10 60 $B --scene builtin --mode ao --w 1920 --h 1080 --frames 64 --warmup 1 $T
} > gpurun_out/r02_builtin_mi355x.jsonl
cat gpurun_out/r02_builtin_mi355x.jsonl
