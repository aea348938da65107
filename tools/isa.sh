#!/bin/bash
# usage: tools/isa.sh [extra -D flags]: device ISA of kl_kernels.hip into build/isa/kl_kernels.s, plus per-kernel register use
mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude -Ipath-tracer_amd/csrc -S --cuda-device-only "$@" -o build/isa/kl_kernels.s path-tracer_amd/csrc/kl_kernels.hip 2>&1 | grep -v "hip-link"
grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):" build/isa/kl_kernels.s | paste - - - - - | sed 's/\s\+/ /g'
