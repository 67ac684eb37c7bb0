#!/bin/bash
# Tuning aid: ISA of the step kernels alone (the out-of-line paths become external calls), ~10 s instead of the full build's minutes.
#   tools/inspect_step.sh [extra hipcc flags]  ->  /tmp/dis/step.s, /tmp/dis/step_<TRACE><FP>.s
set -e
mkdir -p /tmp/dis
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -mllvm -structurizecfg-skip-uniform-regions -fgpu-rdc --cuda-device-only -S \
  -DWSP_INSPECT "$@" -o /tmp/dis/step.ll "$(dirname "$0")/../whisper_amd/csrc/wsp_lib.hip" 2>&1 | grep -v "hip-link" || true
/opt/rocm/lib/llvm/bin/clang -x ir -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -mllvm -structurizecfg-skip-uniform-regions -S /tmp/dis/step.ll -o /tmp/dis/step.s
for k in 11 10 01 00; do
  awk -v n="_Z15wsp_step_kernelILb${k:0:1}ELb${k:1:1}EEv8DevStatePKS0_mj" '$0 ~ "\\.type[ \t]+" n ",@function" {p=1} p {print} p && /\.end_amdhsa_kernel/ {exit}' /tmp/dis/step.s > /tmp/dis/step_$k.s
done
grep -E "^\s+\.amdhsa_next_free_vgpr|private_segment_fixed_size" /tmp/dis/step_01.s | head -4
wc -l /tmp/dis/step_??.s
