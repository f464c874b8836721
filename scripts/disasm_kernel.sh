#!/bin/bash
# disasm_kernel.sh <object.o> <mangled-name-prefix> [out.s]: the gfx950 ISA of one kernel of a HIP object (+ its register / scratch use)
set -e
obj=$1; pat=$2; out=${3:-/tmp/dis/kernel.s}
L=/opt/rocm/lib/llvm/bin; mkdir -p /tmp/dis
$L/llvm-objcopy --dump-section .hip_fatbin=/tmp/dis/fat.bin $obj
$L/clang-offload-bundler --unbundle --input=/tmp/dis/fat.bin --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/dis/dev.co
$L/llvm-objdump -d /tmp/dis/dev.co > /tmp/dis/dev.s
awk -v pat="<$pat" 'index($0, pat) && /^[0-9a-f]+ </ {on=1; print; next} on && /^[0-9a-f]+ </ {exit} on {print}' /tmp/dis/dev.s > $out
echo "$(wc -l < $out) lines, fp64 VALU: $(grep -c 'v_.*_f64' $out), ds_read: $(grep -c 'ds_read' $out), scratch: $(grep -c 'scratch_' $out)"
$L/llvm-readelf --notes /tmp/dis/dev.co | grep -A30 "$pat" | grep -E "vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill" | head -4
