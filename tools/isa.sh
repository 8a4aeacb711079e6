#!/bin/bash
# tools/isa.sh <file.hip> [kernel-name-substring]: gfx950 assembly of one translation unit into /tmp/isa/<file>.s
# (and, with a second argument, the body of the first kernel whose mangled name contains it into /tmp/isa/k.s)
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1
B=$(basename "$F" .hip)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -S \
    -I"$R/end-to-end-asr-pytorch_amd/csrc" "$R/end-to-end-asr-pytorch_amd/csrc/$B.hip" -o /tmp/isa/$B.s || exit 1
grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):" /tmp/isa/$B.s | paste - - - - - > /tmp/isa/$B.meta
if [ -n "$2" ]; then
  awk -v pat="$2" '$0 ~ "^_Z.*"pat".*:$" && !on {on=1} on {print} on && /s_endpgm/ {exit}' /tmp/isa/$B.s > /tmp/isa/k.s
  wc -l /tmp/isa/k.s
fi
