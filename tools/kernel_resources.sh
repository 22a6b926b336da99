#!/bin/bash
# Per-kernel VGPR / scratch usage of one source file (compile-time remark, no GPU needed).
# Usage: tools/kernel_resources.sh rp_joints [rp_solver ...]
cd "$(dirname "$0")/../rapier_amd/csrc"
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2>&1 |
    awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /ScratchSize/ {printf "%-48s vgprs %4s scratch %6s B/lane\n", name, v, $5}'
done
