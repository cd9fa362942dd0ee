#!/bin/bash
# Register / LDS / scratch use of every kernel in one source file, from the compiler's own remarks (no GPU needed):
#   tools/kernel_resources.sh vtgaussian-slam_amd/csrc/vtgs_composite.hip [-DFOO ...]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/remark: Function Name:/ {name=$(NF-1)}
       /remark: +TotalSGPRs:/ {s=$(NF-1)} /remark: +VGPRs:/ {v=$(NF-1)} /remark: +AGPRs:/ {a=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
       /Occupancy/ {o=$(NF-1)} /LDS Size/ {printf "%s vgpr %s agpr %s sgpr %s scratch %s occ %s lds %s\n", name, v, a, s, sc, o, $(NF-1)}' |
  while read -r n rest; do printf "%-72s %s\n" "$(echo "$n" | c++filt | sed 's/(.*//' | cut -c1-72)" "$rest"; done
