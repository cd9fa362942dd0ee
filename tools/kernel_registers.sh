#!/bin/bash
# Register counts of the kernels of one source file (no GPU needed): VGPRs decide the wavefronts per SIMD (512 / count),
# a project_and_bin beyond 64 loses its second 1,024-thread workgroup per CU.
#   bash tools/kernel_registers.sh vtgs_binning.hip [name filter]
F=${1:-vtgs_binning.hip}; PAT=${2:-.}
T=$(mktemp -d); cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -save-temps=obj -o x.o /root/repo/vtgaussian-slam_amd/csrc/$F 2>/dev/null
grep -E "\.name:|\.sgpr_count:|\.vgpr_count:|\.private_segment_fixed_size:|\.group_segment_fixed_size:" *.s | python3 -c "
import sys, re
cur = {}
for line in sys.stdin:
    k, v = line.strip().split(':', 1)
    k = k.strip().lstrip('.- '); cur[k] = v.strip()
    if k == 'vgpr_count' and re.search(sys.argv[1], cur.get('name', '')):
        print('%-90s sgpr %4s vgpr %4s lds %6s scratch %s' % (cur.get('name', '')[:90], cur.get('sgpr_count'), cur.get('vgpr_count'), cur.get('group_segment_fixed_size'), cur.get('private_segment_fixed_size')))
" "$PAT"
rm -rf $T
