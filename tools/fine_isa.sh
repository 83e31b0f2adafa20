#!/bin/bash
# Dumps the ISA of kernels_fine.hip (device only) to /tmp/asm/fine.s and prints register / spill / scratch use per kernel.
# usage: tools/fine_isa.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/../jello_amd/csrc"
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include "$@" -S --cuda-device-only kernels_fine.hip -o /tmp/asm/fine.s 2>/dev/null
python3 - <<'PY'
import re
t = open('/tmp/asm/fine.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
    pass
# simpler: walk the metadata block
names = re.findall(r'\.name:\s+(_Z\S+)', t)
blocks = t.split('  - .agpr_count:')[1:]
for b in blocks:
    n = re.search(r'\.name:\s+(\S+)', b).group(1)
    tmpl = re.search(r'k_fine_areaILi(\d+)ELb(\d)ELb(\d)', n)
    tag = 'AA=%s clips=%s paints=%s' % tmpl.groups() if tmpl else n
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    print('%-28s vgpr %3s sgpr %3s spill %2s scratch %4s lds %6s' % (tag, g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
PY
