#!/bin/bash
# VGPRs / occupancy of the search kernels of sf_icp.hip as the compiler reports them:  tools/kernel_regs.sh [extra hipcc flags]
cd "$(dirname "$0")/../slam_sensor_fusion_amd/csrc"
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I. -Rpass-analysis=kernel-resource-usage -c sf_icp.hip -o /tmp/kernel_regs.o 2>&1 |
  python3 -c "
import sys, re
name = None
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: name = m.group(1); d = {}
    for key in ('VGPRs', 'Occupancy \[waves/SIMD\]', 'SGPRs Spill', 'ScratchSize \[bytes/lane\]', 'LDS Size \[bytes/block\]'):
        m = re.search(key + r': (\d+)', l)
        if m and name: d[key] = m.group(1)
    if name and 'LDS Size' in l and any(k in name for k in ('k_nn_red', 'k_ref_nn', 'k_ref_fused', 'k_bf', 'k_map_nn', 'k_icp_fused')):
        short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:28]
        print(short, 'vgpr', d.get('VGPRs'), 'occ', d.get('Occupancy \[waves/SIMD\]'), 'sgpr-spill', d.get('SGPRs Spill'), 'scratch', d.get('ScratchSize \[bytes/lane\]'), 'lds', d.get('LDS Size \[bytes/block\]'))
"
