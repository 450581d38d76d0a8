#!/bin/bash
# registers, scratch, occupancy and LDS of every kernel of this build (clang's kernel-resource-usage remarks)
#   bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/../sift_amd/csrc" || exit 1
for f in kernels_*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1
done | python3 -c "
import re, subprocess, sys
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r'remark: Function Name: (\S+)', line)
    if m:
        cur = [m.group(1), {}]; rows.append(cur); continue
    m = re.search(r'remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur: cur[1][m.group(1)] = m.group(2)
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for (mangled, vals), name in zip(rows, names):
    short = re.sub(r'\(.*', '', name.replace('(anonymous namespace)::', '')).replace('sift_hip::', '').replace('void ', '')
    print(short, ' '.join(f'{k}: {v}' for k, v in vals.items()))
"
