#!/bin/bash
# kernel trace of one context running the bench batch with the level chain from octave $1 (default 2), mode $2 (default 1);
# FILTER=<regex> picks the kernels printed (default: the blur family), SIFT_SET="name=value,..." sets library options
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/chain_prof
CHAIN_FROM=${1:-2} CHAIN_MODE=${2:-1} rocprofv3 --kernel-trace -d /tmp/chain_prof -o ct --output-format csv -- python3 $R/tools/chain_trace.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/chain_prof/**/*kernel_trace.csv', recursive=True)[0]
rows = [(r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void sift_hip::', '')[:60], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda t: t[1])
# last batch: from the last widen/first blur
idx = [i for i, r in enumerate(rows) if 'blur_stream_kernel<5, false' in r[0]]
t0 = rows[idx[-1]][1]
for name, a, b in rows[idx[-1]:]:
    import os, re
    if re.search(os.environ.get('FILTER', 'blur|w16'), name):
        print(f"{name:62s} {(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f}")
PY
