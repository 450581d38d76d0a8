#!/bin/bash
# Everything profiles/ keeps for a round, from one box: kernel stats + blur launches of the default (pipelined) bench, timelines,
# per-kernel HBM traffic / L2 hit rate, SQ counters of the big non-blur kernels.   bash tools/profile_round.sh r02
set -u
export TMPDIR=/tmp
R=${1:-r03}
O=gpurun_out/$R
rm -rf $O; mkdir -p $O
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $O/device.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $O/device.txt
# 1. default bench under the kernel trace (same command as the stats in profiles/README.md)
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_profiled.json 2> /dev/null
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/roofline_from_stats.py --stats $O/kernel_stats.csv > $O/roofline.txt
python3 tools/roofline_from_stats.py --trace "$t" --dump $O/blur_launches.csv >> $O/roofline.txt
python3 tools/timeline_window.py "$t" 8000 12500 > $O/timeline_pipelined.txt 2>&1   # (the last ~11 ms of the run are the three stand-alone steps)
# 2. one step at a time: per-kernel timeline
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > /dev/null 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py "$t" > $O/timeline_single_step.txt 2>&1
rm -rf gpurun_out/prof
# 3. unprofiled bench lines
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> /dev/null
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --pipeline-depth 1 > $O/bench_depth1.json 2> /dev/null
timeout 600 python3 bench.py --workload config3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_config3.json 2> /dev/null
timeout 600 python3 bench.py --workload config5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_config5.json 2> /dev/null
for try in 1 2 3; do   # (the first RCCL start of a box sometimes ends without a line: once more then)
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --rccl-loopback > $O/bench_rccl_loopback.json 2> $O/bench_rccl_loopback.err
  [ -s $O/bench_rccl_loopback.json ] && break
  echo "attempt $try: exit $?" >> $O/bench_rccl_loopback.err; sleep 5
done
# 3b. the host-buffer loop: kernels and copies of three batches in flight
rm -rf gpurun_out/htrace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/htrace -- python3 tools/host_trace.py 3 main > $O/host_trace.txt 2>/dev/null
python3 tools/host_trace.py --report gpurun_out/htrace >> $O/host_trace.txt 2>&1
rm -rf gpurun_out/htrace
# 4. counters
bash tools/pmc_kernel.sh > $O/pmc_hbm_traffic.txt 2>&1
DBGS="" bash tools/desc_probe.sh 2>&1 | grep -vE "^dbg" > $O/pmc_sq_counters.txt
rm -rf gpurun_out/pmck_* gpurun_out/pmc_d1 gpurun_out/pmc_d2
# 5. round 4: the practical HBM ceiling, the N > 1 per-rank step on one GPU
./tools/probe/hbm_copy_probe > $O/hbm_copy.txt 2>&1
./tools/probe/valu_rate_probe > $O/valu_rate.txt 2>&1   # (built here: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off, see the file's header)
bash tools/loopback_ab.sh > $O/loopback_ab.txt 2>&1
python3 tools/gather_probe.py 40 2>&1 | grep "ms/step" > $O/gather_probe.txt
# 6. round 6: the streaming blur kernels alone (old against new packing, bit for bit); registers / scratch / occupancy of every kernel
# come from `bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt` in the build container (hipcc's remarks need no GPU)
[ -x tools/probe/blur_probe ] && timeout 300 ./tools/probe/blur_probe > $O/blur_probe.txt 2>&1
bash tools/single_trace.sh > $O/single_trace.txt 2>&1
rm -rf gpurun_out/prof_d1 gpurun_out/prof_d2
cat $O/roofline.txt; python3 -c "
import json
for n in ('bench','bench_depth1','bench_config3','bench_config5'):
    try:
        d=json.load(open('$O/'+n+'.json')); print(n, round(d['ms_per_step'],3),'ms/step', round(d['value']/1e6,1),'Mkp/s frac', round(d['roofline']['frac'],3))
    except Exception as e: print(n, 'failed', e)
"
