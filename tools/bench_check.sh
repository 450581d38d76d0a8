export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "bench exit $?"; python3 -c "
import json; d=json.load(open('gpurun_out/bench_full.json'))
print('ms/step',d['ms_per_step'],'value',d['value'],'frac',d['roofline']['frac'])
print('traffic',d['roofline']['traffic'],d['roofline']['traffic_source'])
print('host_inclusive',d.get('host_inclusive'))
print('single_frame_ms',d.get('single_frame_ms'))
print('cpu',{k:v for k,v in d['cpu_baseline'].items() if k!='reference_binary'})
"; tail -3 gpurun_out/bench_full.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --backend gloo --share-device --frames 8 > gpurun_out/bench_n2.json 2> gpurun_out/bench_n2.err; echo "n2 exit $?"; tail -c 1500 gpurun_out/bench_n2.json; tail -3 gpurun_out/bench_n2.err
