for k in "group" "gather" "pipeline_parity_kept or pipeline_parity_bench" "u8 or cli or cpp"; do
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "$k" > gpurun_out/crash_$RANDOM.log 2>&1; echo "[$k] exit $?"
done
grep -l "double free\|Fatal\|core" gpurun_out/crash_*.log | while read f; do echo "== $f"; tail -40 $f; done
