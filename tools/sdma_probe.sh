#!/bin/bash
# which engine moves page-locked memory, and what it costs the kernels beside it:  bash tools/sdma_probe.sh
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python3 tools/host_trace.py 3 main 2>&1 | grep "ms per step"; }
run X=1
run GPU_FORCE_BLIT_COPY_SIZE=0
run HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0
run DEBUG_CLR_LIMIT_BLIT_WG=16
run DEBUG_CLR_LIMIT_BLIT_WG=4
run HSA_ENABLE_SDMA=0
