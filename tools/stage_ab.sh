#!/bin/bash
# pageable host loop with the staged chunks moved by library kernels (default) / by hipMemcpyAsync, then the example loop
timeout 300 python3 tools/host_rate.py 2>&1 | grep -v amdgpu.ids | tail -7
export LINES_SHOWN=0; N=${N:-200} bash tools/example_loop.sh | tail -1
