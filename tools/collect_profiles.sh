#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench command and the
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# Output goes to gpurun_out/<round>_*; tools/summarize_profiles.py turns it into profiles/.
set -e
ROUND=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${ROUND}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${ROUND}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${ROUND}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${ROUND}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${ROUND}_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 || true
cd $R && python3 bench.py > gpurun_out/${ROUND}_bench.json 2> /dev/null
echo collected
