#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the bench command per workload and the PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate passes, MFMA-busy in a third: MI355X_MICROARCH.md, rocprofv3 PMC slots; never
# combined with a trace domain other than --kernel-trace).  Output goes to gpurun_out/<round>_*;
# tools/summarize_profiles.py turns it into profiles/.
ROUND=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
run() {  # tag, bench args...
  local tag=$1; shift
  echo "$@" > $O/${ROUND}_${tag}_args.txt   # tools/summarize_profiles.py records --points of the PMC passes in traffic.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${ROUND}_${tag}_stats -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-voigt --no-extras > $O/${ROUND}_${tag}_bench_under_rocprof.json 2> /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${ROUND}_${tag}_fetch -o p -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-voigt --no-extras > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${ROUND}_${tag}_write -o p -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-voigt --no-extras > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${ROUND}_${tag}_mfma -o p -- python3 $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-voigt --no-extras > /dev/null 2>&1 || true
}
# C2: --opt 9=0 = MOM_OPT_OVERLAP off: under the kernel trace the two problem sizes run one after the other, so that the
# per-kernel durations are those of bench.py's roofline block (which takes them from serialized steps as well)
run C2 --workload C2 --opt 9=0
run C4 --workload C4 --points 256 --steps 2   # the size of the default bench line's C4 leg
run C1 --workload C1
run C5 --workload C5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${ROUND}_voigt_stats -o p -- python3 $R/bench_voigt.py > $O/${ROUND}_bench_voigt.json 2> /dev/null
cd $R
python3 bench.py > $O/${ROUND}_bench.json 2> /dev/null
python3 bench.py --workload C4 --points 1000 --steps 2 > $O/${ROUND}_bench_C4.json 2> /dev/null
python3 bench.py --workload C1 > $O/${ROUND}_bench_C1.json 2> /dev/null
python3 bench.py --workload C5 > $O/${ROUND}_bench_C5.json 2> /dev/null
python3 bench.py --workload C3 --no-cpu-baseline > $O/${ROUND}_bench_C3.json 2> /dev/null
python3 bench.py --gpus 2 --backend gloo --share-device --no-voigt > $O/${ROUND}_bench_2ranks_one_gpu.json 2> /dev/null
python3 bench.py --workload C3 --scaling strong --gpus 2 --backend gloo --share-device --no-cpu-baseline > $O/${ROUND}_bench_C3_strong_2ranks_one_gpu.json 2> /dev/null
python3 bench.py --workload C5 --gpus 2 --backend gloo --share-device --no-cpu-baseline > $O/${ROUND}_bench_C5_2ranks_one_gpu.json 2> /dev/null
# two ranks over RCCL through torch.distributed (the C5 timing all-reduce of bench.py: ADVICE r4) -- needs two GPUs
if [ "$(python3 -c 'import torch; print(torch.cuda.device_count())')" -ge 2 ]; then
  python3 bench.py --workload C5 --gpus 2 --backend torch --no-cpu-baseline > $O/${ROUND}_bench_C5_2gpus_torch.json 2> /dev/null
  python3 bench.py --gpus 2 --no-voigt --no-cpu-baseline > $O/${ROUND}_bench_2gpus_rccl.json 2> /dev/null
fi
python3 tools/bench_rrs_nt2.py > $O/${ROUND}_rrs_nt.txt 2> /dev/null
# the multi-tile RRS kernels (workgroup per pair above N = 16) under the kernel trace: per-kernel durations of the same six scenes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${ROUND}_rrs_nt_stats -o p -- python3 $R/tools/bench_rrs_nt2.py > /dev/null 2>&1
python3 tools/size_sweep.py > $O/${ROUND}_size_sweep.txt 2> /dev/null
python3 tools/f32_vs_f64.py > $O/${ROUND}_f32_vs_f64.txt 2> /dev/null
echo collected
