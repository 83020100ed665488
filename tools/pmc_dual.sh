# PMC passes over a SHORT Dual run (two layers of the C2 operator shape: a few hundred dispatches; counter collection serialises
# every dispatch, a full 40-layer run does not finish in minutes).  Run on the GPU box: bash tools/pmc_dual.sh
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6_dual
ARGS="2000 3 1 3 33 2"
timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o p -- python3 /root/repo/tools/dual_bench.py $ARGS > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 /root/repo/tools/dual_bench.py $ARGS > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 /root/repo/tools/dual_bench.py $ARGS > /dev/null 2>&1
ls $O/pmc_sq $O/pmc_fetch | head
