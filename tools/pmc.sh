#!/bin/bash
# PMC passes (counters only with --kernel-trace; one counter group per run) on a short bench.
WL=${WL:-c2}
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /root/repo/gpurun_out/pmc_${WL}_$tag -- python3 /root/repo/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > /root/repo/gpurun_out/pmc_${WL}_$tag.log 2>&1; tail -1 /root/repo/gpurun_out/pmc_${WL}_$tag.log | cut -c1-200; }
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
ls /root/repo/gpurun_out/pmc_${WL}_lds/*/ | head
