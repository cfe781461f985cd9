#!/bin/bash
# PMC passes (counters only with --kernel-trace; one counter group per run) on a short bench.
# GROUPS selects the passes (default: all); WL the workload.
WL=${WL:-c2}
GROUPS_=${GROUPS_:-"lds wait fetch write grbm"}
cd /tmp && export TMPDIR=/tmp
# genome-like workloads: the text comes from a file (their generators launch ~1e6 tiny kernels: rocprofv3 --pmc segfaults on them)
TXT=""
case $WL in g*) python3 /root/repo/tools/make_text.py $WL /tmp/pmc_text_$WL.npy && TXT="--text-file /tmp/pmc_text_$WL.npy" ;; esac
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /root/repo/gpurun_out/pmc_${WL}_$tag -- python3 /root/repo/bench.py --workload $WL $TXT --steps 1 --warmup 0 --prewarm-s 0 --no-cpu-baseline --no-host-path --no-verify > /root/repo/gpurun_out/pmc_${WL}_$tag.log 2>&1; tail -1 /root/repo/gpurun_out/pmc_${WL}_$tag.log | cut -c1-120; }
for g in $GROUPS_; do
  case $g in
    lds) run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS ;;
    wait) run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES ;;
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    grbm) run grbm GRBM_GUI_ACTIVE ;;
    tcp) run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum ;;
    tcc) run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
  esac
done
python3 /root/repo/tools/pmc_summary.py $(for g in $GROUPS_; do echo /root/repo/gpurun_out/pmc_${WL}_$g; done) > /root/repo/gpurun_out/pmc_${WL}_summary.txt 2>&1
