#!/bin/bash
# Texture-path (TA / TCP / TD) counters of tq_scan_wg_kernel for engine-option variants: is the load phase bound by the
# address path, the L1, or the L2 round trip?   tools/mem_pmc.sh OUTDIR "label:args" ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for v in "$@"; do
  L=${v%%:*}; A=${v#*:}
  : > $O/$L.mem.txt
  # (two counters of a block per pass: four TA counters at once abort rocprofv3 with "exceeds the capabilities of the hardware",
  # and the aborted profiler then hangs -- hence the timeout)
  for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "TD_TD_BUSY_sum TD_TC_STALL_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VMEM"; do
    rm -rf $O/$L.m
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $O/$L.m --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 $A > /dev/null 2> $O/$L.m.err || echo "pass failed: $set" >> $O/$L.mem.txt
    python3 tools/pmc_kernel.py tq_scan_ $O/$L.m >> $O/$L.mem.txt
    rm -rf $O/$L.m
  done
  echo "== $L ($A)"; cat $O/$L.mem.txt
done
