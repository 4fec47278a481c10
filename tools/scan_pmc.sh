#!/bin/bash
# PMC counters of the scan kernel (tq_scan_wg_kernel / tq_scan_pb_kernel) for a list of engine-option variants (A/B of scan-kernel changes).
#   tools/scan_pmc.sh OUTDIR "label1:--opt a=1 --opt b=2" "label2:" ...
# Two rocprofv3 --pmc passes per variant on `python3 bench.py --phases 1 --no-cpu --steps 3 --warmup 1` (c3, subsample mode).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for v in "$@"; do
  L=${v%%:*}; A=${v#*:}
  rm -rf $O/$L.p1 $O/$L.p2
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM -d $O/$L.p1 --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 $A > /dev/null 2> $O/$L.p1.err
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $O/$L.p2 --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 $A > /dev/null 2> $O/$L.p2.err || \
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/$L.p2 --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 $A > /dev/null 2> $O/$L.p2.err
  python3 bench.py --phases 1 --no-cpu --steps 5 --warmup 2 $A 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L scan_ms', d['roofline']['kernel_ms'])" > $O/$L.txt
  python3 tools/pmc_kernel.py tq_scan_ $O/$L.p1 $O/$L.p2 >> $O/$L.txt
  rm -rf $O/$L.p1 $O/$L.p2
  echo "== $L ($A)"; cat $O/$L.txt
done
