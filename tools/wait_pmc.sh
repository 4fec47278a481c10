#!/bin/bash
# Where the waves of the scan kernel wait: one rocprofv3 --pmc pass per option variant (see tools/scan_pmc.sh).
#   tools/wait_pmc.sh OUTDIR "label1:--opt a=1" "label2:" ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for v in "$@"; do
  L=${v%%:*}; A=${v#*:}
  rm -rf $O/$L.w
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY -d $O/$L.w --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 $A > /dev/null 2> $O/$L.w.err
  python3 tools/pmc_kernel.py tq_scan_ $O/$L.w > $O/$L.wait.txt
  rm -rf $O/$L.w
  echo "== $L ($A)"; cat $O/$L.wait.txt
done
