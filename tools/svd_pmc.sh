#!/bin/bash
# PMC counters of the singular-value kernels (tq_bidiag / tq_bdsqr / tq_score) on the default bench command.
#   tools/svd_pmc.sh OUTDIR LABEL [bench args]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; L=$2; shift; shift
mkdir -p $O
rm -rf $O/$L.p1 $O/$L.p2
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM -d $O/$L.p1 --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 3 --warmup 1 "$@" > /dev/null 2> $O/$L.p1.err
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/$L.p2 --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 3 --warmup 1 "$@" > /dev/null 2> $O/$L.p2.err
for k in tq_bidiag tq_bdsqr tq_score; do echo "== $L $k"; python3 tools/pmc_kernel.py $k $O/$L.p1 $O/$L.p2; done > $O/$L.txt
rm -rf $O/$L.p1 $O/$L.p2
cat $O/$L.txt
