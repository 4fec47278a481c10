#!/bin/bash
# Row f4 (one record per taxon and lane-step): the upper bound of its gain on the load side.  A/B build with the two nibble
# loads of a wave's own rows dropped (-DTQ_DIAG_NO_NIB: pattern codes made up from plane words, results wrong, count masks
# and therefore the walk unchanged) against the product library: time, SQ / LDS counters, texture-address busy.
#   tools/f4_bound.sh OUTDIR     (expects tools/ab/libtetrad_nonib.so)
cd $GRAFT_REPO_ROOT
O=$1
mkdir -p $O
bash tools/scan_pmc.sh $O "base:" > $O/log.txt 2>&1
TQ_LIB_PATH=tools/ab/libtetrad_nonib.so bash tools/scan_pmc.sh $O "nonib:" >> $O/log.txt 2>&1
for L in base nonib; do
  if [ $L = nonib ]; then export TQ_LIB_PATH=tools/ab/libtetrad_nonib.so; fi
  rm -rf $O/$L.m
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $O/$L.m --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/$L.m.err
  python3 tools/pmc_kernel.py tq_scan_ $O/$L.m >> $O/$L.txt
  rm -rf $O/$L.m
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $O/$L.m --output-format csv -- python3 bench.py --no-other-mode --phases 1 --no-cpu --steps 3 --warmup 1 > /dev/null 2>> $O/$L.m.err
  python3 tools/pmc_kernel.py tq_scan_ $O/$L.m >> $O/$L.txt
  rm -rf $O/$L.m
done
cat $O/base.txt $O/nonib.txt
