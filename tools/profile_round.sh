set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --no-cpu --steps 8 --warmup 2 > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f --output-format csv -- python3 bench.py --no-cpu --steps 4 --warmup 1 > /dev/null 2> $O/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w --output-format csv -- python3 bench.py --no-cpu --steps 4 --warmup 1 > /dev/null 2> $O/pmc_w.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d $O/sq1 --output-format csv -- python3 bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/sq2 --output-format csv -- python3 bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sq2.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM -d $O/sq3 --output-format csv -- python3 bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sq3.err || true
python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w --out $O/traffic_c3_sub.json --commit $(cat .build_commit) --copy-to $O/csv
python3 tools/pmc_busy.py $O/sq1 $O/sq2 $O/sq3 --out $O/pmc_busy.json --source "rocprofv3 --pmc passes of python3 bench.py --no-cpu (c3 sub)" --commit $(cat .build_commit)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
# drop the bulky raw traces, keep summaries
rm -rf $O/stats $O/pmc_f $O/pmc_w $O/sq1 $O/sq2 $O/sq3
python3 bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
ls -la $O
head -c 600 $O/kernel_stats.csv
