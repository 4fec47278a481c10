# The command list behind profiles/rNN_final/ (round 4: profiles/r04_final/) (run on the GPU box through gpurun; every rocprofv3 pass under a timeout:
# a profiler that aborts on an over-subscribed counter set hangs afterwards).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/r4prof}
rm -rf $O; mkdir -p $O
P="timeout -k 5 240 rocprofv3"
$P --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 8 --warmup 2 > $O/bench_under_rocprof.json 2> $O/stats.err
echo stats done
$P --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 4 --warmup 1 > /dev/null 2> $O/pmc_f.err
$P --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 4 --warmup 1 > /dev/null 2> $O/pmc_w.err
echo traffic done
$P --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM -d $O/sq1 --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sq1.err
$P --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/sq2 --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sq2.err
$P --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $O/sq3 --output-format csv -- python3 bench.py --no-other-mode --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sq3.err || true
echo pmc done
python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w --out $O/traffic_c3_sub.json --commit $(cat .build_commit) --copy-to $O/csv
python3 tools/pmc_busy.py $O/sq1 $O/sq2 $O/sq3 --out $O/pmc_busy.json --source "rocprofv3 --pmc passes of python3 bench.py --no-cpu (c3 sub)" --commit $(cat .build_commit)
cp $O/pmc_busy.json profiles/pmc_busy_latest.json; cp $O/traffic_c3_sub.json profiles/traffic_c3_sub.json   # what bench.py quotes
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
# the same for full mode (subsample_snps=False, the reference's default: tq_scan_dp_kernel)
$P --kernel-trace --stats -d $O/stats_full --output-format csv -- python3 bench.py --full --no-other-mode --no-cpu --steps 8 --warmup 2 > $O/bench_full_under_rocprof.json 2> $O/stats_full.err
find $O/stats_full -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_full.csv
$P --kernel-trace --pmc FETCH_SIZE -d $O/pmc_ff --output-format csv -- python3 bench.py --full --no-other-mode --no-cpu --steps 4 --warmup 1 > /dev/null 2> $O/pmc_ff.err
$P --kernel-trace --pmc WRITE_SIZE -d $O/pmc_fw --output-format csv -- python3 bench.py --full --no-other-mode --no-cpu --steps 4 --warmup 1 > /dev/null 2> $O/pmc_fw.err
$P --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM -d $O/sqf1 --output-format csv -- python3 bench.py --full --no-other-mode --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sqf1.err
$P --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/sqf2 --output-format csv -- python3 bench.py --full --no-other-mode --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sqf2.err
$P --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $O/sqf3 --output-format csv -- python3 bench.py --full --no-other-mode --no-cpu --steps 3 --warmup 1 > /dev/null 2> $O/sqf3.err || true
python3 tools/pmc_traffic.py $O/pmc_ff $O/pmc_fw --out $O/traffic_c3_full.json --config "c3 full, 1e6 quartets per launch" --commit $(cat .build_commit) --copy-to $O/csv_full
python3 tools/pmc_busy.py $O/sqf1 $O/sqf2 $O/sqf3 --out $O/pmc_busy_full.json --source "rocprofv3 --pmc passes of python3 bench.py --full --no-cpu (c3 full)" --commit $(cat .build_commit)
cp $O/pmc_busy_full.json profiles/pmc_busy_full_latest.json; cp $O/traffic_c3_full.json profiles/traffic_c3_full.json
rm -rf $O/stats_full $O/pmc_ff $O/pmc_fw $O/sqf1 $O/sqf2 $O/sqf3
echo full-mode passes done
rm -rf $O/stats $O/pmc_f $O/pmc_w $O/sq1 $O/sq2 $O/sq3
python3 bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
echo bench done
python3 bench.py --steps 5 --warmup 2 --full > $O/bench_c3_full.json 2> /dev/null
python3 bench.py --steps 5 --warmup 2 --config c2 > $O/bench_c2_n1.json 2> /dev/null
python3 bench.py --steps 5 --warmup 2 --config c2 --full --no-cpu > $O/bench_c2_full.json 2> /dev/null
python3 bench.py --steps 3 --warmup 1 --config c4 > $O/bench_c4_n1.json 2> /dev/null
python3 bench.py --steps 3 --warmup 1 --config c4 --full --no-cpu > $O/bench_c4_full.json 2> /dev/null
echo configs done
python3 bench.py --config c5 --steps 100 --warmup 2 > $O/bench_c5_100_host_sampler.json 2> /dev/null
python3 bench.py --config c5 --steps 100 --warmup 2 --sampler device > $O/bench_c5_100_device_sampler.json 2> /dev/null
echo c5 done
python3 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --gather-ab > $O/bench_two_rank_gloo_rehearsal_bare_command.json 2> $O/gloo.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --sharded --steps 3 --warmup 1 --gather-ab > $O/bench_sharded_one_rank_rccl.json 2> $O/rccl.err || true
for i in 1 2 3; do python3 bench.py --no-cpu --steps 10 --warmup 2 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done > $O/repeatability.txt
ls -la $O
head -c 900 $O/kernel_stats.csv
