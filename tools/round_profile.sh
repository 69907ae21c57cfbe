# the measurement set of a round, on the GPU box:  gpurun -- 'bash tools/round_profile.sh r05_c'
set -e
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-extras --warmup 1 --in-flight 1"      # (ONE launch with the GPU to itself: the line's roofline)
PMC_ISSUE="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
PMC_INSTS="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- $B --steps 5 > $OUT/prof_ks.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC_ISSUE --output-format csv -d $OUT/pmc_issue -- $B --steps 2 > $OUT/prof_pmc_issue.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC_INSTS --output-format csv -d $OUT/pmc_insts -- $B --steps 2 > $OUT/prof_pmc_insts.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 2 > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 2 > $OUT/prof_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $B --steps 2 > $OUT/prof_pmc_l2.log 2>&1 || true
python tools/summarize_pmc.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err || true
# the cut of the batches in flight (mxe_opts.in_flight = 4: 256 workgroups, four pieces per scan), launched ALONE: its counters
BF="$B --cut-for-in-flight 4"
rocprofv3 --kernel-trace --pmc $PMC_ISSUE --output-format csv -d $OUT/flpmc_issue -- $BF --steps 2 > $OUT/prof_flpmc_issue.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC_INSTS --output-format csv -d $OUT/flpmc_insts -- $BF --steps 2 > $OUT/prof_flpmc_insts.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/flpmc_fetch -- $BF --steps 2 > $OUT/prof_flpmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/flpmc_write -- $BF --steps 2 > $OUT/prof_flpmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/flpmc_l2 -- $BF --steps 2 > $OUT/prof_flpmc_l2.log 2>&1 || true
python tools/summarize_pmc.py $OUT chain_kernel_mc flpmc_ > $OUT/in_flight_pmc_summary.csv 2> $OUT/in_flight_pmc_summary.err || true
# the region with four batches in flight itself: every dispatch with start and end (the program itself after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_fl -- python3 bench.py --no-cpu-baseline --no-extras --warmup 1 --in-flight 4 --steps 40 > $OUT/prof_ks_fl.log 2>&1
python tools/in_flight_trace.py $OUT/ks_fl > $OUT/in_flight_trace.txt 2>&1 || true
# chain_kernel_lv (BASELINE config 2 / 3 in binary32): kernel stats and counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_lv -- python3 tools/lv_launches.py 20 > $OUT/prof_ks_lv.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC_ISSUE --output-format csv -d $OUT/lvpmc_issue -- python3 tools/lv_launches.py 4 > $OUT/prof_lvpmc_issue.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/lvpmc_insts -- python3 tools/lv_launches.py 4 > $OUT/prof_lvpmc_insts.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d $OUT/lvpmc_lds -- python3 tools/lv_launches.py 4 > $OUT/prof_lvpmc_lds.log 2>&1 || true
python tools/summarize_pmc.py $OUT chain_kernel_lv lvpmc_ > $OUT/lv_pmc_summary.csv 2> $OUT/lv_pmc_summary.err || true
# the counters first: bench.py reads them from profiles/ (and only when their source hash is that of the library it runs)
cp $OUT/pmc_summary.csv profiles/${TAG}_pmc_summary.csv
cp $OUT/in_flight_pmc_summary.csv profiles/${TAG}_in_flight_pmc_summary.csv
cp $OUT/lv_pmc_summary.csv profiles/${TAG}_lv_pmc_summary.csv
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 3 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err || true
python bench.py --force-comm --no-cpu-baseline --no-extras > $OUT/bench_force_comm.json 2> $OUT/bench_force_comm.err
python tools/profile_phases.py --layout 4 --split 0 --theta 1e-5 --wgpc 1 > $OUT/phases_wg1.txt 2>&1 || true
python tools/profile_phases.py --layout 4 --split 0 --theta 1e-5 --wgpc 2 > $OUT/phases_wg2.txt 2>&1 || true
python tools/api_end_to_end.py > $OUT/api_end_to_end.txt 2>&1 || true
cut -c1-900 $OUT/bench.json; cat $OUT/pmc_summary.csv | head -5; cat $OUT/in_flight_trace.txt | head -30; find $OUT -name "*kernel_stats.csv" | head -5
