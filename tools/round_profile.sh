# the measurement set of a round, on the GPU box:  gpurun -- 'bash tools/round_profile.sh r02_c'
set -e
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-extras --warmup 1 --in-flight 1"      # (the profiles are of ONE launch with the GPU to itself)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- $B --steps 5 > $OUT/prof_ks.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/pmc_issue -- $B --steps 2 > $OUT/prof_pmc_issue.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_insts -- $B --steps 2 > $OUT/prof_pmc_insts.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 2 > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 2 > $OUT/prof_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $B --steps 2 > $OUT/prof_pmc_l2.log 2>&1 || true
python tools/summarize_pmc.py $OUT > $OUT/pmc_summary.csv 2> $OUT/pmc_summary.err || true
# the counters first: bench.py reads them from profiles/ (and only when their source hash is that of the library it runs)
cp $OUT/pmc_summary.csv profiles/${TAG}_pmc_summary.csv
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --force-comm --no-cpu-baseline --no-extras > $OUT/bench_force_comm.json 2> $OUT/bench_force_comm.err
python tools/profile_phases.py --layout 4 --split 0 --theta 1e-5 --wgpc 1 > $OUT/phases_wg1.txt 2>&1 || true
python tools/profile_phases.py --layout 4 --split 0 --theta 1e-5 --wgpc 2 > $OUT/phases_wg2.txt 2>&1 || true
python tools/api_end_to_end.py > $OUT/api_end_to_end.txt 2>&1 || true
cut -c1-600 $OUT/bench.json; cat $OUT/pmc_summary.csv; find $OUT -name "*kernel_stats.csv" | head -3
