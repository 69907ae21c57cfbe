# copy the measurement set of tools/round_profile.sh from gpurun_out/<tag> into profiles/<tag>_*  (run in the build container)
TAG=${1:-r03_b}; S=gpurun_out/$TAG
cp $S/pmc_summary.csv profiles/${TAG}_pmc_summary.csv
cp $S/bench.json profiles/${TAG}_bench.json
cp $S/bench_force_comm.json profiles/${TAG}_bench_force_comm.json
cp "$(ls -t $S/ks/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_mc_kernel_stats.csv
cp $S/phases_wg1.txt profiles/${TAG}_phases_mc_wg1_8waves.txt
cp $S/phases_wg2.txt profiles/${TAG}_phases_mc_wg2.txt
cp $S/api_end_to_end.txt profiles/${TAG}_api_end_to_end.txt
for f in underfilled small_batches e2e_wall; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/${TAG}_$f.txt; done
head -1 profiles/${TAG}_pmc_summary.csv
