# copy the measurement set of tools/round_profile.sh from gpurun_out/<tag> into profiles/<tag>_*  (run in the build container)
TAG=${1:-r03_b}; S=gpurun_out/$TAG
cp $S/pmc_summary.csv profiles/${TAG}_pmc_summary.csv
[ -f $S/in_flight_pmc_summary.csv ] && cp $S/in_flight_pmc_summary.csv profiles/${TAG}_in_flight_pmc_summary.csv
[ -f $S/lv_pmc_summary.csv ] && cp $S/lv_pmc_summary.csv profiles/${TAG}_lv_pmc_summary.csv
cp $S/bench.json profiles/${TAG}_bench.json
[ -f $S/bench_driver_args.json ] && cp $S/bench_driver_args.json profiles/${TAG}_bench_driver_args.json
cp $S/bench_force_comm.json profiles/${TAG}_bench_force_comm.json
cp "$(ls -t $S/ks/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_mc_kernel_stats.csv
[ -d $S/ks_fl ] && cp "$(ls -t $S/ks_fl/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_in_flight_kernel_stats.csv
[ -d $S/ks_fl ] && python - "$S" "$TAG" <<'PY'
import csv, glob, os, sys
S, TAG = sys.argv[1], sys.argv[2]
# the kernel trace of the region with batches in flight, chain kernel only, as a small CSV (start / end of every dispatch)
rows = []
for path in glob.glob(os.path.join(S, 'ks_fl', '**', '*kernel_trace.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'chain_kernel_mc' in r.get('Kernel_Name', ''):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', ''), r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))))
rows.sort()
t0 = rows[0][0] if rows else 0
with open('profiles/%s_in_flight_kernel_trace.csv' % TAG, 'w') as f:
    f.write('start_ns,end_ns,duration_ns,queue,grid_x,workgroup_x   # mxe::chain_kernel_mc dispatches of: bench.py --in-flight 4 --steps 40 (rocprofv3 --kernel-trace); 512 workgroups = grid 131072: one batch at a time, 256: the cut in flight\n')
    for s, e, q, g, w in rows:
        f.write('%d,%d,%d,%s,%s,%s\n' % (s - t0, e - t0, e - s, q, g, w))
PY
[ -f $S/in_flight_trace.txt ] && cp $S/in_flight_trace.txt profiles/${TAG}_in_flight_trace.txt
[ -d $S/ks_lv ] && cp "$(ls -t $S/ks_lv/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_lv_kernel_stats.csv
cp $S/phases_wg1.txt profiles/${TAG}_phases_mc_wg1_8waves.txt
cp $S/phases_wg2.txt profiles/${TAG}_phases_mc_wg2.txt
cp $S/api_end_to_end.txt profiles/${TAG}_api_end_to_end.txt
for f in underfilled small_batches e2e_wall; do [ -f $S/$f.txt ] && cp $S/$f.txt profiles/${TAG}_$f.txt; done
head -1 profiles/${TAG}_pmc_summary.csv
