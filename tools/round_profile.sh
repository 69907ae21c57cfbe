set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/f_tests.log 2>&1
python bench.py > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
rm -rf gpurun_out/prof_f
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f/ks -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/f_prof_ks.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_f/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/f_prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_f/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/f_prof_write.log 2>&1
make -C maxent_amd/csrc prof > /dev/null 2>&1 || true
python tools/profile_phases.py --layout 4 --split 10 --waves 4 --theta 1e-5 > gpurun_out/f_phases_mc.txt 2>&1
tail -3 gpurun_out/f_tests.log; cat gpurun_out/f_bench.json | cut -c1-400; find gpurun_out/prof_f -name "*.csv" | head -20
