# kernel time of ONE rank's shard of the cfg4 batch at N = 1, 2, 4, 8 (no gather), by piece count (0 = the library's choice);
# RANK_OF_SHARD=r times rank r's shard (the job ends with its slowest rank: element 221, on rank 221 mod N, has the
# most expensive cold starts of the batch)
mkdir -p gpurun_out/shard
for n in 1 2 4 8; do for sp in 0 ${SPLITS:-16 34 50}; do
python bench.py --no-cpu-baseline --no-extras --steps 100 --shard-of $n --shard-rank ${RANK_OF_SHARD:-0} --alpha-split $sp > gpurun_out/shard/b.json 2> gpurun_out/shard/b.err
python -c "
import json,sys; d=json.load(open('gpurun_out/shard/b.json')); print('shard 1/%s split %s: problems %d kernel %.3f ms step %.3f ms iters %.3f wgs %d %s' % (sys.argv[1], sys.argv[2], d['config']['problems_on_rank0'], d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['newton_iters_per_solve'], d['config']['workgroups'], d['roofline']['kernel']))" $n $sp
done; done
