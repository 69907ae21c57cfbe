# one launch against the batch size (the library's own cut): kernel time, alpha-solves per second of the kernel and of the step
for n in 8 12 16 24 32 48; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --in-flight 1 --steps 30 --n-orb $n 2>/dev/null > /tmp/ss.json
  python -c "
import json
d=[json.loads(l) for l in open('/tmp/ss.json') if l.startswith('{')][0]; P=d['config']['problems_per_step']; ms=d['kernel_ms']; print('n_orb %2d problems %6d kernel ms %.3f = %.2f M alpha-solves/s; step %.3f ms = %.2f M; evaluations per alpha %.3f; converged %d; %s, %d workgroups' % ($n, P, ms, P/ms/1e3, d['ms_per_step'], d['value']/1e6, d['evals_per_solve'], d['config']['converged_on_rank0'], d['roofline']['kernel'], d['config']['workgroups']))"
done
