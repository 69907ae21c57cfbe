# kernel time per alpha-solve against the batch size (auto alpha_split)
for n in 8 12 16 24 32 48; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --n-orb $n 2>/dev/null > /tmp/ss.json
  python -c "
import json
d=json.load(open('/tmp/ss.json')); P=d['config']['problems']; ms=d['roofline']['kernel_ms']; print('n_orb', $n, 'problems', P, 'kernel ms %.3f' % ms, 'M alpha-solves/s (kernel) %.2f' % (P/ms/1e3), 'iters', d['roofline']['newton_iters_per_solve'], 'conv', d['config']['converged'], d['config']['workgroups'], d['roofline']['kernel'])"
done
