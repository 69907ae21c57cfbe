# kernel time of the cfg4 batch against alpha_split (pieces per alpha scan)
for s in 8 9 10 11 12 13 14; do
  timeout -k 10 100 python bench.py --no-cpu-baseline --steps 50 --alpha-split $s 2>/dev/null > /tmp/ss.json
  python -c "
import json
d=json.load(open('/tmp/ss.json')); print($s, round(d['roofline']['kernel_ms'],4), d['roofline']['newton_iters_per_solve'], d['config']['converged'])"
done
