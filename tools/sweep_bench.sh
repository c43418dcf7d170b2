#!/bin/bash
# Sweep of the bench's solver settings on cfg 3 (inner tolerance, iteration cap, ellipse ratio): one line per setting.
for a in 4000 5000; do for r in 0.02 0.03 0.05 0.1; do for m in 40 50 70; do
  python bench.py --steps 3 --warmup 1 --headline-only --aspect $a --inner-rtol $r --maxit $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('aspect $a rtol $r maxit $m :', d['ms_per_step'], 'ms', d['value'], 'eig/s loops', d['loops'], 'its', d['krylov_iterations_per_step'], 'res %.1e' % d['max_residual'], 'ok' if d['converged'] else 'FAIL')"
done; done; done
