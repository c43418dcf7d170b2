#!/bin/bash
# Trapezoid (fpm[16] = 1) nodes on tall ellipses against the bench's Gauss / 4000 setting: one line per run.
for c in "gauss 4000" "trapezoid 300" "trapezoid 400" "trapezoid 600" "trapezoid 800" "trapezoid 1200" "trapezoid 1600" "trapezoid 2400" "zolotarev 100"; do set -- $c
  python bench.py --steps 3 --warmup 1 --headline-only --contour $1 --aspect $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 $2 :', d['ms_per_step'], 'ms', d['value'], 'eig/s loops', d['loops'], 'its', d['krylov_iterations_per_step'], 'res %.1e' % d['max_residual'], 'ok' if d['converged'] else 'FAIL', sorted(d['node_iterations_last_step'].items(), key=lambda kv: -kv[1])[:3])"
done
