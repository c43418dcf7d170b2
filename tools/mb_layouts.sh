#!/bin/bash
# per-iteration cost of one rank's share for the (node group) x (column group) layouts of N = 1, 2, 4, 8 GPUs
for cfg in "16 64" "16 32" "16 16" "8 64" "8 32" "8 16" "4 64" "4 32" "2 64"; do
  set -- $cfg
  SOLVER=cocg NOPROF=1 M=$2 timeout -k 10 120 python tools/mb_apply.py $1 40 2 | head -1 || exit 1
done
