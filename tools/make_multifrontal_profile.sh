#!/bin/bash
# rocprofv3 kernel trace of the sparse direct solver on cfg 3 (tools/run_wband.py: multifrontal plan unless FH_MF=0), run from
# the repo root through gpurun; then `python tools/mf_trace.py` summarises one factorisation and one substitution sweep.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_mf
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/run_wband.py > $OUT/run.log 2>&1
cd $ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) gpurun_out/mf_kernel_stats.csv
python3 tools/mf_trace.py > gpurun_out/mf_trace_summary.txt
head -20 gpurun_out/mf_trace_summary.txt
