#!/bin/bash
# rocprofv3 --kernel-trace --stats of the dense configurations (cfg 2, cfg 5 fp64, cfg 5 mixed); run from the
# repo root through gpurun.  Usage: tools/make_dense_profiles.sh <tag>
set -e
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_dense
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg2 -- python3 $ROOT/tools/run_cfg2.py > $OUT/cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -- python3 $ROOT/tools/run_cfg5.py > $OUT/cfg5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5m -- python3 $ROOT/tools/run_cfg5.py 8192 24 32 > $OUT/cfg5m.log 2>&1
cd $ROOT
cp $(find $OUT/cfg2 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_cfg2_kernel_stats.csv
cp $(find $OUT/cfg5 -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_cfg5_kernel_stats.csv
cp $(find $OUT/cfg5m -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_cfg5_mixed_kernel_stats.csv
grep -h -E "pass|time=" $OUT/cfg2.log $OUT/cfg5.log $OUT/cfg5m.log > gpurun_out/${TAG}_dense_runs.txt
rm -rf $OUT
cat gpurun_out/${TAG}_dense_runs.txt
