#!/bin/bash
# Regenerate the judged profile artefacts of the bench command on the GPU box (run from the repo
# root through gpurun).  Three rocprofv3 passes of the SAME command: --stats, --pmc FETCH_SIZE,
# --pmc WRITE_SIZE (counters in their own passes, kernel-trace only).  Usage: tools/make_profiles.sh <tag>
set -e
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD="bench.py --steps 3 --warmup 1 --headline-only"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/$CMD > $OUT/bench_stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/$CMD > $OUT/bench_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/$CMD > $OUT/bench_write.log 2>&1
echo "write pass done"
cd $ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_kernel_stats.csv
grep '^{"metric"' $OUT/bench_stats.log | tail -1 > gpurun_out/${TAG}_bench_line_under_rocprof.json
python3 tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) \
    gpurun_out/${TAG}_pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 $CMD (two separate passes)" \
    $(python3 -c "import bench; print(bench.kernel_source_hash())")
rm -rf $OUT/stats $OUT/fetch $OUT/write
head -8 gpurun_out/${TAG}_bench_kernel_stats.csv
