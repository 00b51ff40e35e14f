#!/bin/bash
# round 3: kernel stats of the reference-default job (every model trains on 10 000 rows) at HEAD
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03aj; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --train-rows 10000 --no-cpu-baseline --roofline-steps 1 > $OLDPWD/$O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_train_rows_10000_kernel_stats.csv
grep '^{"metric' $O/trace.log | tail -1 | cut -c1-200
head -14 $O/bench_train_rows_10000_kernel_stats.csv | cut -c1-150
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
