#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the default bench command,
# then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (serial launch groups so counters belong to one kernel).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_round
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes 1 --no-events > $OUT/bench_pmc1.json 2> $OUT/bench_pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes 1 --no-events > $OUT/bench_pmc2.json 2> $OUT/bench_pmc2.err
cd $R
python3 tools/profile_summary.py $OUT/trace 110 > $OUT/kernel_trace_summary.json
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.json
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cat $OUT/kernel_trace_summary.json $OUT/pmc_summary.json
