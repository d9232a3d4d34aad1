#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the default bench command,
# then FETCH_SIZE / WRITE_SIZE in separate --pmc passes per workload (serial launch groups so the counters
# belong to one kernel at a time).  tools/assemble_profiles.py turns the output into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_round
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# the same command with strictly serial launch groups: every kernel owns the chip while it runs (clean per-kernel roofline)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --lanes 1 > $OUT/bench_trace_serial.json 2> $OUT/bench_trace_serial.err
for W in c2 c3 c5; do
  ST=1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$W/fetch -o p -- python3 $R/bench.py --workload $W --steps $ST --warmup 2 --no-cpu-baseline --no-pmc --lanes 1 --no-events > $OUT/bench_pmc_${W}_1.json 2> $OUT/bench_pmc_${W}_1.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$W/write -o p -- python3 $R/bench.py --workload $W --steps $ST --warmup 2 --no-cpu-baseline --no-pmc --lanes 1 --no-events > $OUT/bench_pmc_${W}_2.json 2> $OUT/bench_pmc_${W}_2.err
done
cd $R
python3 tools/assemble_profiles.py $OUT ${ROUND_TAG:-r02}
