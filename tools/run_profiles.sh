#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the default bench command (the
# deconvolution + FIR chain, three chains in flight) and of the same with strictly serial launches, FETCH_SIZE / WRITE_SIZE
# in separate --pmc passes over the chain's kernels (bench.py --pmc-child: serial calls, one counter per pass), and the
# bench lines of the round.  tools/assemble_profiles.py turns the output into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_round
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# the same command with one chain: every kernel owns the chip while it runs (clean per-kernel roofline)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --lanes 1 --blocks 48 > $OUT/bench_trace_serial.json 2> $OUT/bench_trace_serial.err
for W in c2 c3; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$W/fetch -o p -- python3 $R/bench.py --workload $W --pmc-child > /dev/null 2> $OUT/bench_pmc_${W}_1.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$W/write -o p -- python3 $R/bench.py --workload $W --pmc-child > /dev/null 2> $OUT/bench_pmc_${W}_2.err
done
cd $R
python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2_steps20.json 2> $OUT/bench_c2_steps20.err
python3 bench.py --workload c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --workload c5 --steps 8 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python3 tools/assemble_profiles.py $OUT ${ROUND_TAG:-r03}
