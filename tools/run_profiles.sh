#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the default bench command (the
# deconvolution + FIR chain, three chains in flight) and of the same with strictly serial launches, of the resident slice
# (imp_slice) alone, FETCH_SIZE / WRITE_SIZE in separate --pmc passes over the chain's kernels (bench.py --pmc-child: serial
# calls, one counter per pass), the stage table of the staged slice, the whole-column errors, and the bench lines of the
# round.  tools/assemble_profiles.py turns the output into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_round
TAG=${ROUND_TAG:-r04}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-slice > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace done"
# the same command with one chain: every kernel owns the chip while it runs (clean per-kernel roofline)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-slice --lanes 1 --blocks 48 > $OUT/bench_trace_serial.json 2> $OUT/bench_trace_serial.err
echo "serial trace done"
# the resident slice: one stream (serial kernels) and three
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_slice_serial -o s -- python3 $R/tools/slice_resident_rate.py c2 1 8 20 3 > $OUT/slice_rate_serial.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_slice -o s -- python3 $R/tools/slice_resident_rate.py c2 3 8 40 3 > $OUT/slice_rate.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_slice_decay -o s -- python3 $R/tools/slice_resident_rate.py c2 1 8 20 3 0.3 > $OUT/slice_rate_decay_serial.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_eq -o s -- python3 $R/tools/eq_fir_profile.py 48000 20 flat > $OUT/eq_fir.txt 2>&1
echo "slice traces done"
for W in c2 c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$W/fetch -o p -- python3 $R/bench.py --workload $W --pmc-child > /dev/null 2> $OUT/bench_pmc_${W}_1.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$W/write -o p -- python3 $R/bench.py --workload $W --pmc-child > /dev/null 2> $OUT/bench_pmc_${W}_2.err
done
echo "pmc done"
cd $R
python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
echo "c2 done"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2_steps20.json 2> $OUT/bench_c2_steps20.err
python3 bench.py --workload c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo "c3 done"
python3 bench.py --workload c5 --steps 8 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
echo "c5 done"
python3 tools/slice_e2e_rate.py 0 48 > $OUT/slice_e2e.txt 2>&1
python3 tools/slice_e2e_rate.py 3 48 >> $OUT/slice_e2e.txt 2>&1
python3 tools/probes/link_probe.py > $OUT/link_probe.txt 2>&1
python3 tools/slice_stages.py 20 > $OUT/slice_stages.txt 2>&1
python3 tools/knee_stats.py 512 >> $OUT/slice_stages.txt 2>&1
for W in c2 c3 c5; do python3 tools/column_error.py $W pair; python3 tools/column_error.py $W; done > $OUT/column_error.txt 2>&1
python3 tools/assemble_profiles.py $OUT $TAG
