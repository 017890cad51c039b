# usage: bash tools/prof_bench.sh tag  -- rocprofv3 kernel-trace stats of the default bench run; CSV summary -> gpurun_out/<tag>_kernel_stats.csv  (BENCH_ARGS="--dtype fp8": extra bench.py arguments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$1 -o t -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/$1_bench.json 2> $R/gpurun_out/$1_bench.err
cp $(find /tmp/pb_$1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$1_kernel_stats.csv
tail -1 $R/gpurun_out/$1_bench.json
# one replayed step in launch order with launch geometry -> gpurun_out/<tag>_bench_seq.txt
python3 $R/tools/seq_geom.py $(find /tmp/pb_$1 -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$1_bench_seq.txt
