# usage: bash tools/prof_pmc.sh tag  -- FETCH_SIZE and WRITE_SIZE (separate passes, kernel-trace only) of an eager bench pass  (BENCH_ARGS="--dtype fp8": extra bench.py arguments)
# -> gpurun_out/<tag>_pmc_{fetch,write}.txt (per kernel: dispatch count, average counter value)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$1_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$1_$c -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/$1_pmc_bench_$c.json 2> /dev/null
  python3 $R/tools/read_rocpd.py /tmp/pmc_$1_$c | grep "$c" > $R/gpurun_out/$1_pmc_$c.txt
done
