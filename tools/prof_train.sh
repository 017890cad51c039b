# usage: bash tools/prof_train.sh tag [dtype] -- rocprofv3 kernel-trace stats of the training bench (hipGraph replay) -> gpurun_out/<tag>_train_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pt_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$1 -o t -- python3 $R/bench.py --mode train --dtype ${2:-bf16} --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$1_train_bench.json 2> $R/gpurun_out/$1_train_bench.err
cp $(find /tmp/pt_$1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$1_train_kernel_stats.csv
tail -1 $R/gpurun_out/$1_train_bench.json | cut -c1-200
# one replayed step in launch order with launch geometry -> gpurun_out/<tag>_train_seq.txt
python3 $R/tools/seq_geom.py $(find /tmp/pt_$1 -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$1_train_seq.txt
