# usage: bash tools/prof_train.sh tag [dtype] -- rocprofv3 kernel-trace stats of the training bench (hipGraph replay) -> gpurun_out/<tag>_train_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pt_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$1 -o t -- python3 $R/bench.py --mode train --dtype ${2:-bf16} --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$1_train_bench.json 2> $R/gpurun_out/$1_train_bench.err
cp $(find /tmp/pt_$1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$1_train_kernel_stats.csv
tail -1 $R/gpurun_out/$1_train_bench.json | cut -c1-200
# kernel sequence of the last complete step (name, µs) -> gpurun_out/<tag>_train_seq.txt  (who issues the small copies / fills)
python3 - $(find /tmp/pt_$1 -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$1_train_seq.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'conv_stem_kernel' in n or 'stem2_kernel' in n]
lo, hi = (idx[-2], idx[-1]) if len(idx) >= 2 else (0, len(rows))
with open(sys.argv[2], 'w') as f:
    for r in rows[lo:hi]:
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:8.1f}  {r['Kernel_Name'][:110]}\n")
PY
