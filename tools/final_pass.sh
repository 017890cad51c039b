# usage: bash tools/final_pass.sh tag  -- everything profiles/<tag>_* is made of, in one call: GPU tests, bench lines of every configuration, rocprofv3 kernel stats,
# PMC traffic and MFMA-busy passes (each rocprofv3 run is its own process; counters never share a run with --stats)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
python -m pytest tests -m gpu -x -q > $O/${T}_gputests.log 2>&1; tail -1 $O/${T}_gputests.log
python bench.py > $O/${T}_bench_line.json 2> /dev/null && echo default ok
python bench.py --dtype f32 --no-cpu-baseline > $O/${T}_bench_f32.json 2> /dev/null
python bench.py --dtype fp8 --no-cpu-baseline > $O/${T}_bench_fp8_b32.json 2> /dev/null
python bench.py --dtype fp8 --batch 64 --no-cpu-baseline > $O/${T}_bench_fp8_b64.json 2> /dev/null
python bench.py --batch 64 --no-cpu-baseline > $O/${T}_bench_bf16_b64.json 2> /dev/null
python bench.py --mode train --steps 30 --warmup 5 > $O/${T}_train_bf16.json 2> /dev/null
python bench.py --mode train --dtype f32 --steps 30 --warmup 5 > $O/${T}_train_f32.json 2> /dev/null
python bench.py --model mspa_c2f_gd_tood_yolov8_hidc128 --scale s --imgsz 1280 --batch 8 --no-cpu-baseline > $O/${T}_tood_s_1280_b8_infer.json 2> /dev/null
python bench.py --model mspa_c2f_gd_tood_yolov8_hidc128 --scale s --imgsz 1280 --batch 8 --mode train --steps 20 --warmup 3 > $O/${T}_tood_s_1280_b8_train.json 2> /dev/null
echo lines done
python tools/profile_ops.py > $O/${T}_per_launch_bf16_b32.txt 2>&1
bash tools/prof_bench.sh ${T} > /dev/null && echo prof bf16 ok
BENCH_ARGS="--dtype fp8" bash tools/prof_bench.sh ${T}_fp8 > /dev/null && echo prof fp8 ok
bash tools/prof_pmc.sh ${T} && BENCH_ARGS="--dtype fp8" bash tools/prof_pmc.sh ${T}_fp8 && echo pmc ok
bash tools/prof_mfma.sh ${T} && echo mfma ok
for f in bench_line bench_f32 bench_fp8_b32 bench_fp8_b64 bench_bf16_b64 train_bf16 train_f32 tood_s_1280_b8_infer tood_s_1280_b8_train; do python3 -c "import sys,json; d=json.loads(open('$O/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'] if d.get('roofline') else None)"; done
