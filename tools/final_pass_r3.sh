# usage: bash tools/final_pass_r3.sh tag part   (part A | B | C; each is one gpurun call, every step appends to gpurun_out/<tag>_progress.log)
# A: GPU tests, every bench line.  B: rocprofv3 kernel stats (inference + training), per-launch table, phase stamps.  C: PMC passes (HBM traffic, MFMA busy).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; P=$O/${T}_progress.log
cd $R
say() { echo "$(date +%T) $*" >> $P; }
if [ "$2" = "A" ]; then
  say "gpu tests"; python -m pytest tests -m gpu -x -q > $O/${T}_gputests.log 2>&1; tail -1 $O/${T}_gputests.log >> $P
  say "bench default"; python bench.py > $O/${T}_bench_line.json 2> $O/${T}_bench_line.err
  say "bench one batch at a time"; python bench.py --inflight 1 --no-cpu-baseline > $O/${T}_bench_inflight1.json 2> /dev/null
  say "bench f32"; python bench.py --dtype f32 --no-cpu-baseline > $O/${T}_bench_f32.json 2> /dev/null
  say "bench fp8"; python bench.py --dtype fp8 --no-cpu-baseline > $O/${T}_bench_fp8_b32.json 2> /dev/null
  say "bench fp8 b64"; python bench.py --dtype fp8 --batch 64 --no-cpu-baseline > $O/${T}_bench_fp8_b64.json 2> /dev/null
  say "bench bf16 b64"; python bench.py --batch 64 --no-cpu-baseline > $O/${T}_bench_bf16_b64.json 2> /dev/null
  say "train bf16"; python bench.py --mode train --steps 30 --warmup 5 > $O/${T}_train_bf16.json 2> /dev/null
  say "train f32"; python bench.py --mode train --dtype f32 --steps 30 --warmup 5 > $O/${T}_train_f32.json 2> /dev/null
  say "tood s infer"; python bench.py --model mspa_c2f_gd_tood_yolov8_hidc128 --scale s --imgsz 1280 --batch 8 --no-cpu-baseline > $O/${T}_tood_s_1280_b8_infer.json 2> /dev/null
  say "tood s infer fp8"; python bench.py --model mspa_c2f_gd_tood_yolov8_hidc128 --scale s --imgsz 1280 --batch 8 --dtype fp8 --no-cpu-baseline > $O/${T}_tood_s_1280_b8_infer_fp8.json 2> /dev/null
  say "tood s train"; python bench.py --model mspa_c2f_gd_tood_yolov8_hidc128 --scale s --imgsz 1280 --batch 8 --mode train --steps 20 --warmup 3 > $O/${T}_tood_s_1280_b8_train.json 2> /dev/null
  say "done A"
  for f in bench_line bench_inflight1 bench_f32 bench_fp8_b32 bench_fp8_b64 bench_bf16_b64 train_bf16 train_f32 tood_s_1280_b8_infer tood_s_1280_b8_infer_fp8 tood_s_1280_b8_train; do python3 -c "import sys,json; d=json.loads(open('$O/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'] if d.get('roofline') else None)" | tee -a $P; done
elif [ "$2" = "B" ]; then
  say "per launch"; python tools/profile_ops.py > $O/${T}_per_launch_bf16_b32.txt 2>&1
  python tools/per_layer_roofline.py $O/${T}_per_launch_bf16_b32.txt > $O/${T}_per_layer_roofline.txt 2>&1
  say "prof bench"; bash tools/prof_bench.sh ${T} > /dev/null 2>&1
  # one batch at a time on ONE stream (no parallel branch, no batches in flight): the per-kernel durations the roofline object is scaled to
  say "prof bench serial"; export MGDT_SIDE_STREAM=0; BENCH_ARGS="--inflight 1" bash tools/prof_bench.sh ${T}_serial > /dev/null 2>&1; unset MGDT_SIDE_STREAM
  say "prof train"; bash tools/prof_train.sh ${T} > /dev/null 2>&1
  say "csp phases"; MGDT_CSP_DBG=1 python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline 2>&1 | grep "csp_block mode" | sort | uniq -c | sort -rn > $O/${T}_csp_block_phases.txt
  say "cnx phases"; MGDT_CNX_DBG=1 python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline 2>&1 | grep "^cnx_block" | tail -6 > $O/${T}_cnx_block_phases.txt
  say "nms phases"; python tools/nms_dbg.py > $O/${T}_nms_phases.txt 2>&1
  say "done B"
elif [ "$2" = "C" ]; then
  say "pmc"; bash tools/prof_pmc.sh ${T}
  say "mfma"; bash tools/prof_mfma.sh ${T}
  python3 tools/mfma_summary.py $O/${T}_pmc_mfma.txt > $O/${T}_pmc_MFMA_busy_per_kernel.txt 2>&1
  say "done C"
fi
