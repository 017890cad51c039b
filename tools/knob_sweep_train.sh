# usage: bash tools/knob_sweep_train.sh  -- captured bf16 training step under a few values of the training-side experiment knobs -> gpurun_out/knob_sweep_train.txt
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/knob_sweep_train.txt
: > $out
run() { echo -n "$1: " >> $out; env $1 python3 $R/bench.py --mode train --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out; }
run X=0
run MGDT_BN_V=8
run MGDT_WGRAD_FILL=256
run MGDT_WGRAD_FILL=640
run MGDT_WGRAD_FILL=0
run MGDT_WGRAD_SPLITS=256
run MGDT_WGRAD_SPLITS=1024
run MGDT_CONV_PANEL_KIB=128
run MGDT_CONV_GCAP=512
run X=0
cat $out
