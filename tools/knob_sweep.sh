# usage: bash tools/knob_sweep.sh  -- default bench (no CPU baseline) under a few values of the experiment knobs; one line per run -> gpurun_out/knob_sweep.txt
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/knob_sweep.txt
: > $out
run() { echo -n "$1 $2: " >> $out; env $1 python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out; }
run X=0
run MGDT_SPR_VECS=1024
run MGDT_SPR_VECS=4096
run MGDT_SPR_VECS=8192
run MGDT_CONV_MINWG=64
run MGDT_CONV_MINWG=200
run MGDT_CONV_MINWG=256
run MGDT_CONV_WAVES=4
run "MGDT_CONV_WAVES=4 MGDT_CONV_GCAP=512"
run MGDT_DW_TS=8
run MGDT_CONV_PANEL_KIB=128
run X=0
cat $out
