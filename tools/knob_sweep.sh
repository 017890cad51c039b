# usage: bash tools/knob_sweep.sh  -- default bench (no CPU baseline) under a few values of the experiment knobs; one line per run -> gpurun_out/knob_sweep.txt
R=${GRAFT_REPO_ROOT:-.}
out=$R/gpurun_out/knob_sweep.txt
: > $out
run() { echo -n "$1 $2: " >> $out; env $1 python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out; }
for rep in 1 2; do
run X=0
run MGDT_CONV_PANEL_KIB=150
run "MGDT_CONV_PANEL_KIB=150 MGDT_CONV_GCAP=256"
run "MGDT_CONV_PANEL_KIB=144 MGDT_CONV_GCAP=256"
run "MGDT_CONV_PANEL_KIB=150 MGDT_CONV_GCAP=384"
run MGDT_CONV_GCAP=256
run MGDT_CONV_GCAP=384
done
run X=0 "--batch 64"
run "MGDT_CONV_PANEL_KIB=150 MGDT_CONV_GCAP=256" "--batch 64"
run "MGDT_CONV_GCAP=256" "--dtype fp8"
run "MGDT_CONV_PANEL_KIB=150 MGDT_CONV_GCAP=256" "--dtype f32"
run X=0 "--dtype f32"
cat $out
