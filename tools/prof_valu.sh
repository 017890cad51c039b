# usage: bash tools/prof_valu.sh tag  -- instruction-mix counters of an eager bench pass per kernel (kernel-trace only; separate small --pmc passes):
#   SQ_INSTS_VALU SQ_ACTIVE_INST_VALU | SQ_INSTS_LDS SQ_ACTIVE_INST_LDS | SQ_INSTS_MFMA SQ_WAVE_CYCLES | SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
# -> gpurun_out/<tag>_pmc_valu_<n>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=0
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY"; do
  n=$((n+1))
  rm -rf /tmp/pmc_$1_v$n
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$1_v$n -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
  python3 $R/tools/read_rocpd.py /tmp/pmc_$1_v$n > $R/gpurun_out/$1_pmc_valu_$n.txt 2>&1
  echo "pass $n ($c) done" >> $R/gpurun_out/$1_progress.log
done
