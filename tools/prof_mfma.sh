# usage: bash tools/prof_mfma.sh tag  -- MFMA-busy pass of an eager bench pass: SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE per kernel (kernel-trace only)
# -> gpurun_out/<tag>_pmc_mfma.txt ; tools/mfma_summary.py turns it into the per-kernel table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$1_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_$1_mfma -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/read_rocpd.py /tmp/pmc_$1_mfma > $R/gpurun_out/$1_pmc_mfma.txt
