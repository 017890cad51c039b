# usage: bash tools/prof_one_conv.sh "<one_conv args>" tag  -- kernel-trace durations for a batch sweep (text summary only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 2 8 32 128; do
  export B=$b
  rm -rf /tmp/oc_$2_b$b
  rocprofv3 --kernel-trace --stats -d /tmp/oc_$2_b$b -o t -- python3 $R/tools/one_conv.py $1 bf16 8 > /dev/null 2>&1
  python3 $R/tools/read_rocpd.py /tmp/oc_$2_b$b | grep igemm >> $R/gpurun_out/oc_sweep.txt
done
