# usage: bash tools/sweep_conv.sh  -- per-layer sensitivity of conv_igemm to the tile knobs (kernel-trace durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/sweep_conv.txt; rm -f $out
for shape in "128 256 3 2 40 40" "64 128 3 2 80 80" "32 64 3 2 160 160" "16 32 3 2 320 320" "64 96 3 1 80 80" "80 80 3 1 80 80" "32 32 3 1 40 40" "64 64 3 1 20 20" "512 256 1 1 20 20" "480 96 1 1 40 40"; do
  for cfg in "8 128" "4 128" "8 32" "8 512" "4 512"; do
    set -- $cfg
    export MGDT_CONV_WAVES=$1 MGDT_CONV_MINWG=$2
    rm -rf /tmp/sw
    rocprofv3 --kernel-trace --stats -d /tmp/sw -o t -- python3 $R/tools/one_conv.py $shape bf16 6 > /dev/null 2>&1
    t=$(python3 $R/tools/read_rocpd.py /tmp/sw | grep igemm | head -1 | sed 's/.*avg= *//')
    echo "$shape | waves $1 minwg $2 | $t" >> $out
  done
done
