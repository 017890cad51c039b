# usage: bash tools/prof_cmd.sh tag <python script + args>  -- kernel-trace durations -> gpurun_out/<tag>_kernels.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/pc_$tag
rocprofv3 --kernel-trace --stats -d /tmp/pc_$tag -o t -- python3 $R/$@ > /dev/null 2>&1
python3 $R/tools/read_rocpd.py /tmp/pc_$tag > $R/gpurun_out/${tag}_kernels.txt
