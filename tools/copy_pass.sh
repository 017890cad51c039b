# usage: bash tools/copy_pass.sh tag commit  -- copy the artefacts of tools/final_pass_r3.sh <tag> A|B|C from gpurun_out/ into profiles/ (and refresh profiles/pmc_traffic.json)
T=$1; H=$2; cd $(dirname $0)/..
python tools/pmc_summary.py gpurun_out/${T}_pmc_FETCH_SIZE.txt gpurun_out/${T}_pmc_WRITE_SIZE.txt $H gpurun_out/${T}_pmc_bench_FETCH_SIZE.json > gpurun_out/${T}_pmc_traffic_per_kernel.txt
cd gpurun_out
for f in bench_line bench_inflight1 bench_f32 bench_fp8_b32 bench_fp8_b64 bench_bf16_b64 train_bf16 train_f32 tood_s_1280_b8_infer tood_s_1280_b8_infer_fp8 tood_s_1280_b8_train; do tail -1 ${T}_$f.json > ../profiles/${T}_$f.json; done
cp ${T}_kernel_stats.csv ../profiles/${T}_bench_bf16_b32_inflight4_kernel_stats.csv; cp ${T}_serial_kernel_stats.csv ../profiles/${T}_bench_bf16_b32_one_stream_kernel_stats.csv
tail -1 ${T}_bench.json > ../profiles/${T}_bench_line_under_rocprof.json; tail -1 ${T}_serial_bench.json > ../profiles/${T}_bench_one_stream_line_under_rocprof.json
cp ${T}_bench_seq.txt ../profiles/${T}_bench_step_geometry_inflight4.txt; cp ${T}_serial_bench_seq.txt ../profiles/${T}_bench_step_geometry.txt
cp ${T}_train_kernel_stats.csv ../profiles/; tail -1 ${T}_train_bench.json > ../profiles/${T}_train_line_under_rocprof.json; cp ${T}_train_seq.txt ../profiles/${T}_train_step_geometry.txt
cp ${T}_per_launch_bf16_b32.txt ${T}_per_layer_roofline.txt ${T}_csp_block_phases.txt ${T}_cnx_block_phases.txt ${T}_nms_phases.txt ${T}_gputests.log ../profiles/
cp ${T}_pmc_FETCH_SIZE.txt ../profiles/${T}_pmc_FETCH_SIZE_per_kernel.txt; cp ${T}_pmc_WRITE_SIZE.txt ../profiles/${T}_pmc_WRITE_SIZE_per_kernel.txt; cp ${T}_pmc_traffic_per_kernel.txt ../profiles/
cp ${T}_pmc_mfma.txt ../profiles/${T}_pmc_MFMA_BUSY_raw.txt; cp ${T}_pmc_MFMA_busy_per_kernel.txt ../profiles/
