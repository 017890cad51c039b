# usage: bash tools/prof_bench.sh tag  -- rocprofv3 kernel-trace stats of the default bench run; CSV summary -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$1 -o t -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $R/gpurun_out/$1_bench.json 2> $R/gpurun_out/$1_bench.err
cp $(find /tmp/pb_$1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$1_kernel_stats.csv
tail -1 $R/gpurun_out/$1_bench.json
# launch geometry of one replayed step: µs, workgroups, threads, LDS, VGPRs, resident workgroups per CU, rounds over 256 CUs -> gpurun_out/<tag>_bench_seq.txt
python3 - $(find /tmp/pb_$1 -name "*kernel_trace.csv" | head -1) $R/gpurun_out/$1_bench_seq.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem2_kernel' in r['Kernel_Name'] or 'conv_stem_kernel' in r['Kernel_Name']]
lo, hi = (idx[-2], idx[-1]) if len(idx) >= 2 else (0, len(rows))
with open(sys.argv[2], 'w') as f:
    f.write('      us   nwg  thr    lds vgpr wg/cu rounds  kernel\n')
    for r in rows[lo:hi]:
        wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
        grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        nwg = grid // max(wg, 1)
        lds = int(r.get('LDS_Block_Size', 0) or 0)
        vg = int(r.get('VGPR_Count', 0) or 0) + int(r.get('Accum_VGPR_Count', 0) or 0)
        waves = -(-wg // 64)
        per_simd = max(1, min(8, 512 // max(vg, 1)))
        cap = min((per_simd * 4) // waves if waves else 1, (160 * 1024) // lds if lds else 64, 2048 // max(wg, 1))
        cap = max(cap, 1)
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:8.1f} {nwg:5d} {wg:4d} {lds:6d} {vg:4d} {cap:5d} {nwg / (256 * cap):6.2f}  {r['Kernel_Name'][:90]}\n")
PY
