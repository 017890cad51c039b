"""usage: seq_geom.py <kernel_trace.csv> <out.txt> -- one replayed step of a rocprofv3 kernel trace in launch order: us, workgroups, threads, static
LDS, VGPRs, resident workgroups per CU (from VGPRs / static LDS / threads; dynamic LDS is not in the trace) and rounds of workgroups over 256 CUs."""
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
        cap = max(1, min((per_simd * 4) // waves if waves else 1, (160 * 1024) // lds if lds else 64, 2048 // max(wg, 1)))
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000:8.1f} {nwg:5d} {wg:4d} {lds:6d} {vg:4d} {cap:5d} {nwg / (256 * cap):6.2f}  {r['Kernel_Name'][:90]}\n")
