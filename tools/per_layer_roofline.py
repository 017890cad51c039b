"""Per-launch roofline table from tools/profile_ops.py's output (us, GB/s, TFLOP/s of every launch that carries its algorithmic bytes / flops):
   python tools/per_layer_roofline.py profiles/<tag>_per_launch_bf16_b32.txt > profiles/<tag>_per_layer_roofline.txt
Columns: measured us, fraction of the 8 TB/s HBM peak, fraction of the dense bf16 MFMA peak (2.5 PFLOP/s), attainable us = max(bytes / 8 TB/s, flops / 2.5 PFLOP/s)
(per-layer min(flops/peak, bytes/bw) as the attainable time), measured / attainable."""
import re, sys
HBM, MFMA = 8000.0, 2500.0      # GB/s, TFLOP/s (MI355X_MICROARCH.md)
rows, other = [], 0.0
for line in open(sys.argv[1]):
    m = re.match(r'\s*(\d+)\s+(\S+)\s+(\(.*?\))?\s+([\d.]+)(?:\s+(\d+)\s+([\d.]+))?\s*$', line)
    if not m:
        continue
    idx, op, shape, us, gbs, tfs = m.groups()
    us = float(us)
    if gbs is None:
        other += us
        rows.append((int(idx), op, shape or '', us, None, None, None))
        continue
    gbs, tfs = float(gbs), float(tfs)
    att = max(gbs * us / HBM, tfs * us / MFMA)       # bytes / peak bw, flops / peak rate (both in us)
    rows.append((int(idx), op, shape, us, gbs / HBM, tfs / MFMA, att))
print(f'{"#":>3} {"op":<20} {"shape (b,cin,h,w,cout,k,s)":<34} {"us":>7} {"HBM frac":>9} {"MFMA frac":>10} {"attainable us":>14} {"x off":>6}')
tm = ta = 0.0
for idx, op, shape, us, fh, fm, att in rows:
    if att is None:
        print(f'{idx:>3} {op:<20} {shape:<34} {us:>7.1f} {"-":>9} {"-":>10} {"-":>14} {"-":>6}')
    else:
        tm += us; ta += att
        print(f'{idx:>3} {op:<20} {shape:<34} {us:>7.1f} {fh:>9.3f} {fm:>10.3f} {att:>14.1f} {us / att:>6.1f}')
print(f'launches with algorithmic bytes / flops: {tm:.1f} us measured vs {ta:.1f} us attainable ({tm / ta:.1f}x); launches without (pointwise / pooling / NMS): {other:.1f} us')
