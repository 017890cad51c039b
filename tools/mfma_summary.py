"""Per-kernel MFMA occupancy from tools/prof_mfma.sh: python tools/mfma_summary.py gpurun_out/<tag>_pmc_mfma.txt
   SQ_VALU_MFMA_BUSY_CYCLES = cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16: MI355X_MICROARCH.md);
   GRBM_GUI_ACTIVE = active cycles summed over the 8 XCDs.  busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024): the share of SIMD-cycles of the dispatch in
   which the matrix pipe was executing (1.0 = dense MFMA peak)."""
import re, sys
busy, act, dur = {}, {}, {}
for line in open(sys.argv[1]):
    m = re.match(r'(.+?)\s+grid=\s*(\d+) n=\s*(\d+) (\S+)\s+avg=\s*([\d.]+)', line)
    if m:
        name = re.sub(r'\(.*', '', m.group(1).strip())
        d = busy if m.group(4) == 'SQ_VALU_MFMA_BUSY_CYCLES' else act if m.group(4) == 'GRBM_GUI_ACTIVE' else None
        if d is not None:
            e = d.setdefault(name, [0.0, 0]); e[0] += float(m.group(5)) * int(m.group(3)); e[1] += int(m.group(3))
        continue
    m = re.match(r'\S+\s+(.+?)\s+n=\s*(\d+) avg=\s*([\d.]+) us', line)
    if m:
        dur[re.sub(r'\(.*', '', m.group(1).strip())] = (int(m.group(2)), float(m.group(3)))
print(f'{"kernel":<72}{"launches":>9}{"avg us":>9}{"MFMA busy":>11}')
rows = []
for k in busy:
    if k in act and act[k][0] > 0:
        frac = (busy[k][0] / busy[k][1]) / ((act[k][0] / act[k][1]) / 8 * 1024)
        n, us = dur.get(k, (busy[k][1], 0.0))
        rows.append((k, n, us, frac))
for k, n, us, frac in sorted(rows, key=lambda t: -t[2] * t[1]):
    if frac > 0.0005:
        print(f'{k[:70]:<72}{n:>9}{us:>9.1f}{frac:>11.3f}')
