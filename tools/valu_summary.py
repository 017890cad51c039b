"""usage: python tools/valu_summary.py tag  -- per-kernel issue-slot occupancy from the five --pmc passes of tools/prof_valu.sh
(gpurun_out/<tag>_pmc_valu_{1..5}.txt).  Busy fractions follow rocprof's derived metrics: VALUBusy = SQ_ACTIVE_INST_VALU*4 / SIMDs / GUI_ACTIVE,
LDS likewise; MFMA = SQ_INSTS_MFMA * 16 cycles (16x16x32 bf16: 4 passes) / SIMDs / GUI_ACTIVE; GUI_ACTIVE is summed over the 8 XCDs;
waves/SIMD = SQ_WAVE_CYCLES*4 / SIMDs / GUI_ACTIVE."""
import collections
import re
import sys

tag = sys.argv[1]
SIMDS = 1024
D = collections.defaultdict(dict)
for i in range(1, 6):
    for line in open(f'gpurun_out/{tag}_pmc_valu_{i}.txt'):
        m = re.match(r'(.+?)\s+grid=\s*(\d+) n=\s*(\d+) (\S+)\s+avg=\s*([\d.]+)', line)
        if m:
            D[re.sub(r'\(.*', '', m.group(1).strip())][m.group(4)] = (float(m.group(5)), int(m.group(3)))
rows = []
for k, v in D.items():
    if 'GRBM_GUI_ACTIVE' not in v or 'SQ_ACTIVE_INST_VALU' not in v:
        continue
    gui = v['GRBM_GUI_ACTIVE'][0] / 8
    g = lambda n: v.get(n, (0, 0))[0]
    rows.append((gui * v['GRBM_GUI_ACTIVE'][1], k, gui, g('SQ_ACTIVE_INST_VALU') * 4 / SIMDS / gui, g('SQ_ACTIVE_INST_LDS') * 4 / SIMDS / gui,
                 g('SQ_INSTS_MFMA') * 16 / SIMDS / gui, g('SQ_WAVE_CYCLES') * 4 / SIMDS / gui, g('SQ_INSTS_VALU'), g('SQ_INSTS_LDS'), g('SQ_INSTS_MFMA'),
                 v['GRBM_GUI_ACTIVE'][1]))
print(f'{"kernel (eager bench pass, b32 640x640 bf16)":<60}{"n":>4}{"kcycles":>9}{"VALU":>6}{"LDS":>6}{"MFMA":>6}{"waves/SIMD":>11}{"VALU inst":>11}{"LDS inst":>10}{"MFMA inst":>10}')
for _, k, gui, valu, lds, mf, wv, iv, il, im, n in sorted(rows, reverse=True)[:28]:
    print(f'{k[:58]:<60}{n:>4}{gui / 1e3:9.1f}{valu:6.2f}{lds:6.2f}{mf:6.2f}{wv:11.1f}{iv / 1e6:10.1f}M{il / 1e6:9.2f}M{im / 1e6:9.2f}M')
