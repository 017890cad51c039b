"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of tools/prof_pmc.sh:
   python tools/pmc_summary.py gpurun_out/<tag>_pmc_FETCH_SIZE.txt gpurun_out/<tag>_pmc_WRITE_SIZE.txt [commit] -> table on stdout,
   and the conv_igemm family entry of profiles/pmc_traffic.json (bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE, in KiB units of the counters:
   MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B, writes as they are)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r'(.+?)\s+grid=\s*(\d+) n=\s*(\d+) ' + counter + r'\s+avg=\s*([\d.]+)', line)
        if m:
            name = re.sub(r'\(.*', '', m.group(1).strip())
            e = out.setdefault(name, [0.0, 0])
            e[0] += float(m.group(4)) * int(m.group(3)); e[1] += int(m.group(3))
    return out


f, w = read(sys.argv[1], 'FETCH_SIZE'), read(sys.argv[2], 'WRITE_SIZE')
commit = sys.argv[3] if len(sys.argv) > 3 else None
# sys.argv[4]: the bench line the FETCH pass printed (gpurun_out/<tag>_pmc_bench_FETCH_SIZE.json): its roofline.launch_signature is the
# fingerprint of the launch set that was measured; bench.py reports the stored traffic only for that same set
signature = None
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    for line in open(sys.argv[4]):
        if line.startswith('{'):
            signature = json.loads(line).get('roofline', {}).get('launch_signature')
rows = []
for k in f:
    n = f[k][1]
    fetch, write = f[k][0] / n, (w.get(k, [0, 1])[0] / max(w.get(k, [0, 1])[1], 1))
    rows.append((k, n, 2 * fetch * 1024, write * 1024))
print(f'{"kernel":<72}{"launches":>9}{"read MB":>10}{"write MB":>10}{"total MB":>10}')
for k, n, r, wr in sorted(rows, key=lambda t: -(t[2] + t[3]) * t[1]):
    if (r + wr) * n > 1e6:
        print(f'{k[:70]:<72}{n:>9}{r / 1e6:>10.2f}{wr / 1e6:>10.2f}{(r + wr) / 1e6:>10.2f}')
conv = [(n, r, wr) for k, n, r, wr in rows if 'conv_igemm_kernel' in k or 'conv3x3_lds_kernel' in k]      # the conv2d_fwd family of bench.py's roofline
if conv:
    tot = sum(n for n, _, _ in conv)
    by = sum(n * (r + wr) for n, r, wr in conv) / tot
    ent = {'hbm_bytes_per_launch': round(by), 'launches_averaged': tot, 'measured_at': commit, 'launch_signature': signature,
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_pmc.sh, eager bench pass), averaged over the conv_igemm_kernel launches of a step; '
                     'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as read'}
    p = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    d = json.load(open(p)) if os.path.exists(p) else {}
    d[os.environ.get('PMC_KEY', 'conv2d_fwd|bf16|b32|640')] = ent       # PMC_KEY: the bench configuration the passes were run with (BENCH_ARGS of tools/prof_pmc.sh)
    json.dump(d, open(p, 'w'), indent=1)
    print('conv_igemm family:', ent['hbm_bytes_per_launch'], 'bytes per launch over', tot, 'launches')
