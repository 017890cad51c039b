"""Print per-kernel average durations (and PMC counters when present) from rocprofv3 rocpd databases: python tools/read_rocpd.py dir..."""
import glob, sqlite3, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*.db', recursive=True):
        db = sqlite3.connect(f)
        for name, n, avg in db.execute("select name, count(*), avg(end-start) from kernels group by name order by 3 desc"):
            print(f"{d:40s} {name[:70]:70s} n={n:4d} avg={avg / 1e3:9.1f} us")
        try:
            for k, c, v, n, g in db.execute("select kernel_name, counter_name, avg(value), count(*), grid_size from counters_collection "
                                           "group by kernel_name, counter_name, grid_size"):
                print(f"{k[:110]:110s} grid={g:9d} n={n:4d} {c:12s} avg={v:14.1f}")
        except sqlite3.Error:
            pass
