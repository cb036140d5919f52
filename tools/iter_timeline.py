"""GPU box: the kernels of ONE LM iteration (between two k_assemble_fronts launches, the last complete one of the
trace) with start / end relative to the first — shows what overlaps when a side stream is in use.
    python tools/iter_timeline.py <rocprofv3 -d dir> [kernel-regex]"""
import csv, glob, os, re, sys
d = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
f = max(glob.glob(d + '/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_assemble_fronts' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').split('<')[0]
    if pat and not pat.search(n):
        continue
    print("%-24s start %9.1f end %9.1f dur %7.1f us  queue %s" % (n, (int(r['Start_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '')))
