"""print per-kernel stats + one-factorisation timeline from a rocprofv3 --kernel-trace csv dir"""
import csv, glob, sys
d = sys.argv[1]
f = max(glob.glob(d + '/*/*_kernel_stats.csv'), key=__import__('os').path.getmtime)
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').split('<')[0]
    print("%-26s calls %5s total %9.3f ms avg %9.1f us  min %8.1f max %9.1f" % (
        n, r['Calls'], int(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, int(r['MinNs']) / 1e3, int(r['MaxNs']) / 1e3))
if len(sys.argv) > 2:
    f = max(glob.glob(d + '/*/*_kernel_trace.csv'), key=__import__('os').path.getmtime)
    rows = list(csv.DictReader(open(f)))
    idx = [i for i, r in enumerate(rows) if 'k_assemble_blocks' in r['Kernel_Name'] or 'k_assemble_fronts' in r['Kernel_Name']]
    a, b = idx[-2], idx[-1]
    t0 = int(rows[a]['Start_Timestamp'])
    for r in rows[a - 2:b]:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').split('<')[0]
        print("%-24s start %9.1f us dur %8.1f us grid %7s wg %4s" % (n, (int(r['Start_Timestamp']) - t0) / 1e3,
              (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], r['Workgroup_Size_X']))
