"""summarise rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + '/*/*_counter_collection.csv'), key=__import__('os').path.getmtime)[-1:]
    if not f:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').split('<')[0]
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
    print("==", d)
    for n, cs in sorted(acc.items()):
        print("%-26s" % n, "  ".join("%s: n=%d mean=%.4g sum=%.4g" % (c, len(v), sum(v) / len(v), sum(v)) for c, v in sorted(cs.items())))
