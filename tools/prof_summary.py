import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', round(tot / 1e6 / steps, 2))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls'])//steps:>5d} {float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step {float(r['Percentage']):6.2f}%")
