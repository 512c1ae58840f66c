import csv, re, sys
d = sys.argv[1]; nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(csv.DictReader(open(f'{d}/run_kernel_stats.csv')))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    short = re.sub(r'^_Z\d+', '', r['Name'])[:48]
    print(f"{short:50s} calls/step {int(r['Calls'])/nsteps:5.1f} ms/step {float(r['TotalDurationNs'])/1e6/nsteps:7.3f} avg {float(r['AverageNs'])/1e3:8.1f} us")
print('sum ms/step', sum(float(r['TotalDurationNs']) for r in rows)/1e6/nsteps)
