import csv, sys, re, glob
f = (glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv') + glob.glob(sys.argv[1] + '/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    name = r['Name']
    m = re.search(r'(k_[a-z_0-9]+)', name)
    short = (m.group(1) if m else name[:40])
    t = re.search(r'I(.*?)E+v', name)
    print(f"{short:22s} {name[name.find(short)+len(short):][:40]:40s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
