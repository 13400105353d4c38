import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + '/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*counter_collection.csv'))[0]
keys = sys.argv[2].split(',')
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    for key in keys:
        if key in k:
            agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for name in agg:
    print(name, {c: round(v / cnt[(name, c)]) for c, v in agg[name].items()})
