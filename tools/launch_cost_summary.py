import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
by=collections.defaultdict(list)
prev=None
for r in rows:
    name=r['Kernel_Name'].split('(')[0]
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    gap=(int(r['Start_Timestamp'])-int(prev['End_Timestamp']))/1e3 if prev else 0
    key=name+(' after '+prev['Kernel_Name'].split('(')[0] if prev and 'trivial' in name else '')
    by[key].append((dur,gap))
    prev=r
for k,v in by.items():
    v=v[3:]
    if not v: continue
    print('%-60s n=%3d dur %6.2f us  gap before %6.2f us' % (k[:60], len(v), sum(a for a,b in v)/len(v), sum(b for a,b in v)/len(v)))
