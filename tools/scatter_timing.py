"""Summarise TCNN_AMD_SCATTER_TIMING output (per-task phases of the grid gradient kernels; laboratory build) by level and by XCD
(block b runs on XCD b % 8)."""
import collections, re, sys
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"task\s+(\d+) level\s+(\d+) entries\s+(\d+) samples\s+(\d+) atomic (\d): start\s+([\d.]+) zero\s+([\d.]+) accumulate\s+([\d.]+) flush\s+([\d.]+)", line)
    if m: rows.append([float(v) for v in m.groups()])
if not rows: sys.exit("no task lines")
by = collections.defaultdict(list)
for r in rows: by[int(r[1])].append(r)
end = lambda r: r[5] + r[6] + r[7] + r[8]
print("level tasks  start(min..max)   zero  accumulate(avg/max)  flush(avg)   end(max)  xcds")
for l in sorted(by):
    rs = by[l]
    xs = sorted({int(r[0]) % 8 for r in rs})
    print(f"{l:5d} {len(rs):5d}  {min(r[5] for r in rs):6.1f}..{max(r[5] for r in rs):6.1f}  {sum(r[6] for r in rs)/len(rs):5.1f}  {sum(r[7] for r in rs)/len(rs):6.1f}/{max(r[7] for r in rs):6.1f}  {sum(r[8] for r in rs)/len(rs):6.1f}    {max(end(r) for r in rs):6.1f}  {xs}")
print("xcd tasks  busy(sum of task times)  end(max)  levels")
for x in range(8):
    rs = [r for r in rows if int(r[0]) % 8 == x]
    if rs: print(f"{x:3d} {len(rs):5d}  {sum(r[6]+r[7]+r[8] for r in rs):8.1f}  {max(end(r) for r in rs):6.1f}  {sorted({int(r[1]) for r in rs})}")
print("total end", max(end(r) for r in rows))
