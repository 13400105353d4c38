"""Summarise TCNN_AMD_SCATTER_TIMING output (per-task phases of k_grid_scatter) by level."""
import collections, re, sys
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"task\s+(\d+) level\s+(\d+) entries\s+(\d+) samples\s+(\d+) atomic (\d): start\s+([\d.]+) zero\s+([\d.]+) accumulate\s+([\d.]+) flush\s+([\d.]+)", line)
    if m: rows.append([float(v) for v in m.groups()])
by = collections.defaultdict(list)
for r in rows: by[int(r[1])].append(r)
print("level tasks  start(min..max)   zero  accumulate(avg/max)  flush(avg)   end(max)")
for l in sorted(by):
    rs = by[l]
    print(f"{l:5d} {len(rs):5d}  {min(r[5] for r in rs):6.1f}..{max(r[5] for r in rs):6.1f}  {sum(r[6] for r in rs)/len(rs):5.1f}  {sum(r[7] for r in rs)/len(rs):6.1f}/{max(r[7] for r in rs):6.1f}  {sum(r[8] for r in rs)/len(rs):6.1f}    {max(r[5]+r[6]+r[7]+r[8] for r in rs):6.1f}")
print("total end", max(r[5]+r[6]+r[7]+r[8] for r in rows))
