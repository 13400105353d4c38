"""one line per bench json: step time and the pieces"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    p = d["roofline"].get("pieces", {})
    print(f, "ms/step %.4f" % d["ms_per_step"], " ".join(f"{k[:-3]} {v*1e3:.1f}" for k, v in p.items() if k.endswith("_ms")), "frac %.3f" % d["roofline"]["frac"])
