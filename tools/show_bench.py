#!/usr/bin/env python3
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ".join(sys.argv[2:]), round(d["value"]), "ms/it %.2f" % d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
