"""Per-kernel sums of the counters in every *counter_collection.csv under a directory (rocprofv3 --pmc passes)."""
import csv, glob, json, os, sys
from collections import defaultdict
out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        e = out[k][row["Counter_Name"]]
        e[0] += float(row["Counter_Value"]); e[1] += 1
print(json.dumps({k: {c: {"sum": v[0], "launches": v[1]} for c, v in d.items()} for k, d in out.items()
                  if "mmp_level" in k or "screen" in k}, indent=1))
