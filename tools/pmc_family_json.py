"""rocprofv3 --pmc counter_collection csvs of tools/family_one.py passes -> the JSON bench.py's roofline.traffic reads.
python tools/pmc_family_json.py <family> <kernel substring> <tree> <out.json> name=csv [name=csv ...]"""
import csv
import datetime
import json
import sys
from collections import defaultdict

family, substr, tree, out = sys.argv[1:5]
tot, launches = defaultdict(float), 0
for arg in sys.argv[5:]:
    name, path = arg.split("=", 1)
    seen = set()
    for r in csv.DictReader(open(path)):
        if substr in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r.get("Dispatch_Id", r.get("Dispatch_ID", len(seen))))
    launches = max(launches, len(seen))
rec = {"family": family, "kernel_substring": substr, "local_batch": 256, "launches": launches, "tree": tree,
       "date": datetime.date.today().isoformat(), "fetch_size_kb_sum": tot.get("FETCH_SIZE"), "write_size_kb_sum": tot.get("WRITE_SIZE"),
       "counters": dict(tot),
       "note": "sums over one pass of the family's launch mix (tools/family_one.py), one rocprofv3 --pmc pass per counter set; "
               "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (bench.py doubles FETCH_SIZE: gfx950 counts 64 B per 128-B request)"}
if rec["fetch_size_kb_sum"] and rec["write_size_kb_sum"] and launches:
    rec["traffic_bytes_per_launch"] = (2.0 * rec["fetch_size_kb_sum"] + rec["write_size_kb_sum"]) * 1024.0 / launches
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
