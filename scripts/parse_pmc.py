"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per launch."""
import csv, glob, sys, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("hvrt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in out.items() if k.startswith("k_")}
print(json.dumps(res, indent=1))
