"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per launch."""
import csv, glob, sys, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("hvrt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# (the runtime's own fill / copy kernels are kept: a build's counter memsets are part of its traffic)
res = {k: {c: {"mean": sum(v) / len(v), "n": len(v), "sum": sum(v)} for c, v in cs.items()} for k, cs in out.items()
       if k.startswith("k_") or k.startswith("__amd_rocclr")}
print(json.dumps(res, indent=1))
