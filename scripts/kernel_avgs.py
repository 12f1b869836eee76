"""Average durations from a rocprofv3 kernel_stats.csv, names shortened: kernel_avgs.py <file> [substring]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Name"]:
        n = r["Name"].replace("hvrt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{n[:48]:48s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}  max {float(r['MaxNs']) / 1e3:9.1f}")
