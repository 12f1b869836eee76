"""Last launches of a rocprofv3 kernel trace as a timeline: start / end in microseconds relative to the first kernel shown."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("hvrt::(anonymous namespace)::", "").split("(")[0][:40]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f}  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  {name}  grid {r.get('Grid_Size_X','')} wg {r.get('Workgroup_Size_X','')} vgpr {r.get('VGPR_Count','')}")
