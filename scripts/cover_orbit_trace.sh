#!/bin/bash
# kernel trace of scripts/cover_orbit.py: the cover pass's kernels view by view
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/cot -o cot --output-format csv -- python3 /root/repo/scripts/cover_orbit.py > /root/repo/gpurun_out/cot.log 2>&1 || exit 1
python3 - <<'P'
import csv
rows = [r for r in csv.DictReader(open('/root/repo/gpurun_out/cot/cot_kernel_trace.csv')) if 'k_cover' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    n = r['Kernel_Name'].replace('hvrt::(anonymous namespace)::', '').split('(')[0]
    print(f"{n:24s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
P
