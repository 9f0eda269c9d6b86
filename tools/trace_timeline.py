"""Development aid: the last kernels of a rocprofv3 --kernel-trace run as a timeline (start, duration, queue)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f"{r['Kernel_Name'][:58]:58s} queue {r.get('Queue_Id', '?'):>3s}  start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  "
          f"duration {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
