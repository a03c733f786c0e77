"""One step of a kernel trace as a timeline: python tools/step_timeline.py <dir with stats/> <launches per step> [step from the end, default 1]
Every gn_level_kernel dispatch of that step with its start and end relative to the step's first start (microseconds),
its stream and grid -- what overlaps with what when an enqueue uses two streams."""
import csv
import glob
import os
import sys

src, period = sys.argv[1], int(sys.argv[2])
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gn_level_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
n_steps = len(rows) // period
grp = rows[(n_steps - back) * period:(n_steps - back + 1) * period]
t0 = min(int(r["Start_Timestamp"]) for r in grp)
for j, r in enumerate(grp):
    name = r["Kernel_Name"]
    name = name[name.index("gn_level_kernel"):].split("(")[0]
    print(f'{j} stream {r["Stream_Id"]:>2s} {name:58s} wgs {int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]):5d} '
          f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} .. {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f} us')
print(f'step: {(max(int(r["End_Timestamp"]) for r in grp) - t0) / 1e3:.1f} us')
