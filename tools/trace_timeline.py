# Timeline of the last launches of a rocprofv3 --kernel-trace --output-format csv run:
#   python tools/trace_timeline.py <dir or *_kernel_trace.csv> [n_last] [name filter]
# prints start (us, relative), duration, gap to the previous kernel's end, grid, name
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(path)))
rows = [r for r in rows if flt in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
    name = r["Kernel_Name"].replace("void ", "").replace("raht::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  wgs {grid // max(wg, 1):6d}  {name}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us")
