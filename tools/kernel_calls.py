"""List the individual launches of kernels matching a substring within the last frame of a trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "alloc_project" in r["Kernel_Name"]]
lo, hi = marks[-6], marks[-3]
for r in rows[lo:hi]:
    if any(k in r["Kernel_Name"] for k in sys.argv[2:]):
        print(f'{r["Kernel_Name"][:60]:60s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us grid=({r["Grid_Size_X"]},{r["Grid_Size_Y"]}) wg={r["Workgroup_Size_X"]} vgpr={r["VGPR_Count"]}')
