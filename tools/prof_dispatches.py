#!/usr/bin/env python3
"""Per-dispatch kernel durations from a rocprofv3 kernel trace: prof_dispatches.py <dir> <name substring> [group]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void agx::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
       for r in rows if sys.argv[2] in r["Kernel_Name"]]
grp = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for i in range(0, len(seq), grp):
    g = seq[i:i + grp]
    print(f"{g[0][0]:50s}", " ".join(f"{t:6.0f}" for _, t in g))
