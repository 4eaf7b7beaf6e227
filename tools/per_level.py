"""Per-level kernel time from a rocprofv3 kernel trace taken with one level per launch (S3D_BIN_PASS_BYTES=1):
python tools/per_level.py <dir> <kernel-substring> <levels> [skip_launches]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
L = int(sys.argv[3])
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"])
rows = rows[int(sys.argv[4]) if len(sys.argv) > 4 else 0:]
n = len(rows) // L
for rep in range(n):
    print(f"launch {rep}: " + " ".join(f"{(e - s) / 1e3:6.1f}" for s, e in rows[rep * L:(rep + 1) * L]))
