"""Print a rocprofv3 kernel_stats.csv with short kernel names: python tools/kstats.py <dir> [min_pct]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void ", "").replace("s3d::(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:70]
    if float(r["Percentage"]) >= (float(sys.argv[2]) if len(sys.argv) > 2 else 0.5):
        print(f"{n:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} pct={r['Percentage']}")
