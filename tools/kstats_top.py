"""print the top kernels of a rocprofv3 kernel_stats.csv: name (truncated), calls, avg us, total ms, %"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    name = r["Name"].replace("void s3d::(anonymous namespace)::", "").replace("void at::native::", "")[:70]
    print(f"{name:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.1f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f} %")
