"""Average PMC counters per kernel from a rocprofv3 counter_collection.csv: python tools/pmc_kernel.py <dir> <kernel-substring>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc, num = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); num[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k]/num[k]:16.1f}  (n={num[k]})")
