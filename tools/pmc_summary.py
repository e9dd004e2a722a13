import csv, collections, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/pass*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[-2], k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "n=%d" % len(next(iter(v.values()))))
