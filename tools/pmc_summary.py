import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").rsplit("(", 1)[0][:56]
    if any(t in n for t in sys.argv[2].split(",")):
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(agg.items()):
    print(n, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
