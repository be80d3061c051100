import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    t = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        t[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1
    print("==", d)
    for k, v in t.items():
        if "tmp::" in k:
            print(k, n[k], {a: "%.3g" % b for a, b in v.items()})
