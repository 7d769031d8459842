import csv,sys,glob,collections
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        if "mlp" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
