import csv,sys,glob,collections
d=sys.argv[1]; pat=sys.argv[2]
agg=collections.defaultdict(list)
for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print("%-28s %.4g  (n=%d)" % (k, sum(v)/len(v), len(v)))
