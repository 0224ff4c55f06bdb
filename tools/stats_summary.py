import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
calls=max(int(r['Calls']) for r in rows if 'adam' in r['Name'])
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total us/step:", tot/1e3/calls)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    n=r['Name'].replace('void ','').replace('(anonymous namespace)::','')[:70]
    print(f"{n:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} us/step={float(r['TotalDurationNs'])/1e3/calls:8.1f}")
