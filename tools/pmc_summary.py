import csv,sys,glob,collections
d=sys.argv[1]
f=glob.glob(d+'/*/*counter_collection.csv')[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys()) if len(sys.argv)>2 else None
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')[:58]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
# also kernel durations from trace
tr=glob.glob(d+'/*/*kernel_trace.csv')[0]
dur=collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    k=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')[:58]
    dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
names=sorted(agg.keys(), key=lambda k:-sum(dur[k])) 
for k in names:
    if 'at::' in k or 'rocprim' in k or 'amd_' in k: continue
    c={n:sum(v)/len(v) for n,v in agg[k].items()}
    du=sum(dur[k])/len(dur[k])
    s=f"{k:58s} {du:7.1f}us "
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        # matrix-pipe busy cycles (summed over the 1024 SIMDs) against the kernel's duration at the NOMINAL 2.4 GHz the roofline
        # peak is quoted at.  (GRBM_GUI_ACTIVE / 8 / duration is not used as a clock or a denominator any more: it reads high on
        # dispatches under ~0.3 ms -- MI355X_MICROARCH.md, DVFS give-back -- and every kernel here is under 0.1 ms: round 3's
        # column showed 2.6 ... 11 GHz.)
        util=c['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*du*2400)*100 if du else 0
        s+=f"mfma_busy_at_2.4GHz={util:5.1f}% "
    for n in ('SQ_INSTS_MFMA','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_BUSY_CYCLES','FETCH_SIZE','WRITE_SIZE','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_INSTS_LDS','SQ_ACTIVE_INST_LDS','SQ_INSTS_VALU','SQ_ACTIVE_INST_VALU'):
        if n in c: s+=f"{n.replace('SQ_','')}={c[n]:.3g} "
    print(s)
