import csv, collections, sys, glob
for d in sys.argv[1:]:
    cc=glob.glob(d+'/vm/*_counter_collection.csv')[0]; kt=glob.glob(d+'/vm/*_kernel_trace.csv')[0]
    rows=list(csv.DictReader(open(cc)))
    dd=collections.defaultdict(dict)
    for r in rows:
        if 'gemm' in r['Kernel_Name']: dd[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
    t={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(kt))}
    k=list(dd)[-1]; v=dd[k]; ns=t[k]
    print(d, 'ns', ns, {a:int(b) for a,b in v.items()})
