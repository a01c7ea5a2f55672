import csv,sys,glob,collections
d=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else ''
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('void ','').split('(')[0]
        if pat and pat not in n: continue
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in acc.items():
    print(n)
    for k,v in sorted(c.items()):
        print(f"   {k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
