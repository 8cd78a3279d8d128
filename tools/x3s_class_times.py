"""conv_x3s_kernel launch classes by time from a per-launch table (RVC_PROF_CSV=<file> python bench.py ...): python tools/x3s_class_times.py <file>"""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
cls=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    if r['kernel']!='conv_x3s_kernel': continue
    k=(r['Ci'],r['Co'],r['k'],r['stride'],r['Tout'],r['Wd'],r['ksplit'],r['tile'])
    cls[k][0]+=1; cls[k][1]+=float(r['us'])
tot=sum(v[1] for v in cls.values())
print('x3s total ms', tot/1e3, 'launches', sum(v[0] for v in cls.values()))
for k,v in sorted(cls.items(), key=lambda kv:-kv[1][1])[:22]:
    print(k, v[0], round(v[1]/v[0],1), round(v[1]/1e3,3))
