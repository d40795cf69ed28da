import csv,sys
def load(fn):
    d={}
    for r in csv.DictReader(open(fn)):
        d[r['Name'].split('(')[0][-60:]]=(int(r['Calls']),float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3)
    return d
a,b=load(sys.argv[1]),load(sys.argv[2])
rows=[]
for k in set(a)|set(b):
    ca,ta,aa=a.get(k,(0,0,0)); cb,tb,ab=b.get(k,(0,0,0))
    rows.append((tb-ta,k,ca,ta,cb,tb))
for r in sorted(rows,key=lambda r:-abs(r[0]))[:18]: print('%+8.2f ms  %-62s  mid %4d %8.2f   new %4d %8.2f' % r)
print('total mid %.1f new %.1f' % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
