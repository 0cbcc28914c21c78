import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'].startswith(sys.argv[2]):
        d[int(r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g,v in sorted(d.items()):
    v.sort()
    print('grid',g,'n',len(v),'min %.1f med %.1f max %.1f mean %.1f'%(v[0],v[len(v)//2],v[-1],sum(v)/len(v)))
