import csv,collections,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keys=sys.argv[2].split('|') if len(sys.argv)>2 else None
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if keys and not any(k in n for k in keys): continue
    d[(n[:70], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:(kv[0][0],int(kv[0][1]))):
    v2=sorted(v); print('%-72s grid %7s x%-3s n=%4d med %7.1f min %7.1f'%(k[0],k[1],k[2],len(v),v2[len(v)//2],v2[0]))
