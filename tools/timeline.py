import csv,sys
# usage: timeline.py <kernel_trace.csv> [anchor kernel prefix]  -- one replay, node by node (from one anchor kernel to the next)
rows=list(csv.DictReader(open(sys.argv[1])))
anchors=tuple(sys.argv[2:]) or ('sample_kernel','sample_prepare_kernel')
ev=[(r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:30], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
ev.sort(key=lambda x:x[1])
idx=[i for i,e in enumerate(ev) if e[0].startswith(anchors)]
i0=idx[-20]; i1=idx[-19]
t0=ev[i0][1]; prev=None
for n,s,e in ev[i0:i1+1]:
    gap='' if prev is None else f"gap {(s-prev)/1e3:5.1f}"
    print(f"{(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:7.1f} us  {gap:10s} {n}")
    prev=e
