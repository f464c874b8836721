import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Queue_Id','')) for r in rows]
ev.sort()
def short(n):
    import re
    n=n.replace('hps::','')
    return n[:60]
# print a window of 60 kernels from the middle
mid=len(ev)*3//4
t0=ev[mid][0]
for s,e,n,q in ev[mid:mid+70]:
    print(f"{(s-t0)/1000:9.1f} {(e-t0)/1000:9.1f} {(e-s)/1000:7.1f} q{q} {short(n)}")
