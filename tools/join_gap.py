import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find builds: sequences of fock kernels; report gap last-fock-end -> next non-fock start, and first fock start after previous non-fock end
out=[]
i=0
n=len(rows)
isf=lambda r: 'qc_fock' in r['Kernel_Name']
gaps=[];spans=[];pre=[]
k=0
while k<n:
    if isf(rows[k]):
        j=k; end=0
        start=int(rows[k]['Start_Timestamp'])
        while j<n and (isf(rows[j])):
            end=max(end,int(rows[j]['End_Timestamp'])); j+=1
        if j<n and k>0:
            gaps.append(int(rows[j]['Start_Timestamp'])-end); spans.append(end-start); pre.append(start-int(rows[k-1]['End_Timestamp']))
            nxt=rows[j]['Kernel_Name'][:40]
        k=j
    else: k+=1
import statistics as st
m=len(gaps)
sel=slice(m//2,m)
print('builds',m,'span us med',st.median(spans[sel])/1e3,'join gap us med',st.median(gaps[sel])/1e3,'pre gap us med',st.median(pre[sel])/1e3,'next',nxt)
