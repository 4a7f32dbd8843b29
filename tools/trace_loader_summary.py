import csv,glob,sys
d=sys.argv[1]
ks=[];cs=[]
for f in glob.glob(d+"/**/*_kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tsdf' in r['Kernel_Name']: ks.append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
for f in glob.glob(d+"/**/*_memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        cs.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Direction',''),))
ks.sort();cs.sort()
import statistics as st
kd=[(b-a)/1e3 for a,b in ks]
print("kernels",len(ks),"median us",st.median(kd),"max",max(kd))
big=[c for c in cs if c[1]-c[0]>200000]
bd=[(b-a)/1e3 for a,b,_ in big]
print("big copies",len(big),"median us",st.median(bd), "min",min(bd),"max",max(bd))
gaps=[(big[i+1][0]-big[i][1])/1e3 for i in range(len(big)-1)]
print("gaps between big copies: median us",st.median(gaps))
small=[(b-a)/1e3 for a,b,_ in cs if b-a<=200000]
print("small copies",len(small),"median us",st.median(small),"max",max(small))
