import csv, glob, re, sys, collections
sys.path.insert(0,'scratch')
from ablate2 import VARIANTS
f=sorted(glob.glob('gpurun_out/prof_abl/**/*_kernel_trace.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
per=collections.defaultdict(list)
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']); n=re.sub(r"[<(].*","",n).replace("void ","")
    if n.startswith('k_'): per[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for k in ('k_locate','k_splat_hw','k_zcol_fwd','k_silhouette_loss','k_zcol_bwd','k_gather_hw'):
    d=[x[1] for x in sorted(per[k])]
    n=len(d)//len(VARIANTS)
    print(k, ' '.join("%s=%.1f"%(VARIANTS[i][0], sum(d[i*n+5:(i+1)*n])/max(1,len(d[i*n+5:(i+1)*n]))/1e3) for i in range(len(VARIANTS))))
