# the gap between a registration's cold pass and its persistent launch, from a rocprofv3 kernel trace of tools/ab_probe.py
out=gpurun_out/r6l; mkdir -p $out; export TMPDIR=/tmp
for v in 0 1; do
  ( cd /tmp && VISMA_ICP_PERSIST_EARLY=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$out/t$v -o t -- python /root/repo/tools/ab_probe.py 4194304 262144 > /root/repo/$out/ab$v.txt 2>&1 )
  python - <<PY
import csv,glob
f=glob.glob('/root/repo/$out/t$v/**/t_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
P=[i for i,r in enumerate(rows) if 'nn_coop_kernel_persist' in r['Kernel_Name']]
for i in P[1:12:2]:
    j=i-1
    while j>=0 and 'nn_grid_reduce' not in rows[j]['Kernel_Name']: j-=1
    if j<0 or i-j>4: continue
    cs,ce=int(rows[j]['Start_Timestamp']),int(rows[j]['End_Timestamp']); ps,pe=int(rows[i]['Start_Timestamp']),int(rows[i]['End_Timestamp'])
    print("early=$v cold %.1f | gap %.1f | persist %.1f | total %.1f us | between %d"%((ce-cs)/1e3,(ps-ce)/1e3,(pe-ps)/1e3,(pe-cs)/1e3,i-j-1))
PY
done
