mkdir -p gpurun_out/r05
for i in 1 2; do for p in 0 1 2; do VISMA_ICP_PERSIST_PRIO=$p python tools/ab_probe.py 4194304 262144 | sed "s/\"env\": \"\"/\"env\": \"prio=$p\"/"; done; done > gpurun_out/r05/ab_prio.txt 2>&1
cat gpurun_out/r05/ab_prio.txt
for p in 0 1 2; do VISMA_ICP_PERSIST_PRIO=$p VISMA_ICP_PERSIST_TIMELINE=gpurun_out/r05/tl_prio$p.bin python tools/ab_probe.py 4194304 262144 > /dev/null 2>&1; done
python - <<'PY'
import numpy as np
for p in (0,1,2):
    raw=np.fromfile('gpurun_out/r05/tl_prio%d.bin'%p,dtype=np.uint64); pos=0; k=0
    while pos+3<=len(raw):
        passes,blocks,ran=(int(x) for x in raw[pos:pos+3]); n=2*passes*blocks
        rec=(raw[pos+3:pos+3+n].reshape(passes,blocks,2)&np.uint64(0xFFFFFFFFFFF)).astype(np.int64); pos+=3+n; k+=1
        if k not in (2,4): continue
        m=min(passes,ran); rec=rec[:m]; body=(rec[:,:,1]-rec[:,:,0])*0.01
        t0=rec[:,:,0].min(axis=1); step=np.median((t0[1:]-t0[:-1])*0.01)
        b=body[1:].mean(axis=0)
        print("prio",p,"launch",k,"step %.2f"%step,"slot means:",[round(float(b[s*256:(s+1)*256].mean()),2) for s in range(4)],"max body median %.2f"%np.median(body[1:].max(axis=1)))
PY
rm -f gpurun_out/r05/tl_prio*.bin
