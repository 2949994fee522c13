import sys, collections
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import vm_sim as S
tot=0
for nm in ("miller","fe2"):
    p=S.get_program(nm,8,0)
    hd=p['hdrs']; est=0; cnt=collections.Counter()
    for s in range(p['nsteps']):
        typ,mt=hd[2*s],hd[2*s+1]
        if typ==0: est+=1.9; cnt['mul']+=1
        elif typ==3: est+=2.7; cnt['dot2']+=1
        elif typ==1:
            fl=(mt>>8)&3; est+=0.35+0.1*(mt&0xff)+(0.27 if fl&2 else (0.08 if fl else 0)); cnt['sum']+=1
        else: est+=0.5; cnt['copy']+=1
        if mt&0x10000: est+=0.06
        if mt&0x20000: est+=0.06
    print(nm, p['nsteps'], dict(cnt), 'est %.0f us'%est)
    tot+=est
print('total est %.0f us -> %.2f ms per 65536'%(tot, tot*4/1000))
