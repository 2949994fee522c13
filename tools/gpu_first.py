import sys, time
import os; ROOT=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..'); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
from helpers import *
import plonky2_bls12_381_pairing_amd as B
K=int(sys.argv[1]) if len(sys.argv)>1 else 8
t=time.time(); ctx=B.Context(0,K,0); print("ctx create %.2fs"%(time.time()-t), flush=True)
a=random_fq12(5,1); b=random_fq12(5,2)
out=ctx.fq12_mul_batch(fq12_array(a),fq12_array(b))
print("fq12mul", all(fq12_from_row(out[i])==R.fq12_mul(a[i],b[i]) for i in range(5)), flush=True)
ps,qs=random_pairs(6,3)
ml=ctx.miller_loop_batch(g1_array(ps),g2_array(qs))
print("miller", [fq12_from_row(ml[i])==R.multi_miller_loop([ps[i]],[qs[i]]) for i in range(6)], flush=True)
fe=ctx.final_exponentiation_batch(ml)
print("fe", [fq12_from_row(fe[i])==R.native_pairing([ps[i]],[qs[i]]) for i in range(6)], flush=True)
pr=ctx.pairing_batch(g1_array(ps),g2_array(qs))
print("pairing", [fq12_from_row(pr[i])==R.native_pairing([ps[i]],[qs[i]]) for i in range(6)], flush=True)
print("mac rate %.3e"%ctx.calibrate_mac_rate(), flush=True)
# throughput
n=65536
g1=np.tile(g1_array(ps),(n//6+1,1))[:n]; g2=np.tile(g2_array(qs),(n//6+1,1))[:n]
for rep in range(3):
    t=time.time(); out=ctx.pairing_batch(g1,g2); dt=time.time()-t
    ms=ctx.last_timing()
    print("n=%d wall %.3fs  miller %.2f ms fe %.2f ms total-dev %.2f ms -> %.0f pairings/s (device)"%(n,dt,ms[0],ms[1],ms[2],n/(ms[2]*1e-3)), flush=True)
print("spot", fq12_from_row(out[5])==R.native_pairing([ps[5]],[qs[5]]), fq12_from_row(out[6*1000+2])==R.native_pairing([ps[2]],[qs[2]]))
