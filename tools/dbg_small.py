import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import oracle_lib as O
from stark_rings_amd import CyclotomicRing
F=O.GOLDILOCKS; D=24
ring=CyclotomicRing("goldilocks24")
batch=4
a=O.fill_uniform(F,21,0,batch*D); b=O.fill_uniform(F,22,0,batch*D)
got=ring.ntt_mul(a.copy(), b)
want=O.small("sro_g24_ntt_mul", a, b)
p=2**64-2**32+1
for i in range(batch*D):
    if got[i]!=want[i]:
        print(i, i%3, hex((int(got[i])-int(want[i]))%p))
print("done")
