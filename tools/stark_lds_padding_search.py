#!/usr/bin/env python3
"""Brute-force search of the LDS row padding pad(e) = e + a (e >> 3) + b (e >> 5) + c (e >> 7) + d (e >> 9) for the register
layouts of st::tile_kernel (ntt_stark.hpp): worst bank-conflict degree per layout over every 32-lane group, under a size cap."""
import itertools,sys
def layouts(LOGT):
    L=[]
    for q in range(LOGT//2):
        ls=LOGT-2*q-2
        L.append(lambda t,j,ls=ls: ((t>>ls)<<(ls+2))+(j<<ls)+(t&((1<<ls)-1)))
    if LOGT%2: L.append(lambda t,j:4*t+j)
    return L
for LOGT,limit in ((9,1.2),(10,1.1),(11,1.08),(12,1.11)):
    n=1<<LOGT; lanes=n//4
    best=[]
    for a,b,c,d in itertools.product(range(0,3),range(0,6),range(0,6),range(0,6)):
        pad=lambda e:e+a*(e>>3)+b*(e>>5)+c*(e>>7)+d*(e>>9)
        size=pad(n-1)+1
        if size>limit*n: continue
        tot=0;worst=[]
        for f in layouts(LOGT):
            w=0
            for j in range(4):
                for g in range(0,lanes,32):
                    banks={}
                    for t in range(g,g+32):
                        bk=pad(f(t,j))%32
                        banks[bk]=banks.get(bk,0)+1
                    w=max(w,max(banks.values()))
            worst.append(w);tot+=w
        best.append((tot,size,(a,b,c,d),worst))
    best.sort()
    print(LOGT,best[:3])
