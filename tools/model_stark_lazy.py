#!/usr/bin/env python3
"""Python big-integer model of stark_rings_amd/csrc/stark_lazy.hpp (nine signed 28-bit limbs, lazy carries, R = 2^280): the
Montgomery product with its column bounds asserted (every accumulator stays inside int64 for limbs up to +-2^31), relax, fold and
the canonical representative.  Running it checks the model against plain integers and prints the constants the header embeds
(2^560 mod p for tw_from_u64).  tests/test_model_stark_lazy.py runs a shortened version on the CPU."""
import sys
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
import random
p = 2**251 + 17*2**192 + 1
M = (1<<28)-1
PL = [1,0,0,0,0,0,1<<24,1,1<<27]
assert sum(v<<(28*i) for i,v in enumerate(PL)) == p
def val(x): return sum(v<<(28*i) for i,v in enumerate(x))
def to_limbs(v):
    return [(v>>(28*i))&M for i in range(8)] + [v>>224]
maxacc=0
def i32(x):
    assert -2**31 <= x < 2**31, x
    return x
def mont(a,w):
    global maxacc
    acc=0; m=[0]*10; r=[0]*9
    for k in range(19):
        for i in range(9):
            j=k-i
            if 0<=j<9: acc += a[i]*w[j]
        for i in range(10):
            if i<k or k>=10:
                j=k-i
                if j==6: acc += m[i]<<24
                if j==7: acc += m[i]
                if j==8: acc += m[i]<<27
        maxacc=max(maxacc,abs(acc))
        assert -2**63 <= acc < 2**63
        if k<10:
            m[k] = (-acc) & M
            acc += m[k]
            assert acc & M == 0
            acc >>= 28
        elif k<18:
            r[k-10] = acc & M
            acc >>= 28
        else:
            r[8] = i32(acc)
    return r
def relax(x):
    x=list(x)
    for i in range(8):
        c = x[i]>>28
        x[i] &= M
        x[i+1] = i32(x[i+1]+c)
    return x
def fold(x):
    x=list(x)
    q = x[8]>>27
    x[8] &= (1<<27)-1
    x[7]-=q; x[6]-= q<<24; x[0]-=q
    return [i32(v) for v in x]
def canon(x):
    x=fold(relax(x))
    x[0]+=1; x[6]+=1<<24; x[7]+=1; x[8]+=1<<27
    x=relax(x)
    assert 0<=x[8]<(1<<28)
    y=list(x); y[0]-=1; y[6]-=1<<24; y[7]-=1; y[8]-=1<<27
    y=relax(y)
    return x if y[8]<0 else y
R=2**280
rng=random.Random(1)
for t in range(ITER):
    # lazy a: limbs up to +-2^31
    a=[rng.randrange(-2**31+16,2**31-16) for _ in range(9)]
    if t%3==0: a=to_limbs(rng.randrange(p))
    w=to_limbs(rng.randrange(p))
    if t%5==0: w=[rng.randrange(-40,1<<28) for _ in range(8)]+[rng.randrange(-3,1<<27)]
    r=mont(a,w)
    assert (val(r)*R - val(a)*val(w))%p==0
    assert all(0<=v<=M for v in r[:8]), r
    assert abs(val(r)) < 2*p
    c=canon(r)
    assert val(c)==val(r)%p and all(0<=v<=M for v in c)
    c=canon(a)
    assert val(c)==val(a)%p and all(0<=v<=M for v in c), (val(c), val(a)%p)
    f=fold(relax(a))
    assert val(f)%p==val(a)%p and -2**206<val(f)<2**251
print("ok maxacc bits", maxacc.bit_length())
print("R2 limbs (2^560 mod p):", [hex(v) for v in to_limbs(pow(2,560,p))])
print("one table form", [hex(v) for v in to_limbs(pow(2,280,p))])
