"""Dispatch model behind the tile orders of the lean fp32 gather kernel: one XCD = 32 CUs x 8 resident
tiles, in-order dispatch to the least-loaded CU, a CU's MFMA pipe shared equally by its resident tiles, tile length =
its number of valid taps.  Prints the makespan of dec.deconv4 forward (19x19 output pixels, 'valid' 3x3) per XCD for
the tile orders tried, against the ideal (total work / 32).  CPU only; no GPU code."""
import heapq, itertools
def simulate(work, ncu=32, slots=8):
    # processor sharing per CU; in-order dispatch to the CU with a free slot and fewest residents
    t=0.0; q=list(work); qi=0
    cus=[[] for _ in range(ncu)]  # remaining work lists
    # initial fill round robin
    while qi<len(q) and any(len(c)<slots for c in cus):
        c=min(range(ncu), key=lambda i:(len(cus[i]), i))
        if len(cus[c])>=slots: break
        cus[c].append(q[qi]); qi+=1
    while any(cus):
        # next completion: each CU's tiles progress at rate 1/len
        dt=min(min(c)*len(c) for c in cus if c)
        t+=dt
        for c in cus:
            if c:
                r=dt/len(c)
                c[:]=[w-r for w in c]
        for c in cus:
            done=[w for w in c if w<=1e-9]
            if done:
                c[:]=[w for w in c if w>1e-9]
        while qi<len(q):
            c=min(range(ncu), key=lambda i:(len(cus[i]), i))
            if len(cus[c])>=slots: break
            cus[c].append(q[qi]); qi+=1
    return t
def nv(y,H,k=3):  # valid taps for 'valid' stride-1 deconv output coordinate y, input H-2
    return min(y+1,k,H-y)
H=19
o1=[nv(y,H)*3 for y in range(H) for t in range(19)]        # per XCD: 19 tiles per image row
o2=[nv(y,H)*nv(x,H) for y in range(H) for x in range(H)]
ideal1=sum(o1)/32; ideal2=sum(o2)/32
print("order1", simulate(o1), ideal1); print("order2", simulate(o2), ideal2)
o2s=sorted(o2, reverse=True); print("order2 LPT", simulate(o2s))
# centre-out rows
def centre_out(n):
    c=n//2; out=[c]
    for k in range(1,n):
        for s in (c+k,c-k):
            if 0<=s<n and s not in out: out.append(s)
    return out
co=centre_out(H)
o2c=[nv(y,H)*nv(x,H) for y in co for x in co]; print("order2 centre-out", simulate(o2c))
o1c=[nv(y,H)*3 for y in co for t in range(19)]; print("order1 centre-out", simulate(o1c))
print("---- variants")
import random
pix=[(y,x) for y in range(H) for x in range(H)]
w=lambda p: nv(p[0],H)*nv(p[1],H)
# ring order: by Chebyshev distance from centre ascending
c=H//2
ring=sorted(pix,key=lambda p:(max(abs(p[0]-c),abs(p[1]-c)),p))
print("ring", simulate([w(p) for p in ring]))
# interior rows first in natural order, then border rows
rows=[y for y in range(H) if 2<=y<=H-3]+[1,H-2,0,H-1]
cols=rows
print("interior-first rows natural x", simulate([w((y,x)) for y in rows for x in range(H)]))
print("interior-first rows+cols", simulate([w((y,x)) for y in rows for x in cols]))
# all interior (9) pixels first, then 6, 4, 3, 2, 1
print("LPT stable", simulate(sorted([w(p) for p in pix], reverse=True)))
for seed in range(3):
    random.seed(seed); r=pix[:]; random.shuffle(r); print("random", simulate([w(p) for p in r]))
# B=1024: two groups per XCD interleaved
print("B1024 order2", simulate([w(p) for p in pix for g in range(2)]), 2*ideal2)
print("B1024 order2 centre-out", simulate([w((y,x)) for y in co for x in co for g in range(2)]))
# deconv2 'same' 8x8, gx=2, 4 slots per CU effectively (1024 blocks / 256 CU)
def nvs(y,H): return 3-(y==0)-(y==H-1)
H2=8
d2=[nvs(y,H2)*nvs(x,H2) for y in range(H2) for x in range(H2) for b in range(2)]
print("deconv2 order2", simulate(d2), sum(d2)/32, " order1", simulate([nvs(y,H2)*3 for y in range(H2) for t in range(8) for b in range(2)]), sum(nvs(y,H2)*3*16 for y in range(H2))/32)
