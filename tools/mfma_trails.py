import itertools, sys
G=7
pairs={}
idx=0
for a in range(G):
    for b in range(a,G):
        pairs[(a,b)]=idx; idx+=1
def pid(a,b): return pairs[(min(a,b),max(a,b))]
# enumerate closed walks of length 7 (sequence v0..v6), edges (v_i, v_{i+1 mod 7}) all distinct pairs
walks={}
for seq in itertools.product(range(G),repeat=7):
    es=[pid(seq[i],seq[(i+1)%7]) for i in range(7)]
    if len(set(es))<7: continue
    m=0
    for e in es: m|=1<<e
    if m not in walks: walks[m]=seq
print(len(walks),"distinct edge sets")
masks=list(walks)
FULL=(1<<28)-1
# exact cover by 4 masks: index masks by lowest set bit
by_low={}
for m in masks:
    low=(m&-m).bit_length()-1
    by_low.setdefault(low,[]).append(m)
sol=[]
def dfs(cov,chosen):
    if cov==FULL:
        sol.append(list(chosen)); return True
    if len(chosen)==4: return False
    rem=FULL&~cov
    low=(rem&-rem).bit_length()-1
    # need a mask containing bit low, disjoint from cov
    for m in masks_with[low]:
        if m&cov==0:
            chosen.append(m)
            if dfs(cov|m,chosen): return True
            chosen.pop()
    return False
masks_with={b:[m for m in masks if m>>b&1] for b in range(28)}
ok=dfs(0,[])
print(ok)
if ok:
    for m in sol[0]: print(walks[m])
