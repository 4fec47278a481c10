import numpy as np, sys, itertools
sys.path.insert(0,'/root/repo')
from tetrad_amd import synth
from oracle import oracle as orc
orc.build()
tmparr, tmpmap, q = synth.make_config("c3", Q=300)
_, rstat, rscor, dbg = orc.new_infer_resolved_quartets(tmparr, tmpmap, q[:120], True, debug=True)
cm = dbg["cmats"][:,0].reshape(-1,256).astype(float)   # mats[0] row-major = bin index a<<6|b<<4|c<<2|d
print("counted per quartet", cm.sum(1).mean())
rng = np.random.default_rng(0)
def sim(bankmap, nl=22, trials=200):
    tot = 0.0
    for p in cm:
        pr = p/p.sum()
        draws = rng.choice(256, size=(trials, nl), p=pr)
        for d in draws:
            u = np.unique(d)                 # same address broadcast/serialised separately: count distinct addresses
            b = bankmap[u]
            tot += np.bincount(b, minlength=32).max() - 1
    return tot/(len(cm)*trials)
bins = np.arange(256)
cur = bins & 31
print("current  extra cycles per 32-lane group:", sim(cur))
# uniform random mapping as a reference for 'ideal hash'
print("random perm:", sim(rng.permutation(256) & 31))
# linear XOR fold: low nibble ^= g4(a,b); bank = (b0, low nibble')
a = bins>>6; b=(bins>>4)&3
best=None
ab = (a<<2)|b
res=[]
for M in itertools.product(range(16), repeat=4):   # columns: image of a1,a0,b1,b0 bits
    g = np.zeros(256,int)
    for bit,col in zip((3,2,1,0), M):
        g ^= np.where((ab>>bit)&1, col, 0)
    bm = ((bins & 16) | ((bins & 15) ^ g)) & 31
    # cheap proxy: sum p^2 over banks averaged over quartets
    pb = np.zeros((len(cm),32))
    for k in range(32): pb[:,k] = cm[:, bm==k].sum(1)
    pb /= pb.sum(1,keepdims=True)
    res.append(((pb**2).sum(1).mean(), M))
res.sort()
print("proxy current:", ((np.stack([cm[:,cur==k].sum(1) for k in range(32)],1)/cm.sum(1,keepdims=True))**2).sum(1).mean())
print("best proxies:", res[:5])
M=res[0][1]
g = np.zeros(256,int)
for bit,col in zip((3,2,1,0), M):
    g ^= np.where((ab>>bit)&1, col, 0)
bm = ((bins & 16) | ((bins & 15) ^ g)) & 31
print("best linear fold extra cycles:", sim(bm))
