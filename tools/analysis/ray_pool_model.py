# tools/analysis/ray_pool_model.py — development analysis (round 6): would a WORKGROUP-LEVEL RAY POOL (every lane without a ray pulls one;
# results go back to the owner wave) beat the lock-step walk?  Abstract model: rays = jobs of r walk rounds, r from walk_sim's histogram of the headline frame;
# 16 waves alternate between a walk phase and ~8.7 rounds of other work.  Answer: no — the owner waits for its longest ray anyway; lanes are 21 % busy.
# abstract Monte Carlo of a workgroup-level ray pool (16 waves x 64 lanes) vs the lock-step walk, on walk_sim's per-ray round histogram
import numpy as np
rng=np.random.default_rng(1)
cam=np.array([26.08,6.99,15.26,12.87,18.05,8.59,6.77,2.46,1.33,0.59,0.59,0.29,0.07,0.04,0.01,0.01]); bou=np.array([14.22,15.66,30.55,20.83,9.90,3.87,2.47,1.08,0.55,0.31,0.20,0.13,0.08,0.05,0.03,0.02,0.02,0.01,0.01,0.01])
def draw(n):
    k=rng.random(n)<0.379
    a=rng.choice(len(cam),n,p=cam/cam.sum()); b=rng.choice(len(bou),n,p=bou/bou.sum())
    return np.where(k,a,b)
# baseline: max over 62 lanes
m=[draw(62).max() for _ in range(20000)]
print("baseline rounds per wave iteration (iid model): %.2f (walk_sim, correlated: 6.16)"%np.mean(m), "mean per ray %.2f"%draw(200000).mean())
def sim(W=16, other=8.7, T=20000, pull_policy="until_own_done", lanes=64, active=62):
    # time in rounds. wave states: 'other' with remaining time, or 'walk'
    t_other=[rng.uniform(0,other) for _ in range(W)]
    state=['other']*W
    hold=[np.zeros(lanes,int) for _ in range(W)]      # remaining rounds of the ray each lane holds (0 = none)
    owner=[np.full(lanes,-1) for _ in range(W)]
    pending=[0]*W
    pool=[]   # list of (owner, rounds)
    walk_rounds=[0]*W; iters=[0]*W; busy_lane_rounds=0; total_lane_rounds=0
    for tick in range(T):
        for w in range(W):
            if state[w]=='other':
                t_other[w]-=1
                if t_other[w]<=0:
                    r=draw(active); r=r[r>0]
                    pending[w]=len(r)
                    for x in r: pool.append((w,int(x)))
                    state[w]='walk'; iters[w]+=1
        # walking waves do one round each (random order for fairness)
        for w in rng.permutation(W):
            if state[w]!='walk': continue
            can_pull = pending[w]>0 or pull_policy=="always"
            if can_pull:
                for l in np.where(hold[w]==0)[0]:
                    if not pool: break
                    o,x=pool.pop(0); hold[w][l]=x; owner[w][l]=o
            act=hold[w]>0
            if act.any() or pending[w]>0:
                walk_rounds[w]+=1; busy_lane_rounds+=act.sum(); total_lane_rounds+=lanes
                hold[w][act]-=1
                done=act&(hold[w]==0)
                for l in np.where(done)[0]:
                    pending[owner[w][l]]-=1; owner[w][l]=-1
            if pending[w]==0 and not (hold[w]>0).any():
                state[w]='other'; t_other[w]=other
    it=sum(iters); 
    return sum(walk_rounds)/it, busy_lane_rounds/max(1,total_lane_rounds)
for pol in ("until_own_done","always"):
    r,u=sim(pull_policy=pol)
    print("pooled (%s): walk rounds per wave iteration %.2f, lane utilisation in walk rounds %.2f"%(pol,r,u))
