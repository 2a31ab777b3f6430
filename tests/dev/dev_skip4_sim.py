"""CPU simulation (developer tool):   python tests/dev/dev_skip4_sim.py W H N_TILES
How often could the 4-slot scene's march skip (sphere, box), the sphere alone or the link, exactly, from bounds carried
from the last values each primitive was evaluated at (every leaf is 1-Lipschitz)?  Wave-uniform votes over 8x8 tiles."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import numpy_restatement as R, rm_oracle as O
import cases
from raymarching_amd import Material
f=np.float32
W,H=int(sys.argv[1]),int(sys.argv[2]); NT=int(sys.argv[3])
case=cases.Case("x",W,H)
with Material() as mat:
    case.bind(mat); props=O.props_from_material(mat)
ref=O.render(O.Uniforms(props),W,H,want_steps=True)
sp=ref["steps_p"].reshape(H,W).astype(np.int64)
tiles=sp.reshape(H//8,8,W//8,8)
tmax=tiles.max(axis=(1,3)); tsum=tiles.sum(axis=(1,3))
print("wave-steps total (sum over tiles of max steps):", tmax.sum(), " lane-steps", sp.sum())
sc=R.Scene(props)
dx,dy,dz=(a.reshape(H,W) for a in R.pixel_rays(props,W,H))
cam=np.asarray(props["_WorldSpaceCameraPos"],dtype=f)
far=f(np.asarray(props["_ProjectionParams"],dtype=f)[2]); depth=np.fmin(far,f(props["_MaxDist"]))
steps=int(props["_MaxSteps"]); sf=f(props["_StepFactor"])
def run(ty,tx):
    sl=(slice(ty*8,ty*8+8),slice(tx*8,tx*8+8))
    d=[a[sl].ravel() for a in (dx,dy,dz)]
    rdlen=np.sqrt(d[0].astype(np.float64)**2+d[1].astype(np.float64)**2+d[2].astype(np.float64)**2)
    dm=np.zeros(64,dtype=f); act=np.ones(64,bool)
    # bounds per primitive
    lb=np.full((4,64),-np.inf); ub=np.full((4,64),np.inf)
    cnt=dict(evals=0,pair01=0,o0=0,o2=0)
    cost=0.0
    for i in range(steps):
        if not act.any(): break
        px,py,pz=(cam[j]+d[j]*dm for j in range(3))
        o=np.stack([sc.prim(p,px,py,pz) for p in sc.prims]).astype(np.float64)
        a=act
        # decisions (wave-uniform over active lanes)
        lb01=np.minimum(lb[0],lb[1])-0.126; ub23=np.minimum(ub[2],ub[3])
        s01=np.all((ub23+0.51<=lb01)[a])
        s0=(not s01) and np.all((ub[1]+0.51<=lb[0])[a])
        s2=np.all((ub[3]+0.51<=lb[2])[a])
        cnt["evals"]+=1; cnt["pair01"]+=s01; cnt["o0"]+=s0; cnt["o2"]+=s2
        c=100.0+12
        if s01: c-=13+16+18
        elif s0: c-=13+9
        if s2: c-=20+9
        cost+=c
        _,v=sc.eval(px,py,pz)
        adts=np.abs(v)
        step=adts*sf
        newdm=np.where(a,dm+step,dm)
        delta=(np.abs(step.astype(np.float64))*rdlen)*1.0001+1e-4
        ev=[not (s01 or s0), not s01, not s2, True]
        for k in range(4):
            if ev[k]: lb[k]=o[k]-delta; ub[k]=o[k]+delta
            else: lb[k]=lb[k]-delta; ub[k]=ub[k]+delta
        dm=newdm
        stop=a&((adts<dm*f(.0025))|(dm>depth))
        act=a&~stop
    return cnt,cost
# heaviest tiles and random tiles
order=np.dstack(np.unravel_index(np.argsort(-tmax,axis=None),tmax.shape))[0]
for label,sel in (("heaviest",order[:NT]),("random",order[np.random.default_rng(0).integers(0,len(order),NT)])):
    tot=dict(evals=0,pair01=0,o0=0,o2=0); tc=0
    for ty,tx in sel:
        c,cost=run(int(ty),int(tx))
        for k in tot: tot[k]+=c[k]
        tc+=cost
    print(label,tot,"relative VALU cost with the tests (12 instr per step) vs 100 per step: %.3f"%(tc/(100.0*tot["evals"])))
