"""Developer experiment (CPU, fp64): how fast do VARIANTS of projected Gauss-Seidel approach the fixed point of the substep's contact problem?
("reduce the solver residual in the warm state".)  States: random contact configurations of
tests/test_contact_lcp_reference.py after four product substeps (persistent contacts); rows: tests/physics_ref.py.  Variants:
per-corner friction (the spec) vs PATCH friction for the cube-floor / cube-wall contact (4 normals + 2 anchor tangents + torsion), a
3x3 BLOCK update of every finger-cube contact (joint solve of normal + tangents, sequential projection when infeasible), successive
over-relaxation, symmetric / floor-first / repeated-block row orders.  Error = scaled velocity error of k sweeps against the variant's
own fixed point (4000 sweeps).   python tests/dev/pgs_variants.py [cases]      result of round 3: profiles/r3_b_pgs_variants.txt"""
import sys, os
REPO=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,os.path.join(REPO,'tests')); sys.path.insert(0,REPO)
import numpy as np, torch
import test_contact_lcp_reference as L
import physics_ref as PR, test_physics_analytic as T
from oracle_util import load_oracle
from multiprocessing import Pool
H=L.H

def build_rows(q,qd,cube,tau,h,patch):
    """returns (Minv, v_start(after ff), rows list of dict) reusing PR internals by calling ref_substep with max_sweeps=0-like trick"""
    # run the reference to its fixed point to get rows (it returns rows with converged lam); we rebuild our own solver from rows' J/bias
    v=PR.ref_substep(q,qd,cube,tau,h,max_sweeps=50000)
    return v

def pgs(rows, Minv, v0, sweeps, block=False, groups=None, omega=1.0, order=None):
    v=v0.copy(); lam=[0.0]*len(rows)
    W=[Minv@r['J'] for r in rows]; D=[float(r['J']@w) for r,w in zip(rows,W)]
    for it in range(sweeps):
        i=0
        while i<len(rows):
            r=rows[i]
            if block and r['kind']=='normal' and r.get('block3'):
                # 3x3 block: rows i,i+1,i+2
                idx=[i,i+1,i+2]
                J=np.array([rows[k]['J'] for k in idx]); A=J@Minv@J.T
                l0=np.array([lam[k] for k in idx]); vrel=J@v
                rhs=-(vrel+np.array([r['bias'],0,0]))
                # unconstrained (sticking) solution
                dl=np.linalg.solve(A,rhs); ln=l0+dl
                ok = ln[0]>=0 and abs(ln[1])<=r_mu(rows[i+1])*ln[0] and abs(ln[2])<=r_mu(rows[i+2])*ln[0]
                if not ok:
                    # fall back: sequential projected rows
                    ln=l0.copy(); vv=v.copy()
                    for kk,k in enumerate(idx):
                        rr=rows[k]; d=D[k]; vr=float(rr['J']@vv)
                        if rr['kind']=='normal': new=max(ln[kk]-(vr+rr['bias'])/d,0.0)
                        else:
                            lim=rr['mu']*ln[0]; new=float(np.clip(ln[kk]-vr/d,-lim,lim))
                        vv=vv+W[k]*(new-ln[kk]); ln[kk]=new
                for kk,k in enumerate(idx):
                    v=v+W[k]*(ln[kk]-lam[k]); lam[k]=ln[kk]
                i+=3; continue
            d=D[i]
            if d>0:
                vr=float(r['J']@v)
                if r['kind']=='normal': new=max(lam[i]-omega*(vr+r['bias'])/d,0.0)
                elif r['kind']=='tangent':
                    lim=r['mu']*sum(lam[k] for k in r['parents']); new=float(np.clip(lam[i]-omega*vr/d,-lim,lim))
                else:
                    v0_=vr-d*lam[i]; new=(float(np.clip(v0_,r['lo'],r['hi']))-v0_)/d
                v=v+W[i]*(new-lam[i]); lam[i]=new
            i+=1
    return v
def r_mu(r): return r['mu']
def pgs2(rows, Minv, v0, sweeps, mode):
    v=v0.copy(); lam=[0.0]*len(rows)
    W=[Minv@r['J'] for r in rows]; D=[float(r['J']@w) for r,w in zip(rows,W)]
    n=len(rows)
    # contact triplets must keep normal before its tangents: build units
    units=[]; i=0
    while i<n:
        if rows[i]['kind']=='normal': units.append([i,i+1,i+2]); i+=3
        else: units.append([i]); i+=1
    def is_floor(u): return rows[u[0]]['kind']=='normal' and not np.any(rows[u[0]]['J'][:9]!=0)
    def is_fc(u): return rows[u[0]]['kind']=='normal' and np.any(rows[u[0]]['J'][:9]!=0) and np.any(rows[u[0]]['J'][9:]!=0)
    for it in range(sweeps):
        if mode=='sym': order=units if it%2==0 else units[::-1]
        elif mode=='floorfirst': order=[u for u in units if is_floor(u)]+[u for u in units if not is_floor(u)]
        elif mode=='cube2x': # floor rows both before and after fc rows
            fl=[u for u in units if is_floor(u)]; order=fl+[u for u in units if not is_floor(u)]+fl
        elif mode=='fc2x':
            fc=[u for u in units if is_fc(u)]; order=units+fc
        else: order=units
        for u in order:
            for i in u:
                r=rows[i]; d=D[i]
                if d<=0: continue
                vr=float(r['J']@v)
                if r['kind']=='normal': new=max(lam[i]-(vr+r['bias'])/d,0.0)
                elif r['kind']=='tangent':
                    lim=r['mu']*sum(lam[k] for k in r['parents']); new=float(np.clip(lam[i]-vr/d,-lim,lim))
                else:
                    v0_=vr-d*lam[i]; new=(float(np.clip(v0_,r['lo'],r['hi']))-v0_)/d
                v=v+W[i]*(new-lam[i]); lam[i]=new
    return v

def convert(det_rows, patch):
    """PR Row objects -> dict rows; with patch=True the corner friction rows of cube-floor (dirs ex,ey) are replaced by 3 patch rows"""
    rows=[]; idx_of={}
    R=det_rows
    for i,r in enumerate(R):
        d=dict(J=r.J.copy(),kind=r.kind,bias=r.bias,mu=r.mu,lo=r.lo,hi=r.hi)
        if r.kind=='tangent': d['parents']=[R.index(r.parent)]
        rows.append(d)
    # mark finger-cube blocks: normal rows whose J touches both joint dofs and cube dofs
    for i,r in enumerate(rows):
        if r['kind']=='normal' and np.any(r['J'][:9]!=0) and np.any(r['J'][9:]!=0): r['block3']=True
    if not patch: return rows
    # identify cube-only contacts (floor or wall corners): normal rows with zero joint part and nonzero cube part
    corner=[i for i,r in enumerate(rows) if r['kind']=='normal' and not np.any(r['J'][:9]!=0)]
    # group by normal direction (floor: J[9:12]=(0,0,1); wall: horizontal)
    groups={}
    for i in corner:
        key='floor' if abs(rows[i]['J'][11]-1.0)<1e-9 else 'wall'
        groups.setdefault(key,[]).append(i)
    drop=set(); extra=[]
    for key,idxs in groups.items():
        for i in idxs: drop.update([i+1,i+2])
        # anchor: centroid of contact arms. arm r from J: J[12:15] = r x n  -> recover r? we stored cube_map: J = n @ [I, -[r]x] ; easier: recompute from tangents
        # reconstruct r for each corner from the three rows (n,t1,t2): angular parts a_d = r x d. r = sum_d d x a_d /... use least squares
        arms=[]
        for i in idxs:
            Dm=np.array([rows[i+k]['J'][9:12] for k in range(3)]); Am=np.array([rows[i+k]['J'][12:15] for k in range(3)])
            # a_d = r x d  => for orthonormal d's: r = 0.5*sum_d d x a_d
            r=0.5*sum(np.cross(Dm[k],Am[k]) for k in range(3)); arms.append(r)
        rc=np.mean(arms,axis=0); n=rows[idxs[0]]['J'][9:12]
        if key=='wall': n=np.mean([rows[i]['J'][9:12] for i in idxs],axis=0); n/=np.linalg.norm(n)
        t1,t2=PR.tangent_basis(n); mu=rows[idxs[0]+1]['mu']
        for t in (t1,t2):
            J=np.zeros(15); J[9:12]=t; J[12:15]=np.cross(rc,t)
            extra.append(dict(J=J,kind='tangent',bias=0.0,mu=mu,parents=list(idxs),lo=0,hi=0))
        rt=np.mean([np.linalg.norm(a-rc) for a in arms])*(2.0/3.0)
        J=np.zeros(15); J[12:15]=n
        extra.append(dict(J=J,kind='tangent',bias=0.0,mu=mu*rt,parents=list(idxs),lo=0,hi=0))
    # rebuild list with index remap
    keep=[i for i in range(len(rows)) if i not in drop]
    remap={o:nw for nw,o in enumerate(keep)}
    out=[]
    for o in keep:
        r=rows[o]
        if r['kind']=='tangent': r['parents']=[remap[p] for p in r['parents']]
        out.append(r)
    # insert extra rows before limit rows
    first_lim=next((i for i,r in enumerate(out) if r['kind']=='limit'),len(out))
    for e in extra: e['parents']=[remap[p] for p in e['parents']]
    out=out[:first_lim]+extra+out[first_lim:]
    # parents indices unaffected for those < first_lim (extras inserted after all contacts)
    return out

def scaled(v,vs):
    return max(np.abs(v[:9]-vs[:9]).max()/10, np.abs(v[9:12]-vs[9:12]).max(), np.abs(v[12:15]-vs[12:15]).max()/20)

def one(seed):
    torch.set_num_threads(1)
    lib=load_oracle(); rng=np.random.default_rng(seed)
    while True:
        q,qd,cube,tau=L.make_case(rng)
        eng=T.engine(lib,device='cpu',dt=H,substeps=1,solver_iterations=8)
        f32=dict(dtype=torch.float32)
        eng.q[:,0]=torch.tensor(q,**f32); eng.qd[:,0]=torch.tensor(qd,**f32); eng.cube[:,0]=torch.tensor(cube,**f32); eng.tau[:,0]=torch.tensor(tau,**f32)
        for _ in range(4): eng.simulate()
        st=eng.state[:,0].numpy().astype(np.float64); eng.close()
        try: ref=PR.ref_substep(st[0:9],st[9:18],st[18:31],tau,H,max_sweeps=50000)
        except ValueError: continue
        det=ref[3]
        fc=[x for x in det['fc'] if x[3].lam>0]; te=[x for x in det['te'] if x[3].lam>0]
        if det['sweeps']>=50000 or not (fc or te): continue
        break
    # Minv and v_start: recompute as in ref (hack: re-run pieces)
    qq,qdd,cb=st[0:9],st[9:18],st[18:31]
    Minv=np.zeros((15,15)); vfree=np.zeros(15)
    for f in range(3):
        sl=slice(3*f,3*f+3); M=PR.kinetic_matrix(qq[sl]); Mi=np.linalg.inv(M); Minv[sl,sl]=Mi
        acc=Mi@(tau[sl]-PR.bias_forces(qq[sl],qdd[sl],-9.81)); vfree[sl]=(qdd[sl]+H*acc)*(1-H*PR.LINK_DAMP)
    Minv[9:12,9:12]=np.eye(3)/PR.CUBE_MASS; Minv[12:15,12:15]=np.eye(3)/PR.CUBE_INERTIA
    vfree[9:12]=(cb[7:10]+H*np.array([0,0,-9.81])); vfree[12:15]=cb[10:13]*(1-H*PR.CUBE_ANG_DAMP)
    # v after ff pass: ref applies ff impulses: v_ff = vfree + sum; approximate by using details: recompute from ff list is hard -> ignore cases with ff
    if det['ff']: return None
    out={}
    for name,patch,block in (('corner friction (spec)',False,False),('corner + 3x3 block',False,True),('patch friction',True,False),('patch + 3x3 block',True,True)):
        rows=convert(det['rows'],patch)
        vs=pgs(rows,Minv,vfree,4000,block=False)
        for k in (8,16):
            out[(name,k)]=scaled(pgs(rows,Minv,vfree,k,block=block),vs)
    rows=convert(det['rows'],False)
    vstar=pgs(rows,Minv,vfree,4000)
    for om in (1.2,1.35,1.5):
        out[('SOR omega %.2f'%om,8)]=scaled(pgs(rows,Minv,vfree,8,omega=om),vstar)
    for mode in ('plain','sym','floorfirst','cube2x','fc2x'):
        for k in (8,):
            out[(mode,k)]=scaled(pgs2(rows,Minv,vfree,k,mode),vstar)
    out[('plain',10)]=scaled(pgs2(rows,Minv,vfree,10,'plain'),vstar)
    out[('plain',12)]=scaled(pgs2(rows,Minv,vfree,12,'plain'),vstar)
    out['nfloor']=det['n_floor']; out['nfc']=len(fc)
    return out
if __name__=='__main__':
    n=int(sys.argv[1])
    with Pool(8) as p: R=[r for r in p.map(one,range(2000,2000+n)) if r]
    print(len(R),'cases')
    for key in [k for k in R[0] if isinstance(k,tuple)]:
        e=np.array([r[key] for r in R]); print(key,'median %.2e p90 %.2e p99 %.2e max %.2e'%(np.median(e),np.percentile(e,90),np.percentile(e,99),e.max()))
    sel=[r for r in R if r['nfloor']==4 and r['nfc']>=1]
    print('chains (fc + 4 floor corners):',len(sel))
    for key in [k for k in R[0] if isinstance(k,tuple)]:
        e=np.array([r[key] for r in sel]); print(key,'median %.2e p90 %.2e max %.2e'%(np.median(e),np.percentile(e,90),e.max()))
