"""Round 5 probe: step_finalize_ws with different numbers of parameter quads per block (VK_FIN_QUADS builds under vmap_amd/_exp/)."""
import sys, glob, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from vmap_amd import step, synth, _lib
DEV="cuda:0"
def run(lib, name, weights="f32", steps=400):
    cfg = synth.CONFIGS[name]; n,R,S,H = cfg["n_obj"],cfg["R"],cfg["S"],cfg["H"]; ipf=20
    fc,B,sc = synth.make_params(n,H,scale=cfg["scale"],seed=5)
    fr0 = synth.make_batch(n,R*ipf,S,seed=6)
    tfc=[torch.from_numpy(a).to(DEV) for a in fc]; tB=torch.from_numpy(B).to(DEV); tsc=torch.from_numpy(sc).to(DEV)
    fr = tuple(torch.from_numpy(fr0[k]).to(DEV) for k in ("pcs","z","gt_depth","gt_rgb","sem","depth_mask"))
    op = step.VmapStep(n,R,S,H,device=DEV,max_steps=ipf,weights=weights,library=lib)
    opt = step.FusedAdamWState(n,H,DEV,lr=1e-3,weight_decay=0.013)
    b = op.bind(tfc,tB,tsc,*fr,opt=opt)
    for _ in range(3): b.train_steps(ipf)
    torch.cuda.synchronize()
    ts=[]
    for rep in range(3):
        t0=time.perf_counter()
        for _ in range(steps//ipf): b.train_steps(ipf)
        torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/steps*1e3)
    return sorted(ts)[1], [t.cpu() for t in tfc+[tB]]
libs = [None] + sorted(glob.glob('/root/repo/vmap_amd/_exp/libq*.so'))
for name, w in (("background","f32"),):
    ref=None
    for lib in libs:
        try:
            ms, params = run(lib, name, w, steps=100 if name in ("stress_256x64","imap_full") else 400)
        except Exception as e:
            print(name, lib, "ERROR", str(e)[:100]); continue
        if ref is None: ref=params
        same = all(torch.equal(a,b) for a,b in zip(params,ref))
        print(f"{name:16s} {w} {'product (128 quads)' if lib is None else lib.split('/')[-1]:22s} ms/step {ms:.5f}  bit-identical parameters to the product: {same}")
