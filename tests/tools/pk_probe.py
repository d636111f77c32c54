"""Round 5 probe: the hidden-64 kernel with the ray prologue (the source that was not repeatable), built with and without the SLP
vectoriser (= with and without packed-float32 instructions): run-to-run repeatability when two workgroups share a CU.
The libraries it loads are experiment builds (not kept): vmap_amd/_exp/pk/lib_<name>.so = the product's objects with k_wp.o replaced by a build of
k_wp.hip as it was compiled THEN - bad: plain `hipcc -c`; noslp: + -fno-slp-vectorize; badfixed: through csrc/gfx950_errata.compile_unit.  Since
build() routes every unit through the erratum pass, "bad" can only be rebuilt with a bare `hipcc -c` of that unit.  Results: profiles/round5i_*."""
import sys, json, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from vmap_amd import step, synth
DEV="cuda:0"
def trial(lib, n, R, wpo, reps):
    S,H=10,64
    fc0,B0,sc0 = synth.make_params(n,H,seed=3)
    fr0 = synth.make_batch(n,R,S,seed=4)
    fr = {k: torch.from_numpy(v).to(DEV) for k,v in fr0.items()}
    keys=("pcs","z","gt_depth","gt_rgb","sem","depth_mask")
    ref=None; nd=0
    for rep in range(reps):
        fc=[torch.from_numpy(a).to(DEV) for a in fc0]; B=torch.from_numpy(B0).to(DEV); sc=torch.from_numpy(sc0).to(DEV)
        op = step.VmapStep(n,R,S,H,device=DEV,max_steps=1,tuning={"workgroups_per_object": wpo} if wpo else None, library=lib)
        gfc=[torch.zeros_like(t) for t in fc]; gB=torch.zeros_like(B)
        r = op.fwd_bwd(fc,B,sc,*(fr[k] for k in keys),grads_fc=gfc,grad_B=gB,render=True)
        torch.cuda.synchronize()
        out=[r.loss.cpu(), r.render_depth.cpu(), r.render_color.cpu()]+[g.cpu() for g in gfc]
        if ref is None: ref=out
        else: nd += int(not all(torch.equal(x,y) for x,y in zip(out,ref)))
    return nd
res=[]
for name in sys.argv[1:]:
    lib = None if name=="product" else f"/root/repo/vmap_amd/_exp/pk/lib_{name}.so"
    for (n,R,wpo) in ((32,256,0),(32,256,43),(20,256,15),(16,256,15)):
        nd=trial(lib,n,R,wpo,13)
        rec={"library":name,"n":n,"R":R,"wpo":wpo,"runs_differing_from_first":nd,"of":12}
        print(json.dumps(rec),flush=True); res.append(rec)
