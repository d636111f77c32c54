"""Round 5 probe: run-to-run repeatability of the hidden-64 step (step_main_wp<2>) by workgroups per CU - nothing differs while every
workgroup has a CU to itself (<= 256 workgroups), a miscompiled prologue showed up as soon as two shared one (HISTORY.md, round 5)."""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from vmap_amd import step, synth
DEV="cuda:0"
def trial(n, R, wpo, label, reps=5):
    S,H=10,64
    fc0,B0,sc0 = synth.make_params(n,H,seed=3)
    fr0 = synth.make_batch(n,R,S,seed=4)
    fr = {k: torch.from_numpy(v).to(DEV) for k,v in fr0.items()}
    keys=("pcs","z","gt_depth","gt_rgb","sem","depth_mask")
    ref=None; nd=0
    for rep in range(reps):
        fc=[torch.from_numpy(a).to(DEV) for a in fc0]; B=torch.from_numpy(B0).to(DEV); sc=torch.from_numpy(sc0).to(DEV)
        op = step.VmapStep(n,R,S,H,device=DEV,max_steps=1,tuning={"workgroups_per_object": wpo} if wpo else None)
        gfc=[torch.zeros_like(t) for t in fc]; gB=torch.zeros_like(B)
        r = op.fwd_bwd(fc,B,sc,*(fr[k] for k in keys),grads_fc=gfc,grad_B=gB,render=True)
        torch.cuda.synchronize()
        out=[r.loss.cpu(), r.render_depth.cpu(), r.render_color.cpu()]+[g.cpu() for g in gfc]
        if ref is None: ref=out
        else: nd += int(not all(torch.equal(x,y) for x,y in zip(out,ref)))
    p=op.plan()
    print(label, "n",n,"R",R,"NW",p["workgroups_per_object"],"rounds",p["rounds_per_object"],"WGs",n*p["workgroups_per_object"],"-> runs differing from the first:",nd,"of",reps-1)
trial(32,256,0,"multi-round, 480 WGs (2 per CU)")
trial(32,256,43,"single-round, 1376 WGs")
trial(8,256,15,"multi-round, 120 WGs (<= 1 per CU?)")
trial(16,256,15,"multi-round, 240 WGs")
trial(17,256,15,"multi-round, 255 WGs")
trial(20,256,15,"multi-round, 300 WGs (some CUs hold 2)")
trial(8,256,43,"single-round, 344 WGs (some CUs hold 2)")
