import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/texture-gs_amd', '/root/repo/tests']
import torch, numpy as np
from texgs import synth
import helpers as Hh
from oracle import texgs_ref as CR
N, R, W, H, sm, deg, view, bg = (4000, 128, 200, 136, 0.02, 1, 1, (0.0, 0.0, 0.0))
scene = synth.make_scene(N, R, seed=N + R, scale_mean=sm, random_jacobian=True)
cam = synth.fibonacci_cameras(4, W, H)[view]
bg = torch.tensor(bg)
ref, dbg, _ = Hh.oracle_run(scene, cam, deg, bg)
run = CR.RefRun(scene, Hh.settings_for(cam, deg, bg)); cout = torch.tensor(run.forward()).double()
amb = dbg["ambiguity"]
exp = torch.cat([ref[0], ref[1], ref[2], ref[3]], 0).double()
if torch.cuda.is_available():
    out, _ = Hh.hip_run(scene, cam, deg, bg)
    got = torch.cat([out[0], out[1], out[2], out[3]], 0).detach().cpu().double()
else:
    got = cout
for nm, g in (("hip", got), ("c32", cout)):
    err = (g - exp).abs()[:3].amax(0)
    err_m = torch.where(amb < 1e-4, torch.zeros_like(err), err)
    idx = torch.topk(err_m.flatten(), 8).indices
    print(nm, "pixels>1e-4 (unamb):", int((err_m > 1e-4).sum()), "max", float(err_m.max()), "n>5e-5", int((err_m > 5e-5).sum()))
    for i in idx.tolist():
        y, x = divmod(i, W)
        print("  px", (x, y), "err %.3e" % float(err_m[y, x]), "amb %.3e" % float(amb[y, x]), "n_contrib", int(dbg["n_contrib"][y, x]),
              "alpha_err %.2e" % float((g - exp).abs()[7, y, x]), "rgb", [round(float(v), 4) for v in exp[:3, y, x]])
