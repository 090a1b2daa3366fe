"""Diagnostic (not a test): print HIP-vs-oracle errors for the parity cases."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "texture-gs_amd"), os.path.join(ROOT, "tests")]
from texgs import synth
import helpers as Hh
from test_parity_gpu import CASES, _scene

for case in CASES:
    scene, cam, deg, bg = _scene(case)
    target, nhat = synth.make_targets(cam.image_height, cam.image_width, seed=5)
    ref, dbg, gref = Hh.oracle_run(scene, cam, deg, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
    out, ggot = Hh.hip_run(scene, cam, deg, bg, with_grad=True, target=target, nhat=nhat, depth_weight=0.05)
    print("case", case, "D", dbg["binning"]["D"])
    for thr in (1e-3, 1e-4, 1e-5):
        amb = dbg["ambiguity"] < thr
        line = f"  amb<{thr:g}: frac {float(amb.float().mean()):.5f} |"
        for k, name in enumerate(["image", "depth", "norm", "alpha"]):
            err = (out[k].detach().cpu().double() - ref[k].double()).abs()
            clean = err[:, ~amb]
            line += f" {name} clean {float(clean.max()):.2e} all {float(err.max()):.2e} n>1e-4 {int((clean > 1e-4).sum())} |"
        print(line)
    print("  radii mismatches", int((out[4].cpu().long() != ref[4].long()).sum()))
    for n, e in gref.items():
        g = ggot[n]
        if e is None: continue
        print(f"  grad {n:10s} rel {Hh.rel_err(g, e):.3e}  max|ref| {float(e.abs().max()):.3e} max|diff| {float((g.double()-e).abs().max()):.3e}")
