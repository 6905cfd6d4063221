"""distance of the HIP poses to the reference wrapper's (tests/golden/pose_cases.pt) and to the ground truth, per scene (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fast3r_amd import MultiViewDUSt3RLitModule
cases = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pose_cases.pt"), weights_only=False)["cases"]
for c in cases:
    V, B = c["scene"][1], c["scene"][2]
    preds = [{k: v.cuda() for k, v in p.items()} for p in c["preds"]]
    poses, focals = MultiViewDUSt3RLitModule.estimate_camera_poses(preds, niter_PnP=100, focal_length_estimation_method="first_view_from_global_head")
    ref = c["reference"]["first_view_from_global_head"]
    d_ref = max(float(np.abs(poses[b][v] - ref["poses"][b][v]).max()) for b in range(B) for v in range(V))
    d_gt = max(float(np.abs(poses[b][v] - c["gt_cam2world"][v][b].numpy()).max()) for b in range(B) for v in range(V))
    r_gt = max(float(np.abs(ref["poses"][b][v] - c["gt_cam2world"][v][b].numpy()).max()) for b in range(B) for v in range(V))
    print(f"scene {c['scene']}: hip vs reference wrapper {d_ref:.3e}; hip vs ground truth {d_gt:.3e}; reference vs ground truth {r_gt:.3e}", flush=True)
