"""Inference-only shim of MultiViewDUSt3RLitModule (fast3r/models/multiview_dust3r_module.py:67-126).

The reference class is a LightningModule whose training / validation / metric machinery is outside the MI355X hot
path.  What inference callers use is kept with the same names:
    lit = MultiViewDUSt3RLitModule.load_for_inference(net)   (:119-123)
    lit.eval(); lit(views) == net(views)                     (:125-126)
`align_local_pts3d_to_global` (:427-549) runs on the GPU (fast3r_amd/align.py);
`estimate_focal` (:1081-1109) and `estimate_camera_poses` (:807-869) run on the GPU (fast3r_amd/focal.py, fast3r_amd/pose.py;
SURVEY.md section 8f rank 2; the PnP solver is not OpenCV's -- see pose.py).
"""
import torch

from .align import align_local_pts3d_to_global as _align
from .focal import estimate_focal, estimate_focals  # noqa: F401  (module-level in the reference too, :1081)
from .pose import estimate_camera_poses as _estimate_camera_poses


class MultiViewDUSt3RLitModule(torch.nn.Module):
    def __init__(self, net, train_criterion=None, validation_criterion=None, optimizer=None, scheduler=None,
                 compile=False, pretrained=None, resume_from_checkpoint=None, eval_use_pts3d_from_local_head=True):
        super().__init__()
        self.net = net
        self.train_criterion, self.validation_criterion = train_criterion, validation_criterion
        self.pretrained, self.resume_from_checkpoint = pretrained, resume_from_checkpoint
        self.eval_use_pts3d_from_local_head = eval_use_pts3d_from_local_head

    @classmethod
    def load_for_inference(cls, net):
        lit_module = cls(net=net, train_criterion=None, validation_criterion=None, optimizer=None, scheduler=None, compile=False)
        lit_module.eval()
        return lit_module

    def forward(self, views, **kw):
        return self.net(views, **kw)

    @staticmethod
    def estimate_camera_poses(preds, views=None, niter_PnP=10, focal_length_estimation_method="individual"):
        """Reference :807-869: returns (poses_c2w_all, estimated_focals_all), per sample and per view; preds must be on the GPU."""
        return _estimate_camera_poses(preds, views, niter_PnP, focal_length_estimation_method)

    def align_local_pts3d_to_global(self, preds, views, min_conf_thr_percentile=0):
        """Adds `pts3d_local_aligned_to_global` to every pred (reference :427-549); preds must be on the GPU."""
        _align(preds, views, min_conf_thr_percentile)
