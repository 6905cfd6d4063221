"""fast3r_amd: the Fast3R single-forward-pass inference hot path, MI355X-native (hand-written HIP for gfx950 behind
the reference's Python API).  See DESIGN.md."""
from .fast3r import Fast3R  # noqa: F401
from .inference_multiview import inference  # noqa: F401
from .multiview_dust3r_module import MultiViewDUSt3RLitModule  # noqa: F401
from .align import align_local_pts3d_to_global  # noqa: F401
from .focal import estimate_focal, estimate_focals  # noqa: F401
from .pose import estimate_camera_poses, estimate_poses  # noqa: F401
from .image import load_images  # noqa: F401
