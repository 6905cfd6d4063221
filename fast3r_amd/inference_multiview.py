"""`inference()` with the reference's signature and return structure
(fast3r/dust3r/inference_multiview.py:70-99 -> loss_of_one_batch :22-67), driving fast3r_amd.Fast3R.

Precision argument.  The reference turns `dtype` into a torch.autocast context (:41-52): "32" disables autocast (true fp32),
"16-mixed" -> fp16, "bf16-mixed" / torch.bfloat16 -> bf16, and anything else -- notably torch.float32, which the
demo passes -- silently falls through to the *default* autocast dtype.  Here the argument
selects the operand format of the HIP kernels:
    "16-mixed" / torch.float16                                     fp16 operands, the model's `precision`
    "bf16-mixed" / "bf16-mixed-no-grad-scaling" / torch.bfloat16   bf16 operands, the model's `precision`
    "32" / 32                                                      precision="exact": both operands of every GEMM / conv as fp16
                                                                   hi + lo planes (~22 significand bits), attention in fp32 (FMA pipe;
                                                                   from 8192 keys on as three-plane MFMA products with an fp32
                                                                   softmax) -- ~1e-6 of the reference's fp32 path; 3 x the matrix
                                                                   work of the 16-bit modes everywhere (N = 320: 36 s); every model
                                                                   this package builds, view-sharded ones included (one-GPU rank
                                                                   emulation excepted)
    torch.float32                                                  (in the reference: NOT fp32 but the default autocast dtype, SURVEY.md
                                                                   section 0.3) fp16 operands with precision="high" (split weights,
                                                                   split head operands, fp16 attention; DESIGN.md section 3 (Precision modes)); a
                                                                   one-time warning says so
    anything else                                                  the model's own compute_dtype / precision
Accumulation, residual stream, LayerNorm, softmax and outputs are always fp32; the measured distance of every mode to the
reference's true-fp32 CPU path is recorded in DESIGN.md.
"""
import warnings

import numpy as np
import torch

_TENSOR_KEYS = "img pts3d valid_mask camera_pose camera_intrinsics F_matrix corres".split()  # :30-37


def collate_with_cat(whatever, lists=False):
    """fast3r/dust3r/utils/device.py:60-91: recursive collate; tensors are concatenated (or listified)."""
    if isinstance(whatever, dict):
        return {k: collate_with_cat(v, lists=lists) for k, v in whatever.items()}
    if isinstance(whatever, (tuple, list)):
        if len(whatever) == 0:
            return whatever
        elem, T = whatever[0], type(whatever)
        if elem is None:
            return None
        if isinstance(elem, (bool, float, int, str)):
            return whatever
        if isinstance(elem, tuple):
            return T(collate_with_cat(x, lists=lists) for x in zip(*whatever))
        if isinstance(elem, dict):
            return {k: collate_with_cat([e[k] for e in whatever], lists=lists) for k in elem}
        if isinstance(elem, torch.Tensor):
            if not lists and len(whatever) == 1:
                # torch.cat of one tensor: the same values without the copy (N x 8.4 MB of outputs per call otherwise).  The element is
                # returned as it is: for the predictions that is the tensor this call produced; for the views' own fields see ALIAS_HOST_INPUTS
                return elem
            return [x for e in whatever for x in e] if lists else torch.cat(whatever)
        if isinstance(elem, np.ndarray):
            return [x for e in whatever for x in e] if lists else torch.cat([torch.from_numpy(x) for x in whatever])
        return sum(whatever, T())


def to_cpu(x):
    """fast3r/dust3r/utils/device.py:17-53 todevice(x, 'cpu')."""
    if isinstance(x, dict):
        return {k: to_cpu(v) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return type(x)(to_cpu(v) for v in x)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    if torch.is_tensor(x):
        return x.to("cpu")
    return x


def check_if_same_size(imgs):
    shapes = [img["img"].shape[-2:] for img in imgs]  # :102-104
    return all(s == shapes[0] for s in shapes)


# Deviation from the reference, on purpose: inference() returns result["views"][i][name] as THE CALLER'S OWN host tensor when the view came
# from the host (the reference uploads it and `to_cpu` brings back a fresh copy, inference_multiview.py:92-93) -- an in-place edit of the
# result then edits the input.  Set to False to get the reference's fresh copies (one host memcpy of the images: ~1 GB at N = 320).
ALIAS_HOST_INPUTS = True
_warned_fp32 = False
_warned_exact_size = False
_warned_uncalibrated = False
EXACT_WARN_VIEWS = 48  # above this the "exact" mode costs seconds: every product is three MFMAs and the attention is O(T^2) (N = 100: 4 s, N = 320: 36 s)


def _operand_format(precision, model, n_views=0):
    """-> (compute_dtype, precision mode) for this call (see the module docstring).  dtype="32" always runs fp16 hi + lo operand planes
    (the model's compute_dtype is not consulted) with the attention core in fp32."""
    global _warned_fp32, _warned_exact_size
    if precision in ("16-mixed", torch.float16):
        return torch.float16, model.precision
    if precision in ("bf16-mixed", "bf16-mixed-no-grad-scaling", torch.bfloat16):
        return torch.bfloat16, model.precision
    if precision in ("32", 32, torch.float32):
        if precision is not torch.float32:
            if n_views > EXACT_WARN_VIEWS and not _warned_exact_size:
                warnings.warn(f"fast3r_amd.inference(dtype='32') on {n_views} views: the fp32-equivalent mode multiplies every operand as two "
                              "16-bit planes (three MFMAs per product, attention included: quadratic in the number of views, 4 s at 100 views and "
                              "36 s at 320); it is meant for validation.  Build the model with precision='high' and pass dtype='16-mixed' for the "
                              "parity-green production format.", stacklevel=3)
                _warned_exact_size = True
            return torch.float16, "exact"
        if not _warned_fp32:
            warnings.warn(f"fast3r_amd.inference(dtype={precision!r}): running fp16 operands with split hi + lo planes (precision='high': ~22-bit "
                          "weights and head activations, fp16 attention operands, fp32 accumulation); the fp32-equivalent mode is "
                          "dtype='32' (precision='exact').", stacklevel=3)
            _warned_fp32 = True
        return torch.float16, "high"
    return model.compute_dtype, model.precision


def _warn_if_uncalibrated(net):
    """once per process: a LOADED checkpoint running a 16-bit operand tier nobody measured on it.  The 1e-3 parity of fp16 / "high" is established
    on default-init-like and N(0, 1 / fan_in) weights; a noise-amplifying checkpoint measured 2.5e-3 there and needs
    "robust".  Fast3R.calibrate_precision(views[:8]) measures the tiers on the weights actually loaded (~2 s) and silences this."""
    global _warned_uncalibrated
    if _warned_uncalibrated or not getattr(net, "weights_loaded", False) or net.precision == "exact":
        return
    if hasattr(net, "precision_is_calibrated") and not net.precision_is_calibrated():
        _warned_uncalibrated = True
        warnings.warn(f"fast3r_amd.inference: this checkpoint runs with 16-bit operands (precision={net.precision!r}) and was never calibrated: the 1e-3 "
                      "distance to the fp32 reference is measured on default-init-like weights, a noise-amplifying checkpoint can be 2-3 x further out.  "
                      "Run `model.calibrate_precision(views[:8])` once (about 2 s; it reports every tier's distance to the fp32-equivalent mode on THESE "
                      "weights and recommends the cheapest one within 1e-3: 'high' -> 'robust' costs about 2 x), or pass dtype='32'.", stacklevel=4)


def loss_of_one_batch(batch, model, criterion, device, precision, symmetrize_batch=False, use_amp=False, ret=None,
                      profiling=False, host_outputs=False):
    """host_outputs (set by inference(), which returns everything on the CPU anyway): the predictions arrive in pinned host memory through
    the model's overlapped device -> host leg (Fast3R.forward(host_outputs=True)) and the views' tensors that came from the host are
    handed back as they came, so the `to_cpu` of inference() finds nothing left to move."""
    host_side = []  # (view, name, the caller's host tensor): handed back as they came instead of being copied device -> host again
    for view in batch:
        for name in _TENSOR_KEYS:
            if name in view:
                if torch.is_tensor(view[name]) and view[name].device.type == "cpu":
                    host_side.append((view, name, view[name]))
                view[name] = view[name].to(device, non_blocking=True)
    net = getattr(model, "net", model)  # accept the MultiViewDUSt3RLitModule shim too
    saved = (net.compute_dtype, net.precision)
    net.compute_dtype, net.precision = _operand_format(precision, net, n_views=len(batch))
    _warn_if_uncalibrated(net)
    try:
        kw = dict(host_outputs=True) if host_outputs else {}
        out = model(batch, profiling=profiling, **kw) if net is model else (net(batch, profiling=profiling, **kw))
    finally:
        net.compute_dtype, net.precision = saved
    if host_outputs:
        for view, name, t in host_side:
            view[name] = t if ALIAS_HOST_INPUTS else t.clone()
    preds, profiling_info = out if profiling else (out, None)
    loss = criterion(batch, preds) if criterion is not None else None
    result = dict(views=batch, preds=preds, loss=loss)
    if profiling:
        result["profiling_info"] = profiling_info
    return result[ret] if ret else result


@torch.no_grad()
def inference(multiple_views_in_one_sample, model, device, dtype, verbose=True, profiling=False):
    if verbose:
        print(f">> Inference with model on {len(multiple_views_in_one_sample)} images")
    multiple_shapes = not check_if_same_size(multiple_views_in_one_sample)
    res = loss_of_one_batch(collate_with_cat([tuple(multiple_views_in_one_sample)]), model, None, torch.device(device),
                            dtype, profiling=profiling, host_outputs=True)
    profiling_info = res.pop("profiling_info") if profiling and "profiling_info" in res else None
    result = collate_with_cat([to_cpu(res)], lists=multiple_shapes)
    if profiling and profiling_info is not None:
        return result, profiling_info
    return result
