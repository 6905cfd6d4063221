"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (fast3r_amd/).

Imports the *real* reference (facebookresearch/fast3r, mounted read-only at
/root/reference) on CPU so that
  * oracle/fast3r_oracle.py (our CPU restatement) can be pinned against it, and
  * oracle/make_golden.py can generate the fixtures under tests/golden/.

/root/reference only exists in the build container, never on the GPU box, so
every caller must check `reference_available()` first.

The reference pulls in seven packages that are not installed here
(SURVEY.md section 8c): omegaconf, torchvision, cv2, lightning,
lightning_utilities, hydra, pillow_heif.  None of them is used by the hot path
(fast3r/models/fast3r.py:302-497); they are only touched at import time, so a
meta-path finder that hands out permissive dummy modules is enough.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FAST3R_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = (
    "omegaconf",
    "torchvision",
    "cv2",
    "lightning",
    "lightning_utilities",
    "hydra",
    "pillow_heif",
)


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "fast3r"))


class _Dummy:
    """Permissive stand-in: callable, subclassable, attribute-chaining."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        # manufacture a class so that `class X(stub.Base)` and `stub.f(...)` both work
        cls = type(name, (_Dummy,), {})
        setattr(self, name, cls)
        return cls


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


_installed = False


def install():
    """Make `import fast3r...` resolve to the read-only reference checkout."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    sys.meta_path.append(_StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    # fast3r/utils/pylogger.py:46-48 raises if rank_zero_only.rank is None
    import lightning_utilities.core.rank_zero as rz  # noqa: the stub

    rz.rank_zero_only.rank = 0
    _installed = True


def load_reference():
    """Returns (Fast3R class, inference function) of the reference."""
    install()
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):  # pos_embed.py:127-129 prints a warning
        from fast3r.models.fast3r import Fast3R
        from fast3r.dust3r.inference_multiview import inference
    return Fast3R, inference
