"""
ComfyUI custom-node entry point for the MI355X FP8 backend.

Counterpart of the reference's plugin entry (__init__.py:13-61): dropping this
directory into ComfyUI/custom_nodes/ installs the monkey-patch at import time,
so FLUX / SD3.5 FP8 call sites (torch._scaled_mm, Tensor.to(float8_e4m3fn),
Tensor.copy_) run on the hand-written gfx950 kernels without any change to
ComfyUI.  As in the reference, a failure to install is reported and swallowed
so that ComfyUI still starts (__init__.py:43-53), and the node mappings are
empty (:57-61) - this plugin adds no nodes.
"""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)

try:
    import fp8_mps_patch

    if not fp8_mps_patch.is_installed():
        fp8_mps_patch.install()
    print("[fp8-mi355x] FP8 e4m3fn patch installed (torch._scaled_mm, Tensor.to, Tensor.copy_ -> gfx950 HIP kernels)")
except Exception as exc:  # pragma: no cover - mirrors the reference's swallow-and-print
    print(f"[fp8-mi355x] WARNING: could not install the FP8 patch: {exc}")

NODE_CLASS_MAPPINGS = {}
NODE_DISPLAY_NAME_MAPPINGS = {}
__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
