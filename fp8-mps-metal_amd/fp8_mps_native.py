"""
Name-compatibility alias: the reference's op module is called fp8_mps_native
(fp8_mps_native.py) and its tests import it under that name
(test_fp8_metal.py:58, test_mps_vs_cpu.py:217).  The implementation lives in
fp8_mi355x_native.py (HIP / gfx950); this module re-exports it unchanged.
"""
from fp8_mi355x_native import *  # noqa: F401,F403
from fp8_mi355x_native import (fp8_scaled_mm, fp8_dequantize, fp8_encode, fp8_quantize,  # noqa: F401
                               fp8_scaled_mm_auto, fp8_scaled_mm_fast, fp8_amax)
