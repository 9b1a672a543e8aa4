"""object_detector_amd: MI355X (gfx950)-native hot path of ak110/object_detector.

Python host code mirroring the pytoolkit API the reference scripts call (voc_validate.py:24-29, check_assign.py:19-27,
check_generator.py:17-18) over hand-written HIP kernels in libodhip.so (include/odhip.h).  No CPU fallback.
"""
__version__ = "0.1.0"
