"""tk.dl.od: ObjectDetector + od_gen (reference voc_validate.py:25, check_generator.py:17)."""
from ... import od_gen  # noqa: F401
from ...detector import ObjectDetector, ObjectsPrediction  # noqa: F401
from ...pb import ObjectsAnnotation, PriorBoxes  # noqa: F401
