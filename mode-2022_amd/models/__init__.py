"""MODE disparity stage on MI355X: same import surface as the reference's models/__init__.py:1-3
(ModeFusion / Baseline are outside the hot path and not provided)."""
from .mode_disparity import ModeDisparity
from .mode_fusion import ModeFusion, Baseline
from .initModel import initModelPara, loadStackHourglassOnly
