"""MODE on MI355X: same import surface as the reference's models/__init__.py:1-3 -- the disparity stage (the hot path) and the
fusion stage's ModeFusion / Baseline."""
from .mode_disparity import ModeDisparity
from .mode_fusion import ModeFusion, Baseline
from .initModel import initModelPara, loadStackHourglassOnly
