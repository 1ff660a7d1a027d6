"""Deep360 data access for the disparity and fusion stages (reference: dataloader/{list_file,preprocess,deep360_loader}.py).
Host-side IO around the hot path; needs PIL and numpy only (the reference needs cv2 and torchvision)."""
from . import list_file, preprocess
from .deep360_loader import Deep360DatasetDisparity, Deep360DatasetFusion
