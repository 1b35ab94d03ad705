"""Palette PNG writer of the eval / validation loops (mirror of the reference's RSSFormer-TIP2023/module/viz.py:6-23:
`VisualizeSegmm(out_dir, palette)(y_pred, filename)`; host-side utility, PIL as in the reference)."""
import os

import numpy as np

# data/loveda.py:22-30 of the reference (class order = label id), flattened as eval.py:49 does
LOVEDA_PALETTE = [255, 255, 255, 255, 0, 0, 255, 255, 0, 0, 0, 255, 159, 129, 183, 0, 255, 0, 255, 195, 128]


class VisualizeSegmm(object):
    def __init__(self, out_dir, palette):
        self.out_dir = out_dir
        self.palette = palette
        os.makedirs(self.out_dir, exist_ok=True)

    def __call__(self, y_pred, filename):
        """y_pred: 2-D or 3-D array [1 (optional), H, W] of class ids; written as an 8-bit palette image.
        (The ignore label -1 becomes 255 through the uint8 cast and, PIL storing a 7-colour palette at 4 bits, 15 in the file -
        exactly as in the reference, whose files were compared byte for byte.)"""
        from PIL import Image
        y_pred = np.asarray(y_pred).astype(np.uint8).squeeze()
        color_y = Image.fromarray(y_pred)
        color_y.putpalette(self.palette)
        color_y.save(os.path.join(self.out_dir, filename))
