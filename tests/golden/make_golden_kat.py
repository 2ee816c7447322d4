"""Decode the reference's two known-answer images into raw fixtures.

Run once in the build container (needs /root/reference and PIL); the outputs
`minimal_gold.u8` (512*512 bytes) and `minimal_graph_gold.u8` (512 bytes) are data
files held by the reference's own tests (tests/minimal/gold.png,
tests/minimal_graph/gold.png) -- decoded pixel bytes, no reference source.
"""
import os
import numpy as np
from PIL import Image

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
for name, shape in (("minimal", (512, 512)), ("minimal_graph", (1, 512))):
    im = np.asarray(Image.open(os.path.join(REF, name, "gold.png")))
    assert im.shape == shape and im.dtype == np.uint8, (im.shape, im.dtype)
    im.tofile(os.path.join(HERE, f"{name}_gold.u8"))
    print(name, im.shape, im.min(), im.max())
