import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ssvio_amd
from ssvio_amd import orb as sorb
from oracle import pyoracle as po
ctx = ssvio_amd.Context(0)
for px in (48, 49, 50, 51):
    img = np.zeros((200, 300), np.uint8); img[100, px] = 255
    ex = sorb.ORBextractor(ctx, nfeatures=100, nlevels=2)
    ex.DetectAndCompute(img)
    bl = ex.stage_level(0, blurred=True); rb = po.gauss7(img)
    print('impulse at x=%d' % px)
    print(' gpu row100', bl[100, 40:60].tolist())
    print(' ref row100', rb[100, 40:60].tolist())
    print(' gpu col   ', bl[94:107, px].tolist())
    print(' ref col   ', rb[94:107, px].tolist())
