import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mws as O
from raygun_amd import synth
from raygun_amd.segment import compute_mws_segmentation
NB8 = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0], [0, 2, 0], [0, 0, 2], [0, 1, 1], [0, 1, -1]], dtype=np.int32)
for shape in [(16, 16, 16), (10, 10, 10), (24, 24, 24)]:
    w = synth.uniform((8,) + shape, 11, 0, 1)
    ref = O.compute_mws_segmentation(w, NB8, 3, (1, 1, 1), None)
    got = compute_mws_segmentation(w, NB8, 3, (1, 1, 1), mask=None)
    print(os.environ.get("RG_MWS_EXP"), shape, "diff", int((got != ref).sum()), "segments", int(ref.max()), int(got.max()), flush=True)
