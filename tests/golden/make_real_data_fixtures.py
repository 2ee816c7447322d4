"""Derives the small real-data fixtures from the reference's shipped data sets (run in the build container, where /root/reference is
mounted; the GPU box only sees the outputs).  Data only -- no reference code is read or copied.

  sfs_default_q4.npz      examples/data/shape_from_shading/default_*  sampled every 4th pixel (640x480 -> 160x120), intrinsics scaled
  small_armadillo.ply/.mrk  examples/data/small_armadillo.*            copied as they are (5 KB + 117 B)
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from thallo_amd import formats as F

D = "/root/reference/examples/data/"
S = 4
pre = D + "shape_from_shading/default"
depth = F.read_imagedump(pre + "_targetDepth.imagedump")[::S, ::S, 0]
inten = F.read_imagedump(pre + "_targetIntensity.imagedump")[::S, ::S, 0]
init = F.read_imagedump(pre + "_initialUnknown.imagedump")[::S, ::S, 0]
edge = F.read_imagedump(pre + "_maskEdgeMap.imagedump")[:, :, 0]            # rows 0..H-1 = row map, H..2H-1 = column map (SFSSolverInput.h:42-44)
H = edge.shape[0] // 2
mR, mC = edge[:H][::S, ::S], edge[H:][::S, ::S]
prm = F.read_sfs_params(pre + ".SFSSolverParameters")
scalars = np.array([prm["weightFitting"], prm["weightRegularizer"], prm["weightShading"], prm["fx"] / S, prm["fy"] / S, prm["ux"] / S, prm["uy"] / S] +
                   list(prm["lightingCoefficients"]), dtype=np.float32)
np.savez_compressed(os.path.join(HERE, "sfs_default_q4.npz"), depth=np.ascontiguousarray(depth), intensity=np.ascontiguousarray(inten),
                    initial=np.ascontiguousarray(init), edge_r=np.ascontiguousarray(mR), edge_c=np.ascontiguousarray(mC), scalars=scalars)
for n in ("small_armadillo.ply", "small_armadillo.mrk"):
    shutil.copy(D + n, os.path.join(HERE, n))
print({k: os.path.getsize(os.path.join(HERE, k)) for k in ("sfs_default_q4.npz", "small_armadillo.ply", "small_armadillo.mrk")})
