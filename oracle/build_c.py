"""Compile oracle/daf_ref.c (gcc) into oracle/_build/libdaf_ref.so and bind it with ctypes. Test
infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "daf_ref.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libdaf_ref.so")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(OUT_DIR, exist_ok=True)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def daf_forward(feat, spatial_shape, scale_start_index, loc, weights):
    """numpy/torch-CPU arrays in the operator's layouts -> f32 [bs, A, C]."""
    lib = ctypes.CDLL(build())
    f = np.ascontiguousarray(np.asarray(feat, np.float32))
    ss = np.ascontiguousarray(np.asarray(spatial_shape, np.int32))
    st = np.ascontiguousarray(np.asarray(scale_start_index, np.int32))
    lc = np.ascontiguousarray(np.asarray(loc, np.float32))
    w = np.ascontiguousarray(np.asarray(weights, np.float32))
    bs, num_feat, c = f.shape
    cams, lvls = ss.shape[:2]
    a, p = lc.shape[1:3]
    g = w.shape[5]
    out = np.zeros((bs, a, c), np.float32)
    P = ctypes.c_void_p
    lib.daf_ref_forward.argtypes = [P] * 6 + [ctypes.c_int] * 8
    lib.daf_ref_forward.restype = None
    lib.daf_ref_forward(out.ctypes.data, f.ctypes.data, ss.ctypes.data, st.ctypes.data, lc.ctypes.data, w.ctypes.data,
                        bs, cams, num_feat, c, lvls, a, p, g)
    return out
