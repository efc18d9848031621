"""ctypes wrapper of oracle/vq_oracle.c -- TEST INFRASTRUCTURE (see that file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvq_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.vq_oracle_assign.restype = None
    lib.vq_oracle_assign.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int] + [C.c_void_p] * 6
    return lib


def assign(x, embed):
    """x[..., 64] float32, embed[64, n_embed] float32 -> dict(ind, q_ste, sq_sum, counts, esum, best_dist)."""
    x = np.ascontiguousarray(x, np.float32)
    embed = np.ascontiguousarray(embed, np.float32)
    nvec, n_embed = x.size // 64, embed.shape[1]
    ind = np.empty(nvec, np.int64)
    q = np.empty((nvec, 64), np.float32)
    sq = C.c_double(0)
    counts = np.empty(n_embed, np.float32)
    esum = np.empty((64, n_embed), np.float32)
    best = np.empty(nvec, np.float32)
    _lib().vq_oracle_assign(x.ctypes.data, nvec, embed.ctypes.data, n_embed, ind.ctypes.data, q.ctypes.data,
                            C.addressof(sq), counts.ctypes.data, esum.ctypes.data, best.ctypes.data)
    return dict(ind=ind.reshape(x.shape[:-1]), q_ste=q.reshape(x.shape), sq_sum=sq.value, counts=counts, esum=esum,
                best_dist=best)
