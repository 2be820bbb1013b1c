"""Shared test helpers (test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
MAX_LEGAL = 136


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


_hc = None


def hostcheck():
    """Host build of alphaquoridorgnn_amd/csrc/quoridor_core.hpp (the header the HIP kernels compile)."""
    global _hc
    if _hc is None:
        src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
        hdr = os.path.join(REPO, "alphaquoridorgnn_amd", "csrc", "quoridor_core.hpp")
        so = os.path.join(HERE, "hostcheck", "libhostcheck.so")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, src])
        _hc = ctypes.CDLL(so)
    return _hc


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hc_legal(N, recs):
    recs = np.ascontiguousarray(recs, dtype=np.uint8).reshape(-1, 72)
    B = recs.shape[0]
    out = np.empty((B, MAX_LEGAL), dtype=np.int16)
    cnt = np.empty(B, dtype=np.int32)
    assert hostcheck().hc_legal_actions_batch(N, _p(recs), B, _p(out), _p(cnt)) == 0
    return out, cnt


def hc_next(N, recs, actions):
    recs = np.ascontiguousarray(recs, dtype=np.uint8).reshape(-1, 72)
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    out = np.empty_like(recs)
    assert hostcheck().hc_next_batch(N, _p(recs), _p(actions), recs.shape[0], _p(out)) == 0
    return out


def hc_status(N, recs, draw):
    recs = np.ascontiguousarray(recs, dtype=np.uint8).reshape(-1, 72)
    out = np.empty(recs.shape[0], dtype=np.uint8)
    assert hostcheck().hc_status_batch(N, _p(recs), recs.shape[0], draw, _p(out)) == 0
    return out
