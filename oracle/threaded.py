"""The CPU oracle on every host core: the transform is independent per channel, so a scene is cut into
channel blocks and each block runs the scalar C oracle on its own thread (ctypes releases the GIL).
TEST INFRASTRUCTURE (checker side only), used by the full-size parity tests and by bench.py's
cpu_baseline leg."""
import os
import threading

import numpy as np


def host_threads(D, cap=16):
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    return max(1, min(ncpu, D, cap))       # every thread re-walks the plan's lists: narrower blocks stop paying


def _blocks(D, nthr):
    cuts = [round(i * D / nthr) for i in range(nthr + 1)]
    return [(cuts[i], cuts[i + 1]) for i in range(nthr) if cuts[i + 1] > cuts[i]]


def forward(orc, C64, param, nthreads=None):
    """-> (T float64 (N, D), w float64 (N, 1)) == orc.raht_fwd(C64, param), threaded over channel blocks."""
    N, D = C64.shape
    nthr = host_threads(D) if nthreads is None else nthreads
    T = np.empty((N, D), dtype=np.float64)
    wout = [None]

    def work(lo, hi, first):
        Tb, w = orc.raht_fwd(np.ascontiguousarray(C64[:, lo:hi]), param)
        T[:, lo:hi] = Tb
        if first:
            wout[0] = w
    th = [threading.Thread(target=work, args=(lo, hi, i == 0)) for i, (lo, hi) in enumerate(_blocks(D, nthr))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return T, wout[0]


def inverse(orc, T64, param, nthreads=None):
    N, D = T64.shape
    nthr = host_threads(D) if nthreads is None else nthreads
    Cr = np.empty((N, D), dtype=np.float64)

    def work(lo, hi):
        Cr[:, lo:hi] = orc.raht_inv(np.ascontiguousarray(T64[:, lo:hi]), param)
    th = [threading.Thread(target=work, args=b) for b in _blocks(D, nthr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return Cr


def split(C64, nthreads=None):
    """channel blocks of C64 (contiguous copies), one per thread"""
    nthr = host_threads(C64.shape[1]) if nthreads is None else nthreads
    return [np.ascontiguousarray(C64[:, lo:hi]) for lo, hi in _blocks(C64.shape[1], nthr)]


def fwd_inv_blocks(orc, blocks, param):
    """One forward + inverse pass over pre-split channel blocks, one thread each (what bench.py times as the all-cores
    CPU baseline: no assembling of the result inside the timed region). -> (list of T blocks, list of C blocks)"""
    Ts, Rs = [None] * len(blocks), [None] * len(blocks)

    def work(i):
        Ts[i], _ = orc.raht_fwd(blocks[i], param)
        Rs[i] = orc.raht_inv(Ts[i], param)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(blocks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return Ts, Rs
