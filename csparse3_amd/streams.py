"""Independent matrices with DISTINCT patterns, one matrix per HIP stream (BASELINE configs[4], "one matrix per GPU
stream"; SURVEY.md section 8d, second variant of config 5).

Matrices that share a pattern go through ONE batched handle (csc_hip.Factorization(batch=...)): one analysis, every
launch covers the whole batch.  Matrices with different patterns cannot share launches; what they can share is the GPU:
each gets its own handle (its own analysis, HBM state and hipGraph) and its factor + solve call is issued on one of a
small pool of streams, so that the per-level launches of different matrices overlap instead of queueing behind each
other.  torch supplies the streams and the device buffers; the numeric work is the C ABI's.
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from csparse3_amd import csc_hip


class DistinctBatch:
    """Handles for a list of matrices (m, n, Ap, Ai) with different patterns; analysis on construction.

    The analyses (ordering + symbolic, host C++ inside the library) are independent and run in a pool of threads: the
    ctypes call releases the interpreter lock and cs3_analyze keeps no shared state (its error string is per thread), so N
    patterns cost N / workers analyses of wall time instead of N (analysis_s records it).  workers=1 analyses in line."""

    def __init__(self, patterns, kind=csc_hip.CS3_LU, nstreams=8, device=None, workers=None):
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.kind = kind
        if workers is None:
            workers = min(len(patterns), os.cpu_count() or 1, 32)
        t0 = time.perf_counter()
        make = lambda p: csc_hip.Factorization(p[0], p[1], p[2], p[3], kind=kind)
        if workers > 1 and len(patterns) > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                self.handles = list(pool.map(make, patterns))           # (results in the order of `patterns`)
        else:
            self.handles = [make(p) for p in patterns]
        self.analysis_s = time.perf_counter() - t0
        self.workers = workers
        self.n = [int(p[1]) for p in patterns]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, min(nstreams, len(patterns))))]

    def close(self):
        for h in self.handles:
            h.close()
        self.handles = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def factor_solve(self, values, rhs, tol=0.0):
        """values[i]: device tensor of matrix i's entries; rhs[i]: device tensor [n_i] or [n_i, k], overwritten by
        the solution.  Matrix i runs on stream i mod nstreams; returns after every stream has been waited for."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)                       # the inputs were produced on the current stream
        for i, h in enumerate(self.handles):
            s = self.streams[i % len(self.streams)]
            x = rhs[i]
            k = 1 if x.dim() == 1 else x.shape[1]
            h.factor_solve_dev(values[i].data_ptr(), x.data_ptr(), k, tol, s.cuda_stream)
        for s in self.streams:
            cur.wait_stream(s)
        return rhs

    def status(self):
        """Raises for the first matrix whose factorisation failed (after synchronising)."""
        for i, h in enumerate(self.handles):
            h.factor_status(self.streams[i % len(self.streams)].cuda_stream)
