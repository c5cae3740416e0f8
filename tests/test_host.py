"""Host-side plumbing that needs no GPU: the pool of page-locked output buffers behind run_numpy (device.py).  The arrays a
reference-style caller gets back are the DMA targets themselves; a buffer may be written again only when the caller can no
longer see it -- through the array it was handed or through any view derived from it."""
import gc
import importlib

import numpy as np
import pytest


def test_host_pool_never_reuses_a_buffer_the_caller_can_still_see(monkeypatch):
    torch = pytest.importorskip("torch")
    dev = importlib.import_module("distancetransform-depthcompletion_amd.device")
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self: self)  # no GPU here: pageable memory stands in
    pool = dev._HostPool(cap_bytes=1 << 20)
    h1 = pool.take((2, 3), np.float32)
    a1, p1 = h1.array, h1.tensor.data_ptr()
    a1[...] = 7
    del h1
    h2 = pool.take((2, 3), np.float32)
    assert h2.tensor.data_ptr() != p1  # a1 is alive: another buffer
    view = np.expand_dims(a1, -1)[0]  # what tools.DT_complete_batch hands out is a view, and callers slice further
    del a1
    gc.collect()
    assert pool.take((2, 3), np.float32).tensor.data_ptr() != p1 and (view == 7).all()  # the view alone keeps it
    del view
    gc.collect()
    h4 = pool.take((2, 3), np.float32)
    assert h4.tensor.data_ptr() == p1  # nobody can see it any more: reused
    assert pool.take((2, 3), np.int32).array.dtype == np.int32 and pool.take((4, 3), np.float32).array.shape == (4, 3)
    # beyond the cap, buffers are simply released with their arrays
    small = dev._HostPool(cap_bytes=16)
    p = small.take((8, 8), np.float32).tensor.data_ptr()
    gc.collect()
    assert small._cached == 0 and p
