"""Device-side plumbing: torch owns the HBM buffers and the stream, libdtfill.so does the work.

`DtFill` is the batched operator behind the reference-named functions in tools.py.  It holds one
workspace + output set per (B, H, W) so that repeated calls (the reference calls the op once per
frame in a loop, demo.py:268-290 / eval_NYU.py:138-195) allocate nothing.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_NP_DTYPES = {"depth": np.float32, "dt": np.float32, "index": np.int32, "status": np.int32}
WANT_ALL = ("depth", "dt", "index")


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "dtfill needs a HIP device (torch.cuda.is_available() is False); the operator has "
            "no CPU fallback -- the CPU restatement under oracle/ is test infrastructure only"
        )


class DtFill:
    """Batched DT + nearest-valid-depth fill on one GPU.

    Parameters mirror the literals of the reference: src_thr is the 0.1 of tools.py:8 (0.001 in
    eval_NYU.py:115), val_thr the 0.1 of tools.py:22; metric "l1_cv" is the reference's
    cv2.DIST_L1 / mask 5 / DIST_LABEL_PIXEL transform.
    """

    def __init__(self, device=None, metric="l1_cv"):
        _require_gpu()
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if metric not in _lib.METRICS:
            raise ValueError("metric must be one of %s" % sorted(_lib.METRICS))
        self.metric = _lib.METRICS[metric]
        self._shape = None
        self._ws = None
        self._out = None

    # -- buffers -------------------------------------------------------------------------------
    def _ensure(self, B, H, W):
        if self._shape == (B, H, W):
            return
        nbytes = self.lib.dtfill_workspace_bytes(B, H, W, self.metric)
        if nbytes == 0:
            _lib.check(-2)
        self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        self._ws_off = (-self._ws.data_ptr()) % 256
        self._ws_bytes = nbytes
        self._out = {
            "depth": torch.empty((B, H, W), dtype=torch.float32, device=self.device),
            "dt": torch.empty((B, H, W), dtype=torch.float32, device=self.device),
            "index": torch.empty((B, H, W), dtype=torch.int32, device=self.device),
            "status": torch.empty((B,), dtype=torch.int32, device=self.device),
        }
        self._shape = (B, H, W)

    def workspace_bytes(self, B, H, W):
        return int(self.lib.dtfill_workspace_bytes(B, H, W, self.metric))

    # -- the op --------------------------------------------------------------------------------
    def run(self, x, src_thr=0.1, val_thr=0.1, want=WANT_ALL, timed=False, path="auto", depth_rows_from=0, depth_floor=None,
            outlier_removal=False):
        """x: float32 CUDA tensor [B,H,W] (contiguous).  Returns a dict of device tensors
        (views of buffers owned by this object, overwritten by the next call) for the names in
        `want`, plus "status" (int32 [B], bit set of _lib.FRAME_*).  Asynchronous on the current
        stream unless timed.  path: "auto" (window kernel for dense frames, any-distance kernels for the
        others), "general" or "fused" (tests / benchmarks).  depth_rows_from / depth_floor: the drivers'
        post-fill steps folded into the depth stores (demo.py:292-293 rows 96:, eval_NYU.py:205
        relu(d - 0.9) + 0.9): "depth" is then [B, H - depth_rows_from, W]; l1_cv only.  outlier_removal: the loader's filter
        (data_read.py:103-128, 168-169) in front of the predicates -- the pass equals run(outlier_removal_device(x)) without
        the filtered map being written."""
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or not x.is_contiguous():
            raise ValueError("x must be a contiguous float32 CUDA tensor [B,H,W]")
        B, H, W = x.shape
        with torch.cuda.device(x.device):
            if x.device != self.device:
                raise ValueError("x lives on %s, operator on %s" % (x.device, self.device))
            self._ensure(B, H, W)
            o = self._out
            epi = depth_rows_from != 0 or depth_floor is not None
            if epi:
                if timed or not (0 <= depth_rows_from < H):
                    raise ValueError("depth epilogue: 0 <= depth_rows_from < H, and not with timed=True")
                if getattr(self, "_crop", None) is None or self._crop.shape != (B, H - depth_rows_from, W):
                    self._crop = torch.empty((B, H - depth_rows_from, W), dtype=torch.float32, device=self.device)
                o = dict(o, depth=self._crop)
            ptr = lambda name: o[name].data_ptr() if name in want else None
            stream = torch.cuda.current_stream(self.device).cuda_stream
            args = [
                x.data_ptr(), B, H, W, float(src_thr), float(val_thr), self.metric,
                ptr("depth"), ptr("dt"), ptr("index"), o["status"].data_ptr(),
                self._ws.data_ptr() + self._ws_off, self._ws_bytes, stream,
            ]
            flags = _lib.PATHS[path] | (_lib.FLAG_OUTLIER_REMOVAL if outlier_removal else 0)
            if timed:
                nk = self.lib.dtfill_num_kernels(self.metric)
                ms = (ctypes.c_float * nk)()
                _lib.check(self.lib.dtfill_batch_timed(*args, flags, ctypes.cast(ms, ctypes.c_void_p)))
                names = [self.lib.dtfill_kernel_name(self.metric, k).decode() for k in range(nk)]
                self.last_kernel_ms = dict(zip(names, [float(v) for v in ms]))
            elif epi:
                _lib.check(self.lib.dtfill_batch_epilogue(*args, flags, int(depth_rows_from), int(depth_floor is not None),
                                                          float(depth_floor or 0.0)))
            else:
                _lib.check(self.lib.dtfill_batch_flags(*args, flags))
        res = {k: o[k] for k in want}
        res["status"] = o["status"]
        return res

    def upload(self, xh):
        """Host frames [B,H,W] (any real dtype / strides) -> this operator's float32 device input buffer, asynchronously on the
        current stream: a small thread pool copies the caller's pageable array into pinned staging memory a few frames at a
        time (numpy releases the GIL inside the copy), every chunk's DMA starting as soon as its copy is done."""
        B, H, W = xh.shape
        self._ensure(B, H, W)
        if getattr(self, "_pin_shape", None) != (B, H, W):
            self._pin_in = torch.empty((B, H, W), dtype=torch.float32).pin_memory()
            self._dev_in = torch.empty((B, H, W), dtype=torch.float32, device=self.device)
            self._pin_status = torch.empty((B,), dtype=torch.int32).pin_memory()
            self._pin_shape = (B, H, W)
        nchunk = min(B, 8)
        cuts = [B * c // nchunk for c in range(nchunk + 1)]
        pin_in = self._pin_in.numpy()
        pool = _copy_pool()
        with torch.cuda.device(self.device):
            stage = [pool.submit(np.copyto, pin_in[cuts[c]:cuts[c + 1]], xh[cuts[c]:cuts[c + 1]], "same_kind") for c in range(nchunk)]
            for c in range(nchunk):
                stage[c].result()  # (one pass: gathers strided input, casts if needed)
                self._dev_in[cuts[c]:cuts[c + 1]].copy_(self._pin_in[cuts[c]:cuts[c + 1]], non_blocking=True)
        return self._dev_in

    def pass_stats(self):
        """Which kernel family owned how many pixels in the last run() of this operator (dtfill_pass_stats): dict of ints."""
        B, H, W = self._shape
        out = torch.zeros(len(_lib.STATS), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dtfill_pass_stats(self._ws.data_ptr() + self._ws_off, self._ws_bytes, B, H, W, self.metric, out.data_ptr(),
                                                  torch.cuda.current_stream(self.device).cuda_stream))
        return dict(zip(_lib.STATS, [int(v) for v in out.cpu().tolist()]))

    def run_numpy(self, x, src_thr=0.1, val_thr=0.1, want=WANT_ALL, depth_rows_from=0, depth_floor=None, outlier_removal=False):
        """numpy in / numpy out: H2D, run, D2H.  x: float32 [B,H,W].  Raises IndexError exactly
        where numpy would in depth_list[label_list-1] (tools.py:26) when depth is wanted.

        Host side of the transfer (what a reference-style caller pays on top of the pass):
          in   the caller's pageable array is copied into pinned staging memory a few frames at a time by a small thread
               pool (numpy releases the GIL inside the copy), every chunk's DMA starting as soon as its copy is done;
          out  the device-to-host DMA lands directly in the array that is returned: a page-locked buffer taken from a
               rotating pool (_HostPool).  A buffer goes back to the pool when the caller has dropped the array it was handed
               (and every view of it), so each call still returns arrays nothing else will ever write to -- the contract of
               the reference, which builds fresh arrays (tools.py:29-35) -- without the extra host copy."""
        xh = np.asarray(x)
        if xh.ndim != 3:
            raise ValueError("x must be [B,H,W]")
        B, H, W = xh.shape
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            dev_in = self.upload(xh)
            res = self.run(dev_in, src_thr, val_thr, want, depth_rows_from=depth_rows_from, depth_floor=depth_floor,
                           outlier_removal=outlier_removal)
            self._pin_status.copy_(res["status"], non_blocking=True)
            out = {}
            for k, v in res.items():
                if v.dim() != 3:
                    continue
                # contiguous in the shape that is returned (a cropped depth is [B, H - r0, W]): one DMA per output
                host = _host_pool.take(tuple(v.shape), _NP_DTYPES[k])
                host.tensor.copy_(v, non_blocking=True)
                out[k] = host.array
            stream.synchronize()
            out["status"] = self._pin_status.numpy().copy()
        if "depth" in want:
            bad = np.nonzero(out["status"] & _lib.FRAME_INDEX_ERROR)[0]
            if bad.size:
                raise IndexError(
                    "frame %d: index out of bounds in depth_list[label_list-1] "
                    "(value list shorter than a label, or empty with label 0)" % int(bad[0])
                )
        return out


class _HostBuf:
    __slots__ = ("tensor", "array", "alive")


class _HostPool:
    """Page-locked host buffers handed out as numpy arrays.  A buffer is free again once the array made from it (and every
    view: numpy views keep their base alive) has been garbage collected; until then nobody writes to it.  At most
    `cap_bytes` stay cached; beyond that a request gets a buffer that is simply released with its array."""

    def __init__(self, cap_bytes=2 << 30):
        import threading

        self._free = {}
        self._lock = threading.Lock()
        self._cached = 0
        self._cap = cap_bytes

    def take(self, shape, dtype):
        import weakref

        key = (tuple(shape), np.dtype(dtype).str)
        with self._lock:
            lst = self._free.get(key)
            t = lst.pop() if lst else None
        if t is None:
            t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype).pin_memory()
        else:
            with self._lock:
                self._cached -= t.numel() * t.element_size()
        # a fresh ndarray object over the pinned memory.  numpy collapses the base of every view (slices, expand_dims, ...) to
        # this object, so it is alive exactly as long as anything of the caller's still looks at the buffer; when it dies the
        # tensor goes back to the pool
        a = t.numpy()
        h = _HostBuf()
        h.tensor, h.array = t, a
        weakref.finalize(a, self._give_back, key, t)
        return h

    def _give_back(self, key, t):
        nbytes = t.numel() * t.element_size()
        with self._lock:
            if self._cached + nbytes <= self._cap:
                self._free.setdefault(key, []).append(t)
                self._cached += nbytes


_host_pool = _HostPool()
_pool = None


def _copy_pool():
    """Threads for host-side staging copies (numpy's copy loops release the GIL)."""
    global _pool
    if _pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        _pool = ThreadPoolExecutor(max(1, min(8, n)), thread_name_prefix="dtfill-copy")
    return _pool


def host_link_bandwidth(nbytes=64 << 20, device=None, repeats=5):
    """Measured pinned host <-> device copy rates of this box in GB/s: what bounds a numpy-in / numpy-out caller."""
    _require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rates = {}
    for name, (dst, src) in (("h2d", (d, h)), ("d2h", (h, d))):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(repeats):
            dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize(dev)
        rates[name] = nbytes * repeats / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return rates


def outlier_removal_device(x):
    """x: contiguous float32 CUDA tensor [B,H,W] -> new tensor, data_read.py:103-128 on the device."""
    _require_gpu()
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or not x.is_contiguous():
        raise ValueError("x must be a contiguous float32 CUDA tensor [B,H,W]")
    out = torch.empty_like(x)
    B, H, W = x.shape
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dtfill_outlier_removal(x.data_ptr(), B, H, W, out.data_ptr(),
                                                      torch.cuda.current_stream(x.device).cuda_stream))
    return out


def generate_multi_channel_device(data, mask, table_size=7, scale_num=4):
    """net.py:83-122 on the device.  data, mask: contiguous float32 CUDA tensors [B,H,W].
    Returns (lidar_1, lidar_2, lidar_3, lidar_4) with None beyond scale_num, like the reference."""
    _require_gpu()
    for t in (data, mask):
        if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 3 or not t.is_contiguous():
            raise ValueError("data and mask must be contiguous float32 CUDA tensors [B,H,W]")
    if data.shape != mask.shape:
        raise ValueError("data and mask shapes differ")
    B, H, W = data.shape
    outs = [torch.empty_like(data) for _ in range(scale_num - 1)]
    ptrs = [o.data_ptr() for o in outs] + [None] * (4 - scale_num)
    with torch.cuda.device(data.device):
        _lib.check(_lib.load().dtfill_generate_multi_channel(
            data.data_ptr(), mask.data_ptr(), B, H, W, table_size, scale_num, ptrs[0], ptrs[1], ptrs[2],
            torch.cuda.current_stream(data.device).cuda_stream))
    return tuple([data] + outs + [None] * (4 - scale_num))


def _check_frames(t, what):
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 3 or not t.is_contiguous():
        raise ValueError("%s must be a contiguous float32 CUDA tensor [B,H,W]" % what)


def crop_floor_device(x, rows=None, cols=None, floor=None):
    """out = f(x[:, rows[0]:rows[1], cols[0]:cols[1]]); f = relu(d - floor) + floor when floor is given
    (demo.py:292-293, eval_NYU.py:202-205).  x: contiguous float32 CUDA tensor [B,H,W] -> new tensor."""
    _require_gpu()
    _check_frames(x, "x")
    B, H, W = x.shape
    r0, r1 = (0, H) if rows is None else rows
    c0, c1 = (0, W) if cols is None else cols
    if not (0 <= r0 < r1 <= H and 0 <= c0 < c1 <= W):
        raise ValueError("empty or out-of-frame crop rows=%s cols=%s of a %dx%d frame" % (rows, cols, H, W))
    out = torch.empty((B, r1 - r0, c1 - c0), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dtfill_crop_floor(x.data_ptr(), B, H, W, r0, r1, c0, c1, int(floor is not None),
                                                 float(floor or 0.0), out.data_ptr(),
                                                 torch.cuda.current_stream(x.device).cuda_stream))
    return out


def png16_device(x, pad_top=96, floor=0.9, lo=0.0, hi=100.0, scale=256.0):
    """test.py:133-148 on the device: x float32 CUDA [B,H,W] -> uint16 CUDA [B, pad_top+H, W]."""
    _require_gpu()
    _check_frames(x, "x")
    B, H, W = x.shape
    out = torch.empty((B, H + pad_top, W), dtype=torch.uint16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dtfill_png16(x.data_ptr(), B, H, W, pad_top, int(floor is not None), float(floor or 0.0),
                                            lo, hi, scale, out.data_ptr(),
                                            torch.cuda.current_stream(x.device).cuda_stream))
    return out


def metrics_device(output, target, kind="kitti"):
    """evaluation.py:82-123 (kind "kitti") / :196-239 ("nyu"), one row per frame.  output, target: contiguous
    float32 CUDA tensors [B, ...] of equal shape -> float64 CUDA tensor [B, 9], columns _lib.METRICS_COLUMNS."""
    _require_gpu()
    kinds = {"kitti": _lib.METRICS_KITTI, "nyu": _lib.METRICS_NYU}
    if kind not in kinds:
        raise ValueError("kind must be 'kitti' or 'nyu'")
    for t in (output, target):
        if t.dtype != torch.float32 or not t.is_cuda or t.dim() < 2 or not t.is_contiguous():
            raise ValueError("output and target must be contiguous float32 CUDA tensors [B, ...]")
    if output.shape != target.shape:
        raise ValueError("output and target shapes differ")
    B = output.shape[0]
    n = output[0].numel()
    L = _lib.load()
    nbytes = L.dtfill_metrics_workspace_bytes(B)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=output.device)
    out = torch.empty((B, len(_lib.METRICS_COLUMNS)), dtype=torch.float64, device=output.device)
    with torch.cuda.device(output.device):
        _lib.check(L.dtfill_metrics(output.data_ptr(), target.data_ptr(), B, n, kinds[kind], out.data_ptr(),
                                    ws.data_ptr(), nbytes, torch.cuda.current_stream(output.device).cuda_stream))
    return out


_default_ops = {}


def default_op(metric="l1_cv"):
    """Process-wide operator on the current CUDA device (what the reference-named functions use)."""
    _require_gpu()
    key = (torch.cuda.current_device(), metric)
    if key not in _default_ops:
        _default_ops[key] = DtFill(metric=metric)
    return _default_ops[key]


def fill(x, src_thr=0.1, val_thr=0.1, metric="l1_cv", want=WANT_ALL):
    """Batched numpy API: x [B,H,W] float32 -> dict(depth, dt, index, status)."""
    return default_op(metric).run_numpy(x, src_thr, val_thr, want)
