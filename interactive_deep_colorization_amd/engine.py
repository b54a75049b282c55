"""HipColorizer -- thin Python owner of one ``idc_handle`` (one GPU, one stream).

Stands where ``SIGGRAPHGenerator`` (``models/pytorch/model.py:5-175``) stands in
the reference: ``forward(L_mc, ab, mask, maskcent)`` returns the raw ab map the
reference returns at ``data/colorize_image.py:263``.  All arithmetic happens in
the HIP library behind the C ABI (``include/ideepcolor.h``); this file only
marshals numpy arrays and weight dictionaries.
"""
import ctypes
import threading

import numpy as np

from . import _native as N

_PREC = {"fp32": N.IDC_FP32, "f32": N.IDC_FP32, "float32": N.IDC_FP32, 0: N.IDC_FP32,
         "bf16": N.IDC_BF16, "bfloat16": N.IDC_BF16, 1: N.IDC_BF16,
         # operand-split precisions (round 6): fp32 values carried as 2 / 3 bf16 parts, 3 / 6 bf16 MFMA products per fp32 product
         "bf16x3": N.IDC_BF16X3, 2: N.IDC_BF16X3, "bf16x6": N.IDC_BF16X6, 3: N.IDC_BF16X6,
         "fp16x3": N.IDC_FP16X3, 4: N.IDC_FP16X3, "fp16": N.IDC_FP16, 5: N.IDC_FP16}


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32c(a, shape=None):
    """float32, C-contiguous view/copy (bool masks and f64 GUI arrays accepted)."""
    out = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    if shape is not None and tuple(out.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), out.shape))
    return out


def state_dict_to_numpy(sd):
    """Accept a torch ``state_dict`` (tensors) or a dict of arrays; return {name: f32 ndarray}."""
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[k] = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
    return out


def _tensor_descs(sd):
    sd = state_dict_to_numpy(sd)
    names = sorted(sd)
    arr = (N.TensorDesc * len(names))()
    keep = []
    for i, k in enumerate(names):
        a = sd[k]
        if a.ndim > 4:
            raise ValueError("tensor %s has %d dims" % (k, a.ndim))
        kb = k.encode()
        keep.append((kb, a))
        arr[i].name = kb
        arr[i].data = _fptr(a)
        arr[i].ndim = a.ndim
        for d in range(a.ndim):
            arr[i].dims[d] = a.shape[d]
    return arr, len(names), keep


TILE_POLICIES = {"auto": 0, "small": 1, "large": 2}


def set_tile_policy(policy="auto"):
    """Process-wide tile-shape policy of the conv kernels (speed only; same results): "auto",
    "small" (conv_igemm only) or "large" (conv_igemm_v2 wherever it applies, bf16)."""
    N.check(N.load().idc_set_tile_policy(TILE_POLICIES[policy] if isinstance(policy, str) else int(policy)))


def _flags(dist=False, global_hints=False, dist313=False, throughput_blob=False):
    return ((N.IDC_FLAG_DIST_HEAD if dist else 0) | (N.IDC_FLAG_GLOBAL_HINTS if global_hints else 0) |
            (N.IDC_FLAG_DIST313 if dist313 else 0) | (N.IDC_FLAG_THROUGHPUT_BLOB if throughput_blob else 0))


SPLITK_POLICIES = {"auto": 0, "never": 1, "always": 2}


def set_option(name, value):
    """Process-wide switches (``idc_set_option``; speed / kernel choice only -- every setting computes the same function).  Names and values are
    documented in ``include/ideepcolor.h``: 'fuse_conv1', 'click', 'winograd', 'mfma16', 'v2p', 'ds_mfma16', 'kwave', 'kwave_chain', and since
    round 6 'split_ds_fuse', 'conv1_1_split', 'conv1_2_split', 'spin_sync', 'pcie_kernel' (former environment switches) and the test hook 'kw_force_abort'.  'mfma16' = 0 / 'ds_mfma16' = 0 select the
    32x32x16-MFMA partner kernels, which exist only in a ``make EXTRA=-DIDC_AB_PARTNERS`` build: the default library raises IdcError (UNSUPPORTED)."""
    N.check(N.load().idc_set_option(name.encode(), int(value)))


def set_splitk_policy(policy="auto"):
    """Process-wide split-K policy of the small-tile kernels (speed only): "auto", "never" or "always"."""
    N.check(N.load().idc_set_splitk_policy(SPLITK_POLICIES[policy] if isinstance(policy, str) else int(policy)))


def pack_weights(sd, precision="bf16", dist=False, global_hints=False, dist313=False, throughput_blob=False):
    """Host-only: reference ``state_dict`` -> packed device-ready blob (uint8 ndarray).
    Needs no GPU (used by rank 0 before the RCCL broadcast).  ``throughput_blob``: without the Winograd images of the
    batch-1 / fp32 kernels (IDC_FLAG_THROUGHPUT_BLOB: fp32 136 MB instead of 384 MB; a bf16 blob is 68 MB either way) -- must match the handle's."""
    lib = N.load()
    prec = _PREC[precision]
    flags = _flags(dist, global_hints, dist313, throughput_blob)
    nbytes = lib.idc_weights_blob_bytes(prec, flags)
    blob = np.zeros(nbytes, dtype=np.uint8)
    arr, n, keep = _tensor_descs(sd)
    N.check(lib.idc_pack_weights(prec, flags, arr, n, blob.ctypes.data_as(ctypes.c_void_p), nbytes))
    del keep
    return blob


class _PinnedBuffer(object):
    """Owner of one idc_alloc_host allocation.  numpy arrays made from it (``np.asarray`` through the array interface)
    hold it as their base, so the pinned memory is returned to the driver only when the last view is collected."""

    def __init__(self, lib, nbytes):
        self._lib = lib
        self.nbytes = max(int(nbytes), 1)
        self.ptr = lib.idc_alloc_host(self.nbytes)
        if not self.ptr:
            raise MemoryError("idc_alloc_host(%d) failed" % self.nbytes)

    @property
    def __array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (int(self.ptr), False), "version": 3}

    def __del__(self):
        try:
            if self.ptr:
                self._lib.idc_free_host(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


class _PinnedPool(object):
    """Recycled pinned result buffers of the blocking calls.  ``take`` returns a fresh numpy array over pinned memory (the
    library then copies device -> caller in place, no staging memcpy); when the last view of it is collected the memory goes
    back on the free list instead of to the driver (hipHostMalloc / hipHostFree cost more than the copy they save).  An array
    is never handed out while another array still views the same memory, so results keep the value semantics of ``np.empty``
    outputs.  Bounded on both sides: buffers above ``MAX_ONE`` or beyond ``MAX_TOTAL`` retained (free-list) bytes are plain
    numpy / freed, and at most ``MAX_LIVE`` bytes of pinned memory are in callers' hands at any time -- a caller that keeps
    many results alive gets pageable ``np.empty`` arrays beyond that (the library stages those through its own pinned
    buffer), so kept outputs cannot pin unbounded host RAM.  Counters are guarded by a lock (results may be
    dropped on any thread)."""
    MAX_ONE = 32 << 20
    MAX_TOTAL = 256 << 20
    MAX_LIVE = 512 << 20

    def __init__(self, lib):
        self.lib = lib
        self.free = {}                      # nbytes -> [ptr]
        self.retained = 0                   # bytes on the free lists
        self.live = 0                       # pinned bytes currently viewed by arrays handed out
        # re-entrant: give() runs from __del__, and a cyclic-GC pass triggered by an allocation INSIDE a locked region can finalise
        # another pooled buffer on the same thread (a plain Lock would deadlock there); the counters tolerate the nesting
        self._lock = threading.RLock()

    def take(self, shape, dtype):
        shape = tuple(int(x) for x in shape)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if nbytes == 0 or nbytes > self.MAX_ONE:
            return np.empty(shape, dtype)
        with self._lock:
            if self.live + nbytes > self.MAX_LIVE:
                return np.empty(shape, dtype)
            lst = self.free.get(nbytes)
            ptr = lst.pop() if lst else None
            if ptr is not None:
                self.retained -= nbytes
            self.live += nbytes
        try:
            owner = _PooledPinned(self, nbytes, ptr)
        except MemoryError:
            with self._lock:
                self.live -= nbytes
            return np.empty(shape, dtype)
        return np.asarray(owner)[:nbytes].view(dtype).reshape(shape)

    def give(self, ptr, nbytes):
        with self._lock:
            self.live -= nbytes
            keep = self.retained + nbytes <= self.MAX_TOTAL
            if keep:
                self.free.setdefault(nbytes, []).append(ptr)
                self.retained += nbytes
        if not keep:
            self.lib.idc_free_host(ctypes.c_void_p(ptr))

    def drain(self):
        with self._lock:
            ptrs = [p for lst in self.free.values() for p in lst]
            self.free = {}
            self.retained = 0
        for p in ptrs:
            self.lib.idc_free_host(ctypes.c_void_p(p))


class _PooledPinned(_PinnedBuffer):
    def __init__(self, pool, nbytes, ptr=None):
        self._pool = pool
        if ptr:
            self._lib, self.nbytes, self.ptr = pool.lib, nbytes, ptr
        else:
            _PinnedBuffer.__init__(self, pool.lib, nbytes)

    def __del__(self):
        try:
            if self.ptr:
                self._pool.give(self.ptr, self.nbytes)
                self.ptr = None
        except Exception:
            pass


_POOLS = {}


def _result_pool(lib):
    pool = _POOLS.get(id(lib))
    if pool is None:
        pool = _POOLS[id(lib)] = _PinnedPool(lib)
    return pool


def trim_pinned_pool():
    """Return the recycled pinned result buffers nobody views any more to the driver (at most ``_PinnedPool.MAX_TOTAL`` bytes
    are ever retained; buffers still viewed by a live array are untouched)."""
    for pool in _POOLS.values():
        pool.drain()


class HipColorizer(object):
    def __init__(self, H=256, W=None, max_batch=1, precision="bf16", device=0, dist=False, global_hints=False, dist313=False,
                 throughput_blob=False):
        self.lib = N.load()
        self.H, self.W = int(H), int(H if W is None else W)
        self.max_batch = int(max_batch)
        self.precision = precision
        self._prec = _PREC[precision]
        self.dist = bool(dist)
        self.device = int(device)
        self.global_hints = bool(global_hints)
        self.dist313 = bool(dist313)
        self.throughput_blob = bool(throughput_blob)
        self._flags = _flags(dist, global_hints, dist313, throughput_blob)
        self._h = ctypes.c_void_p()
        N.check(self.lib.idc_create(self.device, self.H, self.W, self.max_batch, self._prec, self._flags,
                                    ctypes.byref(self._h)))
        self._blob_keepalive = None
        self._pool = _result_pool(self.lib)
        self.forward_serial = 0             # bumped by every call that replaces the handle's resident results (api.py's lazy output attributes)
        self.l_serial = 0                   # bumped by every call that may write image slot 0's L plane (set_image_l and every forward): api.py's resident L
        self.before_overwrite = None        # callable run ONCE right before the next such call: whoever still wants the resident results fetches them

    # ---- lifetime -------------------------------------------------------------------------
    def close(self):
        """Destroy the handle (idc_destroy drains its streams first).  Pinned buffers handed out by ``pinned_empty``
        are NOT freed here: each is released when the last numpy view of it is garbage-collected, so an array that
        outlives the engine stays valid memory."""
        if getattr(self, "_h", None) is not None and self._h:
            cb, self.before_overwrite = getattr(self, "before_overwrite", None), None
            if cb is not None:                        # somebody still wants the resident results (api.py's lazy output attributes): last chance
                try:
                    cb()
                except Exception:
                    pass
            for slot in (0, 1):                       # batches still in flight own caller buffers: finish them first
                try:
                    self.lib.idc_wait(self._h, slot)
                except Exception:
                    pass
            self.lib.idc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, status):
        return N.check(status, self._h)

    # ---- weights --------------------------------------------------------------------------
    def load_state_dict(self, sd):
        arr, n, keep = _tensor_descs(sd)
        self._chk(self.lib.idc_load_weights(self._h, arr, n))
        del keep

    def blob_bytes(self):
        return int(self.lib.idc_weights_blob_bytes(self._prec, self._flags))

    def set_weights_blob(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(self.lib.idc_set_weights_host(self._h, blob.ctypes.data_as(ctypes.c_void_p), blob.size))

    def set_weights_device(self, dev_ptr, nbytes, copy=False, keepalive=None):
        """Adopt (or copy) a packed blob already in device memory, e.g. the torch uint8 tensor that
        received the RCCL broadcast; ``keepalive`` is held so the memory outlives the handle."""
        self._chk(self.lib.idc_set_weights_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(nbytes), 1 if copy else 0))
        self._blob_keepalive = None if copy else keepalive

    def set_io_scales(self, l_div=100., ab_div=110., mask_mul=1., out_mul=110.):
        self._chk(self.lib.idc_set_io_scales(self._h, l_div, ab_div, mask_mul, out_mul))

    # ---- global hints (models/global_model/deploy_nodist.prototxt inputs) ------------------
    def set_global_hints(self, glob_ab_313_mask, s_avg_mask=None):
        """(n,314) histogram+flag rows and optional (n,2) saturation+flag rows; in effect until cleared."""
        g = _f32c(np.atleast_2d(glob_ab_313_mask))
        if g.shape[1] != 314:
            raise ValueError("glob_ab_313_mask must be (n, 314), got %s" % (g.shape,))
        sat = None if s_avg_mask is None else _f32c(np.atleast_2d(s_avg_mask), (g.shape[0], 2))
        self._chk(self.lib.idc_set_global_hints(self._h, g.shape[0], _fptr(g), _fptr(sat) if sat is not None else None))

    def clear_global_hints(self):
        self._chk(self.lib.idc_clear_global_hints(self._h))

    # ---- forward --------------------------------------------------------------------------
    def _results_will_be_replaced(self):
        cb, self.before_overwrite = self.before_overwrite, None
        if cb is not None:
            cb()
        self.forward_serial += 1
        self.l_serial += 1

    def _prep(self, L_mc, ab, mask):
        L_mc = np.asarray(L_mc)
        if L_mc.ndim == 3:                      # reference call shape: (1,X,X),(2,X,X),(1,X,X)
            L_mc, ab, mask = L_mc[None], np.asarray(ab)[None], np.asarray(mask)[None]
        n = L_mc.shape[0]
        L = self._f32in(L_mc, (n, 1, self.H, self.W))
        A = self._f32in(ab, (n, 2, self.H, self.W))
        M = self._f32in(mask, (n, 1, self.H, self.W))
        return n, L, A, M

    def _f32in(self, a, shape):
        """float32 C-contiguous view of an input.  An array that needs converting anyway (the GUI's float64 hint planes) is
        converted straight into a pooled pinned buffer, which the library then uploads in place."""
        a = np.asarray(a)
        if a.dtype == np.float32 and a.flags.c_contiguous:
            return _f32c(a, shape)
        if tuple(a.shape) != tuple(shape):
            return _f32c(a, shape)              # raises the shape error
        dst = self._pool.take(shape, np.float32)
        np.copyto(dst, a, casting="unsafe")
        return dst

    def forward(self, L_mc, ab, mask, maskcent=0.0):
        """(N,1,H,W),(N,2,H,W),(N,1,H,W) -> (N,2,H,W) float32 ab.  3-D inputs = one image."""
        n, L, A, M = self._prep(L_mc, ab, mask)
        out = self._pool.take((n, 2, self.H, self.W), np.float32)
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward(self._h, n, _fptr(L), _fptr(A), _fptr(M), float(maskcent), _fptr(out)))
        return out

    def forward_dist(self, L_mc, ab, mask, maskcent=0.0, want_dist=True):
        """Also returns the 529-bin distribution at quarter resolution (N,529,H/4,W/4); ``want_dist=False`` leaves it
        on the device (``dist_at`` / ``suggest_colors`` / ``get_dist``) and returns None in its place."""
        n, L, A, M = self._prep(L_mc, ab, mask)
        out = np.empty((n, 2, self.H, self.W), np.float32)
        dq = np.empty((n, 529, self.H // 4, self.W // 4), np.float32) if want_dist else None
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_dist(self._h, n, _fptr(L), _fptr(A), _fptr(M), float(maskcent),
                                            _fptr(out), _fptr(dq) if want_dist else None))
        return out, dq

    # ---- click session: resident L plane + device-side hint rasterisation ------------
    def set_image_l(self, L_mc, img=0):
        """Upload L-50 of one image slot ((H,W) or (1,H,W)); stays resident for ``forward_resident``."""
        L = _f32c(np.asarray(L_mc).reshape(self.H, self.W))
        self.l_serial += 1
        self._chk(self.lib.idc_set_image_l(self._h, int(img), _fptr(L)))

    def set_hints(self, hints, mode="ab", img=0, mask_value=1.0):
        """Rasterise a hint list on the device.  hints: rows (y0, x0, y1, x1, c0, c1[, c2]) -- inclusive rectangle and
        its (a, b) [mode 'ab', the notebook's put_point] or uint8 (r, g, b) [mode 'rgb', UIControl.get_input +
        rgb2lab]; later rows paint over earlier ones."""
        rows = [tuple(r) for r in hints]
        arr = (N.Hint * max(len(rows), 1))()
        for i, r in enumerate(rows):
            arr[i].y0, arr[i].x0, arr[i].y1, arr[i].x1 = int(r[0]), int(r[1]), int(r[2]), int(r[3])
            arr[i].c0, arr[i].c1, arr[i].c2 = float(r[4]), float(r[5]), float(r[6]) if len(r) > 6 else 0.0
        self._chk(self.lib.idc_set_hints(self._h, int(img), len(rows), arr, {"ab": N.IDC_HINT_AB, "rgb": N.IDC_HINT_RGB}[mode],
                                         float(mask_value)))

    def hint_planes(self, img=0):
        """(ab (2,H,W), mask (1,H,W)) float32 copies of the resident hint planes of one slot."""
        ab = np.empty((2, self.H, self.W), np.float32)
        mask = np.empty((1, self.H, self.W), np.float32)
        self._chk(self.lib.idc_get_hint_planes(self._h, int(img), _fptr(ab), _fptr(mask)))
        return ab, mask

    def forward_resident(self, n=1, maskcent=0.0, l_cent=50.0, want_ab=True, want_rgb=True, want_lab=True):
        """Forward of slots 0..n-1 from the resident planes -> (out_ab | None, rgb | None, lab_q | None)."""
        out = np.empty((n, 2, self.H, self.W), np.float32) if want_ab else None
        rgb = np.empty((n, self.H, self.W, 3), np.uint8) if want_rgb else None
        labq = np.empty((n, 3, self.H, self.W), np.float64) if (want_rgb and want_lab) else None
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_resident(self._h, int(n), float(maskcent), float(l_cent), _fptr(out) if want_ab else None,
                                                rgb.ctypes.data_as(ctypes.c_void_p) if want_rgb else None,
                                                labq.ctypes.data_as(ctypes.c_void_p) if labq is not None else None))
        return out, rgb, labq

    # ---- colour suggestions on the resident distribution ---------------------------
    def dist_bins(self):
        return int(self.lib.idc_dist_bins(self._h))

    def keep_dist(self, on=True):
        self._chk(self.lib.idc_keep_dist(self._h, 1 if on else 0))

    def dist_at(self, y, x, img=0):
        """The predicted distribution (529 or 313 probabilities, float32) of pixel (y, x) from the last forward."""
        pdf = np.empty(self.dist_bins(), np.float32)
        self._chk(self.lib.idc_dist_at(self._h, int(img), int(y), int(x), _fptr(pdf)))
        return pdf

    def get_dist(self, n=1):
        """The whole resident distribution: (n,529,H/4,W/4) or (n,313,H,W) float32."""
        shape = (n, 313, self.H, self.W) if self.dist_bins() == 313 else (n, 529, self.H // 4, self.W // 4)
        d = np.empty(shape, np.float32)
        self._chk(self.lib.idc_get_dist(self._h, int(n), _fptr(d)))
        return d

    def suggest_colors(self, y, x, centres, K=5, N_draws=25000, seed=0, img=0, want_counts=False):
        """``get_ab_reccs`` on the device: (centres (K,2) f64, conf (K,) f64[, counts (B,) uint32]) ordered by occupancy."""
        B = self.dist_bins()
        c = _f32c(centres, (B, 2))
        oc, of = np.empty((K, 2), np.float64), np.empty(K, np.float64)
        cnt = np.empty(B, np.uint32) if want_counts else None
        vp = ctypes.c_void_p
        self._chk(self.lib.idc_suggest_colors(self._h, int(img), int(y), int(x), int(K), int(N_draws), int(seed) & 0xFFFFFFFF, _fptr(c),
                                              oc.ctypes.data_as(vp), of.ctypes.data_as(vp),
                                              cnt.ctypes.data_as(vp) if want_counts else None))
        return (oc, of, cnt) if want_counts else (oc, of)

    def global_histogram(self, rgb, centres, want_sat=True):
        """Global statistics of reference image(s) (the reference's global_stats.prototxt): rgb (n,H,W,3) or (H,W,3)
        uint8, centres (313,2) -> (hist (n,313) float32 summing to 1, s_avg (n,) mean HSV saturation or None)."""
        rgb = np.ascontiguousarray(np.asarray(rgb), dtype=np.uint8)
        if rgb.ndim == 3:
            rgb = rgb[None]
        n = rgb.shape[0]
        if rgb.shape != (n, self.H, self.W, 3):
            raise ValueError("rgb must be (n,%d,%d,3), got %s" % (self.H, self.W, rgb.shape))
        c = _f32c(centres, (313, 2))
        hist = np.empty((n, 313), np.float32)
        sat = np.empty(n, np.float32) if want_sat else None
        self._chk(self.lib.idc_global_histogram(self._h, n, rgb.ctypes.data_as(ctypes.c_void_p), _fptr(c), _fptr(hist),
                                                _fptr(sat) if want_sat else None))
        return hist, sat

    def lab2rgb(self, L, ab, want_lab=True):
        """Device colour step: L (n,1,H,W) in [0,100], ab (n,2,H,W) -> (rgb (n,H,W,3) uint8, lab_q (n,3,H,W) f64 or None)
        = ``lab2rgb_transpose`` + the rgb->Lab refresh of the reference (colorize_image.py:20-36,196-198)."""
        L = np.asarray(L)
        if L.ndim == 3:
            L, ab = L[None], np.asarray(ab)[None]
        n = L.shape[0]
        Lc, Ac = _f32c(L, (n, 1, self.H, self.W)), _f32c(ab, (n, 2, self.H, self.W))
        rgb = np.empty((n, self.H, self.W, 3), np.uint8)
        labq = np.empty((n, 3, self.H, self.W), np.float64) if want_lab else None
        self._results_will_be_replaced()
        self._chk(self.lib.idc_lab2rgb(self._h, n, _fptr(Lc), _fptr(Ac), rgb.ctypes.data_as(ctypes.c_void_p),
                                       labq.ctypes.data_as(ctypes.c_void_p) if want_lab else None))
        return rgb, labq

    def forward_rgb(self, L_mc, ab, mask, maskcent=0.0, l_cent=50.0, want_lab=True):
        """forward + the colour step on the device: (out_ab (n,2,H,W) f32, rgb (n,H,W,3) u8, lab_q (n,3,H,W) f64 | None)."""
        n, L, A, M = self._prep(L_mc, ab, mask)
        out = self._pool.take((n, 2, self.H, self.W), np.float32)       # pinned: the library copies device -> result in place
        rgb = self._pool.take((n, self.H, self.W, 3), np.uint8)
        labq = self._pool.take((n, 3, self.H, self.W), np.float64) if want_lab else None
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_rgb(self._h, n, _fptr(L), _fptr(A), _fptr(M), float(maskcent), float(l_cent), _fptr(out),
                                           rgb.ctypes.data_as(ctypes.c_void_p),
                                           labq.ctypes.data_as(ctypes.c_void_p) if want_lab else None))
        return out, rgb, labq

    def forward_rgb_lazy(self, L_mc, ab, mask, maskcent=0.0, l_cent=50.0):
        """forward + the colour step on the device, only the uint8 image copied back: rgb (n,H,W,3).  The ab map and the refreshed Lab
        stay resident; ``fetch_outputs`` brings them over when somebody reads them (2.0 of the 2.2 MB a 256x256 click sends back).
        ``L_mc=None``: the L plane ``set_image_l`` left in the handle (one image; it is constant between the clicks on that image)."""
        if L_mc is None:
            A = np.asarray(ab)
            A = self._f32in(A[None] if A.ndim == 3 else A, (1, 2, self.H, self.W))
            M = np.asarray(mask)
            M = self._f32in(M[None] if M.ndim == 3 else M, (1, 1, self.H, self.W))
            n, Lp = 1, None
        else:
            n, L, A, M = self._prep(L_mc, ab, mask)
            Lp = _fptr(L)
        rgb = self._pool.take((n, self.H, self.W, 3), np.uint8)
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_rgb_lazy(self._h, n, Lp, _fptr(A), _fptr(M), float(maskcent), float(l_cent),
                                                rgb.ctypes.data_as(ctypes.c_void_p)))
        return rgb

    def fetch_outputs(self, n=1, want_ab=True, want_lab=True):
        """(out_ab (n,2,H,W) f32 | None, lab_q (n,3,H,W) f64 | None) of the last forward, from the device."""
        out = self._pool.take((n, 2, self.H, self.W), np.float32) if want_ab else None
        labq = self._pool.take((n, 3, self.H, self.W), np.float64) if want_lab else None
        self._chk(self.lib.idc_fetch_outputs(self._h, int(n), _fptr(out) if want_ab else None,
                                             labq.ctypes.data_as(ctypes.c_void_p) if want_lab else None))
        return out, labq

    def forward_dist313(self, L_mc, ab, mask, maskcent=0.0, want_dist=True):
        """313-bin head of the Caffe distribution net: returns (out_ab regression, pred_ab soft-decode (N,2,H,W),
        dist_S (N,313,H,W) full-resolution softmax(S*logits) or None)."""
        n, L, A, M = self._prep(L_mc, ab, mask)
        out = np.empty((n, 2, self.H, self.W), np.float32)
        pred = np.empty((n, 2, self.H, self.W), np.float32)
        dist = np.empty((n, 313, self.H, self.W), np.float32) if want_dist else None
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_dist313(self._h, n, _fptr(L), _fptr(A), _fptr(M), float(maskcent), _fptr(out),
                                               _fptr(pred), _fptr(dist) if want_dist else None))
        return out, pred, dist

    def set_dist_temperature(self, S):
        self._chk(self.lib.idc_set_dist_temperature(self._h, float(S)))

    def forward_device(self, n, d_L, d_ab, d_mask, d_out, maskcent=0.0, sync=False):
        """Device-pointer form (ints / objects with ``data_ptr()`` / numpy arrays over PINNED host memory from
        ``pinned_empty``, which the device reads and writes in place); enqueued on the handle stream."""
        def p(x):
            if hasattr(x, "data_ptr"):
                return ctypes.c_void_p(int(x.data_ptr()))
            if isinstance(x, np.ndarray):               # pinned host memory (pinned_empty): zero-copy, see idc_pipeline_times' note
                return ctypes.c_void_p(int(x.ctypes.data))
            return ctypes.c_void_p(int(x))
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_device(self._h, int(n), p(d_L), p(d_ab), p(d_mask), float(maskcent),
                                              p(d_out), 1 if sync else 0))

    def sync(self):
        self._chk(self.lib.idc_sync(self._h))

    # ---- stream ordering against a caller stream (e.g. torch.cuda.current_stream().cuda_stream) --------------
    def stream_wait(self, caller_stream):
        """Work enqueued on this handle after the call waits for everything already enqueued on ``caller_stream``."""
        self._chk(self.lib.idc_stream_wait(self._h, ctypes.c_void_p(int(caller_stream))))

    def stream_signal(self, caller_stream):
        """Work enqueued on ``caller_stream`` after the call waits for everything this handle has enqueued."""
        self._chk(self.lib.idc_stream_signal(self._h, ctypes.c_void_p(int(caller_stream))))

    # ---- overlapped host transfers: two slots ----------------------------------------------
    def pinned_empty(self, shape, dtype=np.float32):
        """numpy array over pinned host memory (``idc_alloc_host``): ``forward_async`` transfers it in place and
        ``forward_device`` may be given it directly (zero-copy: the memory is mapped into the device's address space).
        The memory lives as long as any array that views it (the base buffer frees it when collected), not as long as
        this engine."""
        shape = tuple(int(x) for x in shape)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        raw = np.asarray(_PinnedBuffer(self.lib, nbytes))           # uint8 view; .base keeps the owner alive
        return raw[:nbytes].view(dtype).reshape(shape)

    def forward_async(self, slot, L_mc, ab, mask, out, maskcent=0.0):
        """Enqueue one batch on pipeline slot 0/1 (float32 C-contiguous arrays, used in place -- keep them alive and
        untouched until ``wait(slot)``); ``out`` (n,2,H,W) float32 receives the result."""
        n = L_mc.shape[0]
        for a, shp in ((L_mc, (n, 1, self.H, self.W)), (ab, (n, 2, self.H, self.W)), (mask, (n, 1, self.H, self.W)), (out, (n, 2, self.H, self.W))):
            if a.dtype != np.float32 or not a.flags.c_contiguous or tuple(a.shape) != shp:
                raise ValueError("forward_async needs float32 C-contiguous arrays of shape %s" % (shp,))
        self._results_will_be_replaced()
        self._chk(self.lib.idc_forward_async(self._h, int(slot), int(n), _fptr(L_mc), _fptr(ab), _fptr(mask), float(maskcent), _fptr(out)))

    def wait(self, slot):
        self._chk(self.lib.idc_wait(self._h, int(slot)))

    def pipeline_times(self, slot):
        """ms since the pipeline's first use of (H2D start, H2D end, compute start, compute end, D2H start, D2H end) of the
        slot's last completed batch (after ``wait(slot)``)."""
        ms = np.zeros(6, np.float32)
        self._chk(self.lib.idc_pipeline_times(self._h, int(slot), _fptr(ms)))
        return ms

    # ---- RCCL weight broadcast through the C ABI (one call per rank) -------------------------------------------
    def comm_unique_id(self):
        buf = (ctypes.c_char * N.IDC_UNIQUE_ID_BYTES)()
        N.check(self.lib.idc_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def broadcast_weights(self, unique_id, rank, world, root=0):
        if len(unique_id) != N.IDC_UNIQUE_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % N.IDC_UNIQUE_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(unique_id), N.IDC_UNIQUE_ID_BYTES)
        self._chk(self.lib.idc_broadcast_weights(self._h, ctypes.cast(buf, ctypes.c_void_p), int(rank), int(world), int(root)))

    # ---- display / full-resolution step on the device (ui/gui_draw.py:280-283, colorize_image.py:123-158) ------
    def upsample_lab2rgb(self, L_out, source="output_ab", interp="cubic", img=0):
        """ab planes of slot ``img`` resized to ``L_out.shape`` and combined with that L plane -> (oh, ow, 3) uint8.
        source: 'output_ab' (refreshed, float64), 'output_ab_raw', 'input_ab'; interp: 'cubic' (cv2 INTER_CUBIC),
        'linear' / 'nearest' (scipy.ndimage.zoom order 1 / 0)."""
        Ld = np.ascontiguousarray(np.asarray(L_out, dtype=np.float64))
        if Ld.ndim == 3 and Ld.shape[0] == 1:
            Ld = Ld[0]
        oh, ow = Ld.shape
        rgb = np.empty((oh, ow, 3), np.uint8)
        src = {"output_ab": N.IDC_SRC_OUTPUT_AB, "output_ab_raw": N.IDC_SRC_OUTPUT_AB_RAW, "input_ab": N.IDC_SRC_INPUT_AB}[source]
        itp = {"cubic": N.IDC_INTERP_CUBIC, "linear": N.IDC_INTERP_LINEAR, "nearest": N.IDC_INTERP_NEAREST}[interp]
        self._chk(self.lib.idc_upsample_lab2rgb(self._h, int(img), src, itp, int(oh), int(ow), Ld.ctypes.data_as(ctypes.c_void_p),
                                                rgb.ctypes.data_as(ctypes.c_void_p)))
        return rgb

    @property
    def stream(self):
        return self.lib.idc_stream(self._h)

    # ---- introspection ----------------------------------------------------------------------
    def layer_table(self):
        rows = []
        for i in range(self.lib.idc_num_layers(self._h)):
            info = N.LayerInfo()
            self._chk(self.lib.idc_layer_info_get(self._h, i, ctypes.byref(info)))
            rows.append(dict(index=i, name=info.name.decode(), kernel=info.kernel.decode(), flops=info.flops,
                             min_bytes=info.min_bytes, launches=info.launches))
        return rows

    def set_profiling(self, on):
        self._chk(self.lib.idc_set_profiling(self._h, 2 if on == "forward" else (1 if on else 0)))

    def layer_times_ms(self):
        n = self.lib.idc_num_layers(self._h)
        ms = np.zeros(n, np.float32)
        self._chk(self.lib.idc_layer_times_ms(self._h, _fptr(ms), n))
        return ms

    def layer_times_stats(self):
        """(min, median, max) per layer over the forwards recorded since profiling was switched on (at most the last 32)."""
        n = self.lib.idc_num_layers(self._h)
        lo, med, hi = (np.zeros(n, np.float32) for _ in range(3))
        self._chk(self.lib.idc_layer_times_stats(self._h, _fptr(lo), _fptr(med), _fptr(hi), n))
        return lo, med, hi

    def activation(self, name, n=1):
        """NCHW fp32 copy of an intermediate tensor of the last forward (parity tests)."""
        cap = int(n) * 640 * self.H * self.W
        buf = np.empty(cap, np.float32)
        C, H, W = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._chk(self.lib.idc_get_activation(self._h, name.encode(), int(n), _fptr(buf), cap,
                                              ctypes.byref(C), ctypes.byref(H), ctypes.byref(W)))
        return buf[: n * C.value * H.value * W.value].reshape(n, C.value, H.value, W.value).copy()


# ---- single operators through the same kernels (used by the parity tests) ---------------------
def op_conv2d(x, weight, bias, dilation=1, in_stride=1, act=0, bn_scale=None, bn_shift=None, resid=None,
              precision="fp32", device=0):
    lib = N.load()
    x = _f32c(x); weight = _f32c(weight); bias = _f32c(bias)
    n, cin, h, w = x.shape
    cout, ksize = weight.shape[0], weight.shape[2]
    y = np.empty((n, cout, h // in_stride, w // in_stride), np.float32)
    opt = lambda a: _fptr(_f32c(a)) if a is not None else None
    keep = [opt(bn_scale), opt(bn_shift), opt(resid)]
    N.check(lib.idc_op_conv2d(device, _PREC[precision], n, cin, h, w, _fptr(x), cout, ksize, dilation, in_stride,
                              _fptr(weight), _fptr(bias), act, keep[0], keep[1], keep[2], _fptr(y)))
    return y


def op_deconv4x4s2(x, weight, bias, act=0, resid=None, precision="fp32", device=0):
    lib = N.load()
    x = _f32c(x); weight = _f32c(weight); bias = _f32c(bias)
    n, cin, h, w = x.shape
    cout = weight.shape[1]
    y = np.empty((n, cout, 2 * h, 2 * w), np.float32)
    r = _f32c(resid) if resid is not None else None
    N.check(lib.idc_op_deconv4x4s2(device, _PREC[precision], n, cin, h, w, _fptr(x), cout, _fptr(weight),
                                   _fptr(bias), act, _fptr(r) if r is not None else None, _fptr(y)))
    return y
