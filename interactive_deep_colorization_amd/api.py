"""The reference's model-wrapper API, re-implemented on top of the HIP engine.

Callers of the reference (``ideepcolor.py:60-74``, ``ui/gui_draw.py:109-113,250-286``, the two
notebooks) construct ``ColorizeImageTorch`` / ``ColorizeImageTorchDist`` / ``ColorizeImageCaffe``
objects from ``data/colorize_image.py`` and use: ``prep_net``, ``load_image``, ``set_image``,
``net_forward``, the ``get_*`` getters, ``get_ab_reccs`` and a handful of attributes
(SURVEY.md 8b).  This module provides classes with those names and that observable behaviour
-- including ``net_forward`` returning ``-1`` after printing a message when the image or the
net is missing (``data/colorize_image.py:85-90``) -- while ``self.net`` is a
:class:`~interactive_deep_colorization_amd.engine.HipColorizer`: hand-written gfx950 kernels
behind the C ABI of ``include/ideepcolor.h``.

    reference (CPU torch / caffe)                      here
    ColorizeImageTorch.prep_net     :216-233    torch.load -> idc_load_weights
    ColorizeImageTorch.net_forward  :249-268    idc_forward, then the same Lab->RGB post step
    ColorizeImageTorchDist          :279-372    idc_forward_dist (529-bin head on device)
    ColorizeImageCaffe              :375-442    same graph with Caffe i/o scaling

No CPU fallback exists: without the built library and a gfx950 device ``prep_net`` raises.
"""
from __future__ import print_function

import os

import numpy as np
from scipy.ndimage import zoom

from . import color_bins, colorspace
from .colorspace import lab2rgb_transpose, rgb2lab_transpose  # same helper names as the reference
from ._native import IdcError
from .engine import HipColorizer
from .workloads import put_point  # noqa: F401  notebook helper (DemoInteractiveColorization.ipynb:131-139)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))

# every public name of the reference's data/colorize_image.py (:10-36 helpers, :39-558 classes) + the notebook helper:
# `from interactive_deep_colorization_amd.api import *` and the `colorize_image` shim export exactly these
__all__ = ['create_temp_directory', 'lab2rgb_transpose', 'rgb2lab_transpose', 'put_point', 'read_state_dict',
           'ColorizeImageBase', 'ColorizeImageTorch', 'ColorizeImageTorchDist', 'ColorizeImageCaffe',
           'ColorizeImageCaffeGlobDist', 'ColorizeImageCaffeDist']


def create_temp_directory(path_template, N=1e8):
    """Make a fresh ``path_template % random_int`` directory (reference helper, ``:10-17``)."""
    print(path_template)
    while True:
        candidate = path_template % np.random.randint(0, N)
        if not os.path.exists(candidate):
            break
    print('Creating directory: %s' % candidate)
    os.mkdir(candidate)
    return candidate


def read_state_dict(path):
    """Weights file -> dict.  ``.pth`` goes through ``torch.load`` exactly like the reference
    (``:222-224``, including dropping ``_metadata``); ``.npz`` is accepted for torch-free use.
    The reference's InstanceNorm key patch (``:235-246``) has nothing to patch in this network."""
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            return dict((k, z[k]) for k in z.files)
    import torch
    sd = torch.load(path, map_location="cpu")
    if hasattr(sd, "_metadata"):
        del sd._metadata
    return sd


def read_caffe_weights(path, state_dict=None, net=None):
    """What the Caffe classes' ``prep_net(gpu_id, prototxt_path, caffemodel_path)`` loads -> ``(state_dict, out_mul)``.
    A ``.caffemodel`` (the reference's default invocation, ``ideepcolor.py:60-66`` / ``data/colorize_image.py:392-403``) is read
    straight from its protobuf wire format by :mod:`caffe_io` -- Caffe layer names mapped to the engine's keys, ``bw_conv1_1`` +
    ``ab_conv1_1`` merged, Caffe BatchNorm statistics un-scaled -- and ``out_mul`` is the net's final ``Scale`` blob (100);
    a ``.pth`` / ``.npz`` with the torch key names (e.g. the reference's converted ``caffemodel.pth``) or an explicit
    ``state_dict`` is taken as is, with the prototxt's 100."""
    if state_dict is not None:
        return state_dict, 100.
    from . import caffe_io
    if str(path).endswith(".caffemodel") or (not str(path).endswith((".pth", ".npz")) and caffe_io.is_caffemodel(path)):
        sd, info = caffe_io.read_caffemodel_state_dict(path, net=net)    # net: the prototxt of the calling class (layers it lacks are skipped, as Caffe does)
        if info["ignored"]:
            print('caffemodel layers not used (injected at load time or outside the path): %s' % ', '.join(info["ignored"]))
        return sd, (100. if info["out_mul"] is None else info["out_mul"])
    return read_state_dict(path), 100.


def _lazy(name, refresh):
    """Attribute that is materialised from device memory on first read after a forward (the reference fills these
    eagerly on the host in every ``net_forward``; here a click does not pay for copies nobody reads)."""
    slot = '_lazy_' + name

    def fget(self):
        getattr(self, refresh)()
        return self.__dict__.get(slot)

    def fset(self, value):
        self.__dict__[slot] = value

    return property(fget, fset)


_OUT_ATTRS = ('output_ab_raw', 'output_lab', 'output_ab')


def _lazy_out(name):
    """``output_ab_raw`` / ``output_lab`` / ``output_ab``: what ``net_forward`` leaves besides the image it returns.  The reference
    fills them on the host inside every call (``:263-267,196-198``); here they stay on the device (2.0 of the 2.2 MB a 256x256 click
    would send back) and are copied over the first time one of them is READ -- the same values, ordinary numpy arrays from then on.
    Assigning one (the reference's ``_set_out_ab_``, or a caller) makes it a plain attribute again."""
    slot = '_lazy_' + name

    def fget(self):
        self._refresh_outputs()
        try:
            return self.__dict__[slot]
        except KeyError:
            raise AttributeError(name)

    def fset(self, value):
        self.__dict__[slot] = value
        pend = self.__dict__.get('_out_pending')
        if pend:
            pend.discard(name)
        if name == 'output_ab':
            self.__dict__['_dev_out_valid'] = False          # a map the caller supplied: the device copy is no longer "the" output_ab

    return property(fget, fset)


class ColorizeImageBase(object):
    """Image state + getters shared by every backend (reference ``:39-198``).

    Backend classes define the five normalisation constants ``l_norm, ab_norm, l_mean,
    ab_mean, mask_mult`` before any image is set."""

    def __init__(self, Xd=256, Xfullres_max=10000):
        self.Xd = Xd
        self.Xfullres_max = Xfullres_max      # cap on the larger side of the full-res copy
        self.img_l_set = False
        self.net_set = False
        self.img_just_set = False
        self._l_resident = False              # the image's L plane is in the engine's slot 0
        self._hints_on_device = False         # input_ab / input_mask live on the device (net_forward_hints)

    def prep_net(self):
        raise Exception("Should be implemented by base class")

    def _new_engine(self, net):
        """A fresh engine handle replaces ``self.net``: nothing of the previous handle's device state (resident L plane,
        rasterised hints, resident distribution) exists in it, so every "already on the device" flag is cleared."""
        old = getattr(self, 'net', None)
        self.net = net
        self._l_resident = False
        self._hints_on_device = False
        self._dev_out_valid = False
        self._out_pending = set()
        if hasattr(self, '_dist_on_device'):
            self._dist_on_device = False
            self.dist_ab_set = False
        if old is not None and old is not net and hasattr(old, 'close'):
            old.close()
        self.net_set = True

    # input_ab / input_mask: plain attributes after net_forward; after net_forward_hints they are read back from the
    # device planes only when something (get_input_img, get_sup_img, ...) asks for them
    def _refresh_hint_planes(self):
        if self._hints_on_device:
            self._hints_on_device = False
            ab, mask = self.net.hint_planes(0)
            self.input_ab, self.input_mask = ab, mask / self.mask_mult

    input_ab = _lazy('input_ab', '_refresh_hint_planes')
    input_mask = _lazy('input_mask', '_refresh_hint_planes')

    # ------------------------------------------------------------------ image ingestion
    def _ingest(self, rgb_fullres, rgb_net):
        """Common tail of load_image / set_image: full-res Lab, net-res Lab, mean-centred L."""
        big = rgb_fullres
        longest = max(big.shape[0], big.shape[1])
        if longest > self.Xfullres_max:
            f = 1. * self.Xfullres_max / longest
            big = zoom(big, (f, f, 1), order=1)
        self.img_rgb_fullres = big
        self.img_lab_fullres = colorspace.rgb2lab(big).transpose((2, 0, 1))
        self.img_l_fullres = self.img_lab_fullres[[0]]
        self.img_ab_fullres = self.img_lab_fullres[1:]

        self.img_rgb = rgb_net
        self.img_lab = colorspace.rgb2lab(rgb_net).transpose((2, 0, 1))
        self.img_l = self.img_lab[[0]]
        self.img_ab = self.img_lab[1:]

        scale = np.array((self.l_norm, self.ab_norm, self.ab_norm))[:, None, None]
        shift = np.array((self.l_mean, self.ab_mean, self.ab_mean))[:, None, None] / scale
        self.img_lab_mc = self.img_lab / scale - shift
        self.img_l_mc = self.img_lab_mc[[0]]          # = L - 50 for every shipped backend
        # the plane every net_forward hands to the engine, converted once per image instead of once per click
        self._img_l_mc_f32 = np.ascontiguousarray(self.img_l_mc, dtype=np.float32)
        self._img_l_pinned = None
        self.img_l_set = True
        self._l_resident = False

    def _l_plane(self):
        """The float32 L plane of the current image in pinned memory (made once per image): every click uploads it in place."""
        if getattr(self, '_img_l_pinned', None) is None:
            alloc = getattr(self.net, 'pinned_empty', None)      # engines without a pinned allocator take the plain array
            if alloc is None:
                return self._img_l_mc_f32
            p = alloc(self._img_l_mc_f32.shape, np.float32)
            p[...] = self._img_l_mc_f32
            self._img_l_pinned = p
        return self._img_l_pinned

    def load_image(self, input_path):
        """Read a file, keep the full-res copy, bilinear-resize to Xd x Xd (``:52-66``)."""
        full = colorspace.imread_rgb(input_path)
        small = colorspace.resize_bilinear_u8(full, self.Xd, self.Xd)
        self._ingest(full.copy(), small)

    def set_image(self, input_image):
        """Take an already Xd x Xd RGB uint8 array; like the reference it is NOT resized (``:68-77``)."""
        self._ingest(input_image.copy(), input_image)

    # ------------------------------------------------------------------ forward bookkeeping
    def net_forward(self, input_ab, input_mask):
        """Guards + input bookkeeping (``:79-96``).  ab 2xXxX raw Lab units, mask 1xXxX (float or bool)."""
        for ready, complaint in ((self.img_l_set, 'I need to have an image!'),
                                 (self.net_set, 'I need to have a net!')):
            if not ready:
                print(complaint)
                return -1
        self._hints_on_device = False
        self.input_ab = input_ab
        self.input_mask = input_mask
        # (ab_mean 0 / ab_norm 1 / mask_mult 1 in the torch classes: the arithmetic is the identity, the arrays are passed on)
        self.input_ab_mc = input_ab if (self.ab_mean == 0 and self.ab_norm == 1) else (input_ab - self.ab_mean) / self.ab_norm
        self.input_mask_mult = input_mask if self.mask_mult == 1 else input_mask * self.mask_mult
        return 0

    def _stage_hints(self, hints, mode):
        """Guards of net_forward, then: L plane resident (uploaded once per image), hint list rasterised on the
        device -- what ``UIControl.get_input`` + ``rgb2lab`` (``ui/ui_control.py:177-187``, ``ui/gui_draw.py:273-277``)
        or the notebook's ``put_point`` do on the host before every forward."""
        for ready, complaint in ((self.img_l_set, 'I need to have an image!'),
                                 (self.net_set, 'I need to have a net!')):
            if not ready:
                print(complaint)
                return -1
        if self.ab_mean != 0 or self.ab_norm != 1:
            raise ValueError('device-side hints assume raw ab inputs (ab_mean 0, ab_norm 1)')
        self._ensure_l_resident()
        self.net.set_hints(hints, mode=mode, img=0, mask_value=self.mask_mult)
        self._hints_on_device = True
        return 0

    def net_forward_hints(self, hints, mode='rgb'):
        """``net_forward`` for a list of edits instead of rasterised planes (not in the reference: it removes the last
        per-click host work).  hints: rows ``(y0, x0, y1, x1, r, g, b)`` (mode 'rgb': uint8
        colours, rectangle corners inclusive as ``cv2.rectangle``) or ``(y0, x0, y1, x1, a, b)`` (mode 'ab').
        Returns what ``net_forward`` returns."""
        if self._stage_hints(hints, mode) == -1:
            return -1
        raw, rgb, lab_q = self.net.forward_resident(1, getattr(self, 'mask_cent', 0), l_cent=self.l_mean)
        self._l_serial = getattr(self.net, 'l_serial', None)
        return self._finish_forward(raw[0], rgb[0], lab_q[0])

    def _ensure_l_resident(self):
        """The image's L plane sits in the engine's slot 0: uploaded once per image (it is constant between the clicks,
        ``colorize_image.py:161-191`` sets it in set_image), and again whenever something else has since used the engine behind this
        object's back (``engine.l_serial`` moves with every call that may write the slot)."""
        if not self._l_resident or self.__dict__.get('_l_serial') != getattr(self.net, 'l_serial', None):
            self.net.set_image_l(self.img_l_mc, 0)
            self._l_resident = True
            self._l_serial = getattr(self.net, 'l_serial', None)

    output_ab_raw = _lazy_out('output_ab_raw')
    output_lab = _lazy_out('output_lab')
    output_ab = _lazy_out('output_ab')

    def _forward_rgb(self, maskcent):
        """forward + colour step; only the image crosses PCIe when the engine can keep the rest (``forward_rgb_lazy``), else the
        eager three-output call (injected engines of the host-logic tests)."""
        lazy = getattr(self.net, 'forward_rgb_lazy', None)
        if lazy is not None and hasattr(self.net, 'fetch_outputs'):
            # a new net_forward replaces the previous call's maps by definition: whatever of them was never read is dropped, not fetched
            self._out_pending = set()
            if getattr(self.net, 'before_overwrite', None) is not None:
                self.net.before_overwrite = None
            if hasattr(self.net, 'set_image_l') and hasattr(self.net, 'l_serial'):
                self._ensure_l_resident()            # then L_mc=None = "the resident plane": one H2D copy less per click
                rgb = lazy(None, self.input_ab_mc, self.input_mask_mult, maskcent, l_cent=self.l_mean)
                self._l_serial = self.net.l_serial
            else:
                rgb = lazy(self._l_plane(), self.input_ab_mc, self.input_mask_mult, maskcent, l_cent=self.l_mean)
            return self._finish_forward_lazy(rgb[0])
        raw, rgb, lab_q = self.net.forward_rgb(self._l_plane(), self.input_ab_mc, self.input_mask_mult, maskcent, l_cent=self.l_mean)
        return self._finish_forward(raw[0], rgb[0], lab_q[0])

    def _finish_forward_lazy(self, rgb):
        """``net_forward``'s return value is the uint8 image (``:264-268``); the ab map and the refreshed Lab were computed on the
        device by the same call and wait there (``engine.fetch_outputs``) until an attribute read asks for them."""
        self.output_rgb = rgb
        self._out_pending = set(_OUT_ATTRS)
        self._out_serial = getattr(self.net, 'forward_serial', None)
        self._dev_out_valid = True
        # anything else that is about to replace the engine's resident results (a direct engine call behind this object's back) first lets
        # this object fetch what it has not read yet: the attributes keep the reference's meaning whatever happens in between
        self.net.before_overwrite = self._refresh_outputs
        return self.output_rgb

    def _refresh_outputs(self):
        pend = self.__dict__.get('_out_pending')
        if not pend:
            return
        names = set(pend)
        #  the pending set is cleared only AFTER a successful fetch: if the serial check or the fetch raises, every later read of
        # output_ab / output_lab / output_ab_raw comes back here and raises again instead of handing out the maps of an older forward
        if getattr(self.net, 'before_overwrite', None) is not None:
            self.net.before_overwrite = None
        if getattr(self.net, 'forward_serial', None) != self.__dict__.get('_out_serial'):
            raise RuntimeError('the engine has run another forward since net_forward: the ab map of that call was never fetched '
                               '(read output_ab / output_lab / output_ab_raw before using the engine directly)')
        want_lab = bool(names & {'output_lab', 'output_ab'})
        raw, lab_q = self.net.fetch_outputs(1, want_ab='output_ab_raw' in names, want_lab=want_lab)
        if 'output_ab_raw' in names:
            self.__dict__['_lazy_output_ab_raw'] = raw[0]
        if 'output_lab' in names:
            self.__dict__['_lazy_output_lab'] = lab_q[0]
        if 'output_ab' in names:
            self.__dict__['_lazy_output_ab'] = lab_q[0][1:]
        pend.clear()

    def _finish_forward(self, raw_ab, rgb=None, lab_q=None):
        """Lab->RGB of the prediction, then refresh ``output_ab`` from the uint8 result -- the
        reference does this round trip too (``:264-267,196-198``), so ``output_ab`` is the
        quantised map while ``output_ab_raw`` (extra) is the net's own output.  Both colour steps run on
        the device (``idc_lab2rgb`` / fused in ``idc_forward_rgb``: float64, skimage's formulas)."""
        self.output_ab_raw = raw_ab
        if rgb is None:
            rgb, lab_q = self.net.lab2rgb(self.img_l[None], raw_ab[None])
            rgb, lab_q = rgb[0], lab_q[0]
        self.output_rgb = rgb
        self.output_lab = lab_q
        self.output_ab = lab_q[1:]
        # the same two maps are resident on the device (refreshed output_ab in float64, the hint planes): the display /
        # full-resolution getters read them there as long as the caller has not replaced these attributes
        self._dev_out_valid = True
        return self.output_rgb

    def _set_out_ab_(self):
        self.output_lab = rgb2lab_transpose(self.output_rgb)
        self.output_ab = self.output_lab[1:]

    # ------------------------------------------------------------------ getters (``:98-158``)
    def _zeros_ab(self, like):
        return np.zeros((2,) + tuple(like.shape[1:]))

    def _up(self, arr, order):
        """Zoom a CxXdxXd map to the full-res image size (bilinear order=1 / nearest order=0)."""
        fh = 1. * self.img_l_fullres.shape[1] / arr.shape[1]
        fw = 1. * self.img_l_fullres.shape[2] / arr.shape[2]
        return zoom(arr, (1, fh, fw), order=order)

    def get_result_PSNR(self, result=-1, return_SE_map=False):
        cur = self.get_img_forward() if np.array(result).flatten()[0] == -1 else result.copy()
        se = (1. * self.img_rgb - cur) ** 2
        psnr = 20 * np.log10(255. / np.sqrt(np.mean(se)))
        return (psnr, se) if return_SE_map else psnr

    def get_img_forward(self):
        return self.output_rgb

    def get_img_gray(self):
        return lab2rgb_transpose(self.img_l, self._zeros_ab(self.img_l))

    def get_img_gray_fullres(self):
        return lab2rgb_transpose(self.img_l_fullres, self._zeros_ab(self.img_l_fullres))

    def _out_on_device(self):
        return self.net_set and bool(self.__dict__.get('_dev_out_valid'))

    def _in_on_device(self):
        # only when the hint planes exist on the device alone (net_forward_hints): arrays the caller handed in may have
        # been edited in place since the last forward (put_point does), and the reference would show those edits
        return self.net_set and self._hints_on_device and self.ab_mean == 0 and self.ab_norm == 1

    def get_img_fullres(self):
        """Bilinear (``scipy.ndimage.zoom`` order 1) upsample of ``output_ab`` + Lab->RGB with the full-res L
        (``:123-131``) -- on the device when the map is still the one the last forward left there."""
        # (the Python-side token can outlive the engine's resident map -- a direct net.forward / forward_async on the
        #  engine, a want_rgb=False forward: the engine then answers IDC_ERR_UNSUPPORTED and the host path takes over)
        if self._out_on_device():
            try:
                return self.net.upsample_lab2rgb(self.img_l_fullres[0], 'output_ab', 'linear')
            except IdcError:
                self._dev_out_valid = False
        return lab2rgb_transpose(self.img_l_fullres, self._up(self.output_ab, 1))

    def get_input_img_fullres(self):
        if self._in_on_device():
            try:
                return self.net.upsample_lab2rgb(self.img_l_fullres[0], 'input_ab', 'linear')
            except IdcError:
                pass
        return lab2rgb_transpose(self.img_l_fullres, self._up(self.input_ab, 1))

    def get_result_window(self, l_win):
        """The display step of ``GUIDraw.compute_result`` (``ui/gui_draw.py:280-283``) on the device:
        ``cv2.resize(output_ab, (win_w, win_h), INTER_CUBIC)`` + ``lab2rgb`` with the window-sized L plane ``l_win``
        (win_h, win_w) -> (win_h, win_w, 3) uint8.  Not in the reference's model class (the GUI does it on the host
        with cv2 + skimage after every click); SURVEY.md 8f rank 1."""
        if not self._out_on_device():
            raise RuntimeError('get_result_window needs the result of the last net_forward (output_ab was replaced)')
        try:
            return self.net.upsample_lab2rgb(np.asarray(l_win), 'output_ab', 'cubic')
        except IdcError as ex:
            self._dev_out_valid = False
            raise RuntimeError('get_result_window: the engine no longer holds the last net_forward result (%s)' % ex)

    def get_input_img(self):
        return lab2rgb_transpose(self.img_l, self.input_ab)

    def get_img_mask(self):
        return lab2rgb_transpose(100. * (1 - self.input_mask), self._zeros_ab(self.img_l))

    def get_img_mask_fullres(self):
        m = self._up(self.input_mask, 0)
        return lab2rgb_transpose(100. * (1 - m), self._zeros_ab(m))

    def get_sup_img(self):
        return lab2rgb_transpose(50 * self.input_mask, self.input_ab)

    def get_sup_fullres(self):
        if self._in_on_device():
            try:
                return self.net.upsample_lab2rgb(50 * self._up(self.input_mask, 0)[0], 'input_ab', 'nearest')
            except IdcError:
                pass
        return lab2rgb_transpose(50 * self._up(self.input_mask, 0), self._up(self.input_ab, 0))


def _grid_529():
    """23x23 ab grid in the torch class's bin order (a varies fastest), ``:213,283``."""
    axis = np.arange(-110, 120, 10)
    return np.array(np.meshgrid(axis, axis)).reshape((2, 529)).T


class ColorizeImageTorch(ColorizeImageBase):
    """PyTorch-backend wrapper (``:201-276``) with the network on the MI355X.

    ``precision`` (not in the reference): ``'fp32'`` = exact-fp32 MFMA, the parity path (default);
    ``'bf16'`` = bf16 MFMA with fp32 accumulation, the throughput path."""

    def __init__(self, Xd=256, maskcent=False, precision='fp32'):
        print('ColorizeImageTorch instantiated')
        ColorizeImageBase.__init__(self, Xd)
        self.l_norm, self.ab_norm = 1., 1.
        self.l_mean, self.ab_mean = 50., 0.
        self.mask_mult = 1.
        self.mask_cent = .5 if maskcent else 0
        self.precision = precision
        self.pts_in_hull = _grid_529()

    def prep_net(self, gpu_id=None, path='', dist=False, state_dict=None):
        """``gpu_id=None`` selects device 0 (the reference's torch backend stayed on the CPU:
        ``ideepcolor.py:68-72``).  ``state_dict`` may replace ``path``."""
        print('path = %s' % path)
        print('Model set! dist mode? ', dist)
        sd = read_state_dict(path) if state_dict is None else state_dict
        net = HipColorizer(H=self.Xd, W=self.Xd, max_batch=1, precision=self.precision,
                           device=0 if gpu_id is None else int(gpu_id), dist=dist)
        net.load_state_dict(sd)
        self._new_engine(net)

    def net_forward(self, input_ab, input_mask):
        if ColorizeImageBase.net_forward(self, input_ab, input_mask) == -1:
            return -1
        # the device boundary -- stands for self.net.forward(...)[0].cpu().data.numpy() at :263
        return self._forward_rgb(self.mask_cent)


class ColorizeImageTorchDist(ColorizeImageTorch):
    """Regression + 529-bin colour distribution and colour suggestions (``:279-372``).

    The distribution stays on the device after ``net_forward``: ``get_ab_reccs`` runs there
    (``idc_suggest_colors``) and ``dist_ab`` / ``dist_ab_full`` / ``dist_ab_grid`` are copied out (and x4
    nearest-upsampled, ``model.py:131,160``) only when read -- the reference moves 529 x X x X floats to the host
    on every call."""

    def __init__(self, Xd=256, maskcent=False, precision='fp32'):
        ColorizeImageTorch.__init__(self, Xd, precision=precision)
        self.dist_ab_set = False
        self._dist_on_device = False
        self.pts_grid = _grid_529()
        self.in_hull = np.ones(529, dtype=bool)
        self.AB = 529
        self.A = self.B = 23
        self.dist_ab = None
        self.dist_ab_full = np.zeros((self.AB, Xd, Xd))
        self.dist_ab_grid = np.zeros((self.A, self.B, Xd, Xd))
        self.dist_entropy = np.zeros((Xd, Xd))
        self.mask_cent = .5 if maskcent else 0

    def _refresh_dist(self):
        if self._dist_on_device:
            self._dist_on_device = False
            dist_q = self.net.get_dist(1)[0]
            # device holds softmax(0.2*logits) at X/4; the reference's out_cl is its nearest x4 upsample
            self.dist_ab = np.repeat(np.repeat(dist_q, 4, axis=1), 4, axis=2)
            full = self.__dict__['_lazy_dist_ab_full']
            full[self.in_hull] = self.dist_ab
            self.dist_ab_grid = full.reshape((self.A, self.B, self.Xd, self.Xd))

    dist_ab = _lazy('dist_ab', '_refresh_dist')
    dist_ab_full = _lazy('dist_ab_full', '_refresh_dist')
    dist_ab_grid = _lazy('dist_ab_grid', '_refresh_dist')

    def prep_net(self, gpu_id=None, path='', dist=True, S=.2, state_dict=None):
        ColorizeImageTorch.prep_net(self, gpu_id=gpu_id, path=path, dist=dist, state_dict=state_dict)

    def net_forward(self, input_ab, input_mask):
        if ColorizeImageBase.net_forward(self, input_ab, input_mask) == -1:
            return -1
        out_ab, _ = self.net.forward_dist(self._l_plane(), self.input_ab_mc, self.input_mask_mult, self.mask_cent,
                                          want_dist=False)
        self._dist_on_device = True
        self.dist_ab_set = True
        # model.py:164-166: with dist=True the network returns ``out_reg * 110`` where out_reg is ALREADY the ab map
        # (tanh * 110); the reference hands that doubly scaled array back (nothing reads it) -- so does this.  The
        # ab map itself is ``output_ab_raw``.
        self.output_ab_raw = out_ab[0]
        return out_ab[0] * np.float32(110.)

    def net_forward_hints(self, hints, mode='rgb'):
        if self._stage_hints(hints, mode) == -1:
            return -1
        out_ab, _, _ = self.net.forward_resident(1, self.mask_cent, want_rgb=False)
        self._dist_on_device = True
        self.dist_ab_set = True
        self.output_ab_raw = out_ab[0]
        return out_ab[0] * np.float32(110.)

    def get_ab_reccs(self, h, w, K=5, N=25000, return_conf=False):
        """K suggested colours at pixel (h,w): N inverse-CDF draws from the predicted pdf, k-means, ordered by
        cluster occupancy (``:322-354``) -- on the device.  Like the reference the draw depends on numpy's global
        RNG state (one ``np.random.randint`` seeds the device generator); unlike sklearn's KMeans the clustering
        itself is deterministic."""
        if not self.dist_ab_set:
            print('Need to set prediction first')
            return 0
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        centers, conf = self.net.suggest_colors(h, w, self.pts_in_hull, K=K, N_draws=N, seed=seed)
        if return_conf:
            return centers, conf
        return centers

    def compute_entropy(self):
        self.dist_entropy = np.sum(self.dist_ab * np.log(self.dist_ab), axis=0)

    def plot_dist_grid(self, h, w):
        import matplotlib.pyplot as plt
        plt.figure()
        plt.imshow(self.dist_ab_grid[:, :, h, w], extent=[-110, 110, 110, -110], interpolation='nearest')
        plt.colorbar(); plt.ylabel('a'); plt.xlabel('b')

    def plot_dist_entropy(self):
        import matplotlib.pyplot as plt
        plt.figure()
        plt.imshow(-self.dist_entropy, interpolation='nearest')
        plt.colorbar()


class ColorizeImageCaffe(ColorizeImageBase):
    """Caffe-backend wrapper (``:375-442``) on the same kernels.

    The Caffe net (``models/reference_model/deploy_nodist.prototxt``) is the same graph with the
    input normalisation folded into its trained weights: it is fed raw ``L-50``, raw ab and
    ``mask*110`` (``:379-383,425``) and ends in ``TanH -> Scale 100`` (prototxt ``:812-821``).
    Caffe cannot be installed here, so ``prep_net`` takes the weights as a ``state_dict`` under
    the torch key names (e.g. the reference's converted ``caffemodel.pth``) or -- the
    reference's default invocation -- a real ``.caffemodel`` (read from its protobuf wire format by :mod:`caffe_io`,
    no Caffe needed), passed as ``caffemodel_path``, or an explicit ``state_dict``; ``prototxt_path`` is accepted
    and not read (the three graphs of the reference's prototxts are built in, selected by the class)."""

    def __init__(self, Xd=256, precision='fp32', color_bins_dir=None):
        """``color_bins_dir`` (not in the reference): a directory holding ``pts_in_hull.npy`` / ``pts_grid.npy`` /
        ``in_hull.npy`` (the reference reads ``./data/color_bins``, ``:398-399,486-489``); ``None`` = the tables built
        into :mod:`color_bins`.  A directory that lacks one of the files raises here, not at first use."""
        print('ColorizeImageCaffe instantiated')
        ColorizeImageBase.__init__(self, Xd)
        self.l_norm, self.ab_norm = 1., 1.
        self.l_mean, self.ab_mean = 50., 0.
        self.mask_mult = 110.
        self.precision = precision
        self.pred_ab_layer = 'pred_ab'
        self.pts_in_hull_path = os.path.join(color_bins_dir, 'pts_in_hull.npy') if color_bins_dir else '<built in>'
        self.pts_in_hull, self._pts_grid_table, self._in_hull_table = color_bins.load(color_bins_dir)

    _global_hints = False
    _dist313 = False

    def prep_net(self, gpu_id, prototxt_path='', caffemodel_path='', state_dict=None):
        print('gpu_id = %d, net_path = %s, model_path = %s' % (gpu_id, prototxt_path, caffemodel_path))
        if gpu_id == -1:
            raise RuntimeError('cpu mode is not available: this backend runs on gfx950 only')
        sd, out_mul = read_caffe_weights(caffemodel_path, state_dict, net="global" if self._global_hints else ("nopred" if self._dist313 else "nodist"))
        out_mul = self.__dict__.pop('_file_out_mul', out_mul)      # (a subclass that already read the file passes what it found)
        self.gpu_id = gpu_id
        net = HipColorizer(H=self.Xd, W=self.Xd, max_batch=1, precision=self.precision, device=int(gpu_id),
                           global_hints=self._global_hints, dist313=self._dist313)
        net.set_io_scales(l_div=1., ab_div=1., mask_mul=1., out_mul=out_mul)
        net.load_state_dict(sd)
        self._new_engine(net)

    def net_forward(self, input_ab, input_mask):
        if ColorizeImageBase.net_forward(self, input_ab, input_mask) == -1:
            return -1
        return self._forward_rgb(0.0)


class ColorizeImageCaffeGlobDist(ColorizeImageCaffe):
    """Caffe colorization with an additional global ab histogram as input (``:445-463``): the Global-Hints
    net ``models/global_model/deploy_nodist.prototxt``.  Its ``conv1_1`` sees only L (``bw_conv1_1``,
    prototxt ``:189-202``; the ab/mask planes are silenced ``:28-32``), and a 314-d histogram+flag vector
    (plus a 2-d saturation vector the wrapper leaves at zero) runs through four 1x1 conv/ReLU/BN stages and
    is added to ``conv4_3norm`` at every position (``:37-172,501-518``).  The state_dict uses the torch key
    names for the trunk plus ``glob.*`` for the branch (see ``include/ideepcolor.h``); a ``model1.0.weight``
    with one input channel (L only) is widened to four with zero ab/mask columns."""
    _global_hints = True

    def __init__(self, Xd=256, precision='fp32', color_bins_dir=None):
        ColorizeImageCaffe.__init__(self, Xd, precision=precision, color_bins_dir=color_bins_dir)
        self.glob_mask_mult = 1.
        self.glob_layer = 'glob_ab_313_mask'

    def prep_net(self, gpu_id, prototxt_path='', caffemodel_path='', state_dict=None):
        sd, self._file_out_mul = read_caffe_weights(caffemodel_path, state_dict, net='global')
        sd = dict(sd)
        w = np.asarray(sd['model1.0.weight'])
        if w.shape[1] == 1:                                   # bw_conv1_1 only: ab / mask never reach the net
            w4 = np.zeros((w.shape[0], 4) + tuple(w.shape[2:]), np.float32)
            w4[:, :1] = w
            sd['model1.0.weight'] = w4
        ColorizeImageCaffe.prep_net(self, gpu_id, prototxt_path, caffemodel_path, state_dict=sd)

    def get_global_histogram(self, ref_rgb, pts_in_hull=None):
        """The notebook's ``get_global_histogram`` (``DemoGlobalHistogramTransfer.ipynb:176-186``), i.e. the
        ``global_stats.prototxt`` net, on the device: an Xd x Xd RGB uint8 reference image -> the 313-bin global ab
        histogram to pass as ``glob_dist``.  ``pts_in_hull`` defaults to the table loaded like the reference does."""
        centres = self.pts_in_hull if pts_in_hull is None else pts_in_hull
        hist, _ = self.net.global_histogram(ref_rgb, np.asarray(centres, np.float32))
        return hist[0]

    def _set_glob(self, glob_dist):
        # glob_dist is a 313 array, or -1 = "no global hint": histogram and flag all zero (reference :451-459)
        g = np.zeros((1, 314), np.float32)
        if np.array(glob_dist).flatten()[0] != -1:
            g[0, :313] = np.asarray(glob_dist, np.float32).ravel()
            g[0, 313] = self.glob_mask_mult
        self.net.set_global_hints(g)

    def net_forward_hints(self, hints, mode='rgb', glob_dist=-1):
        """Edit-list form; the global hint is an argument here too (default -1 = none), never a leftover of an earlier
        call."""
        if not self.net_set:
            print('I need to have a net!')
            return -1
        self._set_glob(glob_dist)
        ret = ColorizeImageCaffe.net_forward_hints(self, hints, mode)
        if isinstance(ret, int):
            return ret
        self._set_out_ab_()
        return ret

    def _set_out_ab_(self):
        # the reference refreshes output_ab from output_rgb once more after this class's net_forward (:461-463);
        # _finish_forward has just done exactly that on the device (the float64 map still resident there), so the
        # attributes and the device token stay as they are -- get_result_window / get_img_fullres keep working
        if not self.__dict__.get('_dev_out_valid'):
            ColorizeImageCaffe._set_out_ab_(self)

    def net_forward(self, input_ab, input_mask, glob_dist=-1):
        if not self.net_set:
            print('I need to have a net!')
            return -1
        self._set_glob(glob_dist)
        self.output_rgb = ColorizeImageCaffe.net_forward(self, input_ab, input_mask)
        if isinstance(self.output_rgb, int):
            return self.output_rgb
        self._set_out_ab_()
        return self.output_rgb


class ColorizeImageCaffeDist(ColorizeImageCaffe):
    """Caffe model which includes distribution prediction (``:466-561``): the 313-bin net
    ``models/reference_model/deploy_nopred.prototxt``.  ``net_forward`` returns the colourised image built from
    ``pred_ab`` (the annealed-mean decode ``sum_q softmax(2.6 l)_q * pts_in_hull[q]``, prototxt ``:826-850``) and
    keeps ``dist_ab`` = ``dist_ab_S`` (313, X, X) = ``softmax(S * l)`` for ``get_ab_reccs``.  The state_dict carries
    the trunk under the torch key names plus ``pred.*`` (see ``include/ideepcolor.h``); when ``pred.pred_ab.weight``
    is absent it is set from ``pts_in_hull`` exactly as the reference does at load time (``:405-407``)."""
    _dist313 = True

    def __init__(self, Xd=256, precision='fp32', color_bins_dir=None):
        ColorizeImageCaffe.__init__(self, Xd, precision=precision, color_bins_dir=color_bins_dir)
        self.dist_ab_set = False
        self._dist_on_device = False
        self.dist_ab = None
        self.scale_S_layer = 'scale_S'
        self.dist_ab_S_layer = 'dist_ab_S'
        self.pts_grid = self._pts_grid_table          # 529x2, all points
        self.in_hull = self._in_hull_table            # 529 bool
        self.AB, self.A, self.B = 529, 23, 23
        self.dist_ab_full = np.zeros((self.AB, self.Xd, self.Xd))
        self.dist_ab_grid = np.zeros((self.A, self.B, self.Xd, self.Xd))
        self.dist_entropy = np.zeros((self.Xd, self.Xd))

    def prep_net(self, gpu_id, prototxt_path='', caffemodel_path='', S=.2, state_dict=None):
        sd, self._file_out_mul = read_caffe_weights(caffemodel_path, state_dict, net="nopred")
        sd = dict(sd)
        if 'pred.pred_ab.weight' not in sd:
            print('Setting ab cluster centers in layer: %s' % self.pred_ab_layer)
            sd['pred.pred_ab.weight'] = np.ascontiguousarray(np.asarray(self.pts_in_hull, np.float32).T)[:, :, None, None]
        sd.setdefault('pred.pred_ab.bias', np.zeros(2, np.float32))
        self._centres = np.asarray(sd['pred.pred_ab.weight'], np.float32)[:, :, 0, 0].T          # 313x2
        ColorizeImageCaffe.prep_net(self, gpu_id, prototxt_path, caffemodel_path, state_dict=sd)
        self.S = S
        self.net.set_dist_temperature(S)
        self.net.keep_dist(True)          # dist_ab_S stays on the device after every forward

    def _refresh_dist(self):
        if self._dist_on_device:
            self._dist_on_device = False
            self.dist_ab = self.net.get_dist(1)[0]               # in-gamut, 313 x X x X
            full = self.__dict__['_lazy_dist_ab_full']           # full 529 grid, as the reference keeps it (:496-499)
            full[self.in_hull, :, :] = self.dist_ab
            self.dist_ab_grid = full.reshape((self.A, self.B, self.Xd, self.Xd))

    dist_ab = _lazy('dist_ab', '_refresh_dist')
    dist_ab_full = _lazy('dist_ab_full', '_refresh_dist')
    dist_ab_grid = _lazy('dist_ab_grid', '_refresh_dist')

    def net_forward(self, input_ab, input_mask):
        if ColorizeImageBase.net_forward(self, input_ab, input_mask) == -1:
            return -1
        _, pred, _ = self.net.forward_dist313(self._l_plane(), self.input_ab_mc, self.input_mask_mult, 0.0, want_dist=False)
        ret = self._finish_forward(pred[0])
        self._dist_on_device = True
        self.dist_ab_set = True
        return ret

    def net_forward_hints(self, hints, mode='rgb'):
        """Edits rasterised on the device, then the regular 313-head forward on the read-back planes."""
        if self._stage_hints(hints, mode) == -1:
            return -1
        return self.net_forward(self.input_ab, self.input_mask)

    def get_ab_reccs(self, h, w, K=5, N=25000, return_conf=False):
        """Recommended colours at (h, w): N draws of the 313-bin pdf, k-means, sorted by occupancy (``:509-543``) -- on
        the device-resident ``dist_ab_S`` (see ``ColorizeImageTorchDist.get_ab_reccs``)."""
        if not self.dist_ab_set:
            print('Need to set prediction first')
            return 0
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        centers, conf = self.net.suggest_colors(h, w, self._centres, K=K, N_draws=N, seed=seed)
        return (centers, conf) if return_conf else centers

    def compute_entropy(self):
        self.dist_entropy = np.sum(self.dist_ab * np.log(self.dist_ab), axis=0)

    def plot_dist_grid(self, h, w):
        """Plots the (23 x 23 grid) distribution at pixel (h, w) (``:549-555``)."""
        import matplotlib.pyplot as plt
        plt.figure()
        plt.imshow(self.dist_ab_grid[:, :, h, w], extent=[-110, 110, 110, -110], interpolation='nearest')
        plt.colorbar(); plt.ylabel('a'); plt.xlabel('b')

    def plot_dist_entropy(self):
        """Plots the per-pixel entropy map of the predicted distribution (``:557-561``)."""
        import matplotlib.pyplot as plt
        plt.figure()
        plt.imshow(-self.dist_entropy, interpolation='nearest')
        plt.colorbar()
