"""Tensor-level wrappers over the C ABI (include/cwf_hip.h).  torch is used for device memory and the current
stream only; every arithmetic op below is a launch of a hand-written gfx950 kernel.  There is no fallback:
constructing HipBackend without the built library or without a GPU raises.

Activation tensors are 5-D [N, D, H, W, C] with unit channel stride; a channel slice of a dense buffer is
passed zero-copy as (data_ptr, ldc = stride(3)).
"""
from __future__ import annotations

import ctypes

import threading

import torch

from . import _lib
from . import packing as pk

_f32 = torch.float32


def cl(t: torch.Tensor):
    """(tensor, ldc) of a channels-last activation usable by the kernels; copies only if the strides are unusable."""
    assert t.dim() == 5 and t.dtype == _f32, (t.shape, t.dtype)
    n, d, h, w, c = t.shape
    s = t.stride()
    ldc = s[3]
    ok = s[4] == 1 and ldc >= c and ldc % 4 == 0 and s[2] == w * ldc and s[1] == h * w * ldc and s[0] == d * h * w * ldc \
        and t.data_ptr() % 16 == 0
    if not ok:
        t = t.contiguous()
        ldc = c
        if c % 4:
            raise ValueError("channel count must be a multiple of 4 for a copied activation")
    return t, ldc


def _p(t):
    return 0 if t is None else t.data_ptr()


def _tab(ts):
    """pointer table (up to 4 groups) for the grouped kernel forms"""
    return (ctypes.c_void_p * 4)(*([t.data_ptr() for t in ts] + [0] * (4 - len(ts))))


def _ln_params(**kw):
    """struct cwf_ln_group_params from tensors or lists of tensors; returns (struct, groups)"""
    P = _lib.LnGroupParams()
    G = 1
    for name, v in kw.items():
        if v is None:
            continue
        vs = v if isinstance(v, (list, tuple)) else [v]
        G = max(G, len(vs))
        setattr(P, name, _tab(vs))
    return P, G


# Arithmetic of the conv family (forward, data gradient, weight gradient).  Activations / weights are fp32 in HBM in
# every mode; the mode selects the MFMA operand form:
#   "fp32"   v_mfma_f32_16x16x4_f32, exact f32 (the parity reference mode)
#   "bf16x3" split-bf16: hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate (~2^-16 per product)
#   "bf16"   single bf16 product (2^-9 per product)
_PRECISION = "fp32"
_WGRAD_PRECISION = None      # None = follow _PRECISION; "bf16" = single-bf16 products for the weight gradients only (see set_precision)
_DGRAD_PRECISION = None      # the same for the data-gradient convolutions


def set_precision(mode: str, wgrad: str = None, dgrad: str = None):
    """`wgrad` / `dgrad` optionally select the operand form of the weight-gradient / data-gradient kernels alone ("bf16": one
    bf16 product, fp32 accumulate -- the usual mixed-precision choice for gradients), leaving the forward in `mode`."""
    global _PRECISION, _WGRAD_PRECISION, _DGRAD_PRECISION
    ok = ("fp32", "bf16x3", "bf16")
    if mode not in ok or wgrad not in (None,) + ok or dgrad not in (None,) + ok:
        raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")
    for o in (wgrad, dgrad):
        if o is not None and (mode == "fp32") != (o == "fp32"):
            raise ValueError("the fp32 and the bf16 kernel families use different weight packings; choose all from one family")
    _PRECISION, _WGRAD_PRECISION, _DGRAD_PRECISION = mode, wgrad, dgrad


def precision() -> str:
    return _PRECISION


_raw_stream = torch._C._cuda_getCurrentRawStream
_current_device = torch._C._cuda_getDevice


def _priority_stream(device, which):
    """A stream of the lowest ("low") / highest ("high") priority the HIP runtime offers ("normal": torch's default pool).
    torch.cuda.Stream.priority_range() reports (0, -1) on ROCm -- no low priority -- so "low" goes to the runtime directly."""
    if which == "normal":
        return torch.cuda.Stream(device=device)
    if which == "high":
        return torch.cuda.Stream(device=device, priority=torch.cuda.Stream.priority_range()[1])
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    with torch.cuda.device(device):
        if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0 or least.value <= 0:
            return torch.cuda.Stream(device=device, priority=torch.cuda.Stream.priority_range()[1])   # no low level: any non-default pool
        h = ctypes.c_void_p()
        if hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(least.value)) != 0:   # 1 = hipStreamNonBlocking
            raise RuntimeError("hipStreamCreateWithPriority failed")
    return torch.cuda.ExternalStream(h.value, device=device)


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CwfError("no GPU visible: the cwf kernels are gfx950-only and there is no CPU fallback")
        self._ws = {}
        self._arena = {}
        self._arena_off = {}
        self._rng = {}                         # device -> int64[2] {seed, step} (device-resident generator state)
        self._rng_seed = {}
        self._site = 0                         # per-step dropout-site offset (reset by begin_step, static across steps)
        self._wg_stream = {}                   # device -> side stream for weight gradients (opt-in, see wgrad_stream)
        self._wg_part = {}                     # layer key -> persistent split-K slab buffer (wgrad_to)
        self._wg_pending = []                  # descriptor rows awaiting the batched reduce (wgrad_flush)
        self._wg_held = []                     # launches held back while wgrad_defer is set (wgrad_release)
        self._wg_keep = []                     # operands of released launches, referenced until join_wgrad_stream
        import os
        self.flush_every_default = int(os.environ.get("CWF_FLUSH_EVERY", "10"))
        self.flush_every = self.flush_every_default
        self.wgrad_defer = False
        self._wg_tables = {}
        self.wgrad_async = False
        self._rng_lock = threading.Lock()      # the autograd engine may call in from its own thread
        self._no_stem = bool(os.environ.get("CWF_NO_STEM_KERNEL"))      # A/B: the stem on conv16s (K slots 3/4 empty)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _stream():
        # raw hipStream_t of the current torch stream.  torch.cuda.current_stream() costs ~8 us of Python per call (device
        # index resolution, os.environ look-ups) -- with ~470 launches per forward that was 10 % of the host time.
        return _raw_stream(_current_device())

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise _lib.CwfError("%s failed with status %d" % (name, rc))

    def workspace(self, key, nfloats, device):
        """Grow-only scratch buffers (wgrad partial slabs), one per (device, stream): reuse is stream-ordered."""
        # (buffers allocated while capturing live in the graph's private pool: keep them apart from the eager ones)
        k = (key, device, _raw_stream(_current_device()), torch.cuda.is_current_stream_capturing())
        buf = self._ws.get(k)
        if buf is None or buf.numel() < nfloats:
            buf = torch.empty(int(nfloats), dtype=_f32, device=device)
            self._ws[k] = buf
        return buf

    # Small f64 accumulators (InstanceNorm / loss sums) come from ONE arena that is zeroed once per step (begin_step, called
    # by the model's forward) instead of ~340 separate fill launches.  Regions are handed out once per step and never reused
    # within it; the arena is sized for forward + backward of a step with a generous margin and falls back to torch.zeros.
    ARENA_DOUBLES = 1 << 20

    # ------------------------------------------------------------------ dropout generator (K12)
    def rng(self, device):
        """Device generator state {seed, step}.  The seed follows torch.initial_seed(); the step word is advanced by a kernel
        in begin_step, so that a captured step draws fresh masks on every replay."""
        device = torch.device(device)
        st = self._rng.get(device)
        seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        if st is None or (self._rng_seed.get(device) != seed and not torch.cuda.is_current_stream_capturing()):
            st = torch.tensor([seed, 0], dtype=torch.int64, device=device)
            self._rng[device] = st
            self._rng_seed[device] = seed
        return st

    def rng_site(self, n):
        """Counter offset of a dropout site of n elements (two chained masks may be drawn: 2n counters)."""
        with self._rng_lock:
            off = self._site
            self._site = off + 2 * int(n)
        return off

    def begin_step(self, device):
        self._call("cwf_rng_advance", self.rng(device).data_ptr(), self._stream())
        self._site = 0
        a = self._arena.get(device)
        if a is None:
            a = torch.zeros(self.ARENA_DOUBLES, dtype=torch.float64, device=device)
            self._arena[device] = a
        else:
            a.zero_()
        self._arena_off[device] = 0

    def _zeros_f64(self, shape, device):
        n = 1
        for s in shape:
            n *= s
        device = torch.device(device)
        a = self._arena.get(device)
        if a is not None:
            off = self._arena_off[device]
            if off + n <= a.numel():
                self._arena_off[device] = off + ((n + 1) // 2) * 2         # keep 16-byte alignment
                return a[off:off + n].view(shape)
        return torch.zeros(shape, dtype=torch.float64, device=device)

    # ------------------------------------------------------------------ K1
    def conv(self, op, x, wpk, bias, cout, in_scale=None, in_shift=None, slope=1.0, residual=None, out_scale=None,
             stats=None, out=None, w_ref=None, out_channels_alloc=None, fwd_op=None, prec=None, nb=None, bias_ref=None, x16=None, y16=None):
        """Returns the output buffer.  With out_channels_alloc > cout the buffer has zero-filled padding channels
        (2-channel heads live in 4-channel tensors so that every later kernel sees 16-byte voxel rows).
        w_ref / bias_ref (the original parameters; tuples for a fused layer) and fwd_op are not read here: the kernels take the
        packed operands."""
        x, x_ldc = cl(x)
        n, di, hi, wi, cin = x.shape
        if op == pk.CONV3_S2_DGRAD or op == pk.CONVT2_DGRAD:
            assert out is not None, "dgrad ops need an explicit output (its extent is not implied)"
        if out is None:
            do, ho, wo = pk.out_dims(op, di, hi, wi)
            ca = out_channels_alloc or cout
            # padding channels (2-channel head logits in 4-channel rows) are NOT initialised: every consumer of such a tensor reads
            # channels [0, cout) only (upsample_softmax, cwf_head_loss_*); the GRADIENT tensors of the same shape are written with
            # zero padding by their producers, because the data-gradient conv does read all allocated channels
            out = torch.empty((n, do, ho, wo, ca), dtype=_f32, device=x.device)
        y = out
        _, do, ho, wo, _ = y.shape
        y_ldc = y.stride(3)
        r_ldc = 0
        if residual is not None:
            residual, r_ldc = cl(residual)
        mode = prec or ((_DGRAD_PRECISION or _PRECISION) if (fwd_op is not None) else _PRECISION)
        if x16 is not None:
            # the input as a bf16 image (bf16_dgrad_ok): conv16s with LDS-DMA loaders; x itself is not read
            assert mode == "bf16" and in_scale is None and out_scale is None and x16.dtype == torch.bfloat16 and x16.is_contiguous()
            assert tuple(x16.shape) == (n, di, hi, wi, 16) and cin == 16 and cout == 16
            nbp = (0, 0, 0, 0, 1.0)
            if nb is not None:
                nb_x, nb_scale, nb_shift, nb_slope = nb
                nb_x, nb_ldc = cl(nb_x)
                nbp = (nb_x.data_ptr(), nb_ldc, nb_scale.data_ptr(), nb_shift.data_ptr(), float(nb_slope))
            self._call("cwf_conv_mfma_bf16_in16", op, x16.data_ptr(), self.zero16(x.device).data_ptr(), wpk.data_ptr(), _p(bias),
                       y.data_ptr(), y_ldc, _p(residual), r_ldc, _p(stats), *nbp, n, di, hi, wi, self._stream())
            return y
        if y16 is not None:
            # 1x1x1 forward that also leaves its output as a bf16 image (y16 [N,D,H,W,cout] bfloat16): the pointwise stream kernel's side
            # output; any other layer converts afterwards
            assert x16 is None and nb is None and out_scale is None and y16.dtype == torch.bfloat16 and y16.is_contiguous()
            rc = 1
            if mode != "fp32" and op == pk.CONV1:
                rc = self.lib.cwf_conv_mfma_bf16_y16(op, 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, wpk.data_ptr(), _p(bias), y.data_ptr(), y_ldc,
                                                     y16.data_ptr(), _p(in_scale), _p(in_shift), float(slope), _p(residual), r_ldc, _p(stats),
                                                     n, di, hi, wi, cin, do, ho, wo, cout, self._stream())
            if rc != 0:
                self.conv(op, x, wpk, bias, cout, in_scale, in_shift, slope, residual, None, stats, out=y, w_ref=w_ref, prec=prec, bias_ref=bias_ref)
                self.to_bf16(y if y.shape[-1] == cout else y[..., :cout], out=y16)      # (no copy node: a captured step stays plan-able)
            return y
        if (op == pk.CONV3_S1 and cin == 4 and cout == 16 and mode != "fp32" and fwd_op is None and in_scale is None and residual is None
                and nb is None and torch.is_tensor(w_ref) and tuple(w_ref.shape) == (16, 4, 3, 3, 3) and w_ref.is_contiguous()
                and w_ref.dtype == _f32 and do * ho * wo >= 32768 and not self._no_stem):
            # the stem: K = 8 taps x 4 channels (conv_stem.hip), straight from the raw weight
            self._call("cwf_conv_stem_bf16", 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, w_ref.data_ptr(), _p(bias), y.data_ptr(), y_ldc,
                       _p(out_scale), _p(stats), n, di, hi, wi, self._stream())
            return y
        if (op == pk.CONV3_S2 and cin == 16 and cout == 32 and mode != "fp32" and fwd_op is None and in_scale is None and residual is None
                and out_scale is None and nb is None and torch.is_tensor(w_ref) and tuple(w_ref.shape) == (32, 16, 3, 3, 3) and w_ref.is_contiguous()
                and w_ref.dtype == _f32 and do * ho * wo >= 32768 and not self._no_stem
                and (do, ho, wo) == ((di + 1) // 2, (hi + 1) // 2, (wi + 1) // 2)):
            # the first down-sampling layer: persistent prefetching kernel with parity-split halo rows (conv_s2.hip)
            self._call("cwf_conv_s2c16_bf16", 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, w_ref.data_ptr(), _p(bias), y.data_ptr(), y_ldc,
                       _p(stats), n, di, hi, wi, self._stream())
            return y
        if mode == "fp32":
            self._call("cwf_conv_mfma", op, x.data_ptr(), x_ldc, wpk.data_ptr(), _p(bias), y.data_ptr(), y_ldc,
                       _p(in_scale), _p(in_shift), float(slope), _p(residual), r_ldc, _p(out_scale), _p(stats),
                       n, di, hi, wi, cin, do, ho, wo, cout, self._stream())
        elif nb is not None:
            # data gradient whose output feeds the backward of act(IN(nb_x)): stats := (S1, S2) of that backward (see in_bwd_fused)
            nb_x, nb_scale, nb_shift, nb_slope = nb
            nb_x, nb_ldc = cl(nb_x)
            self._call("cwf_conv_mfma_bf16_nb", op, 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, wpk.data_ptr(), _p(bias),
                       y.data_ptr(), y_ldc, _p(in_scale), _p(in_shift), float(slope), _p(residual), r_ldc, _p(out_scale),
                       _p(stats), nb_x.data_ptr(), nb_ldc, nb_scale.data_ptr(), nb_shift.data_ptr(), float(nb_slope),
                       n, di, hi, wi, cin, do, ho, wo, cout, self._stream())
        else:
            self._call("cwf_conv_mfma_bf16", op, 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, wpk.data_ptr(), _p(bias),
                       y.data_ptr(), y_ldc, _p(in_scale), _p(in_shift), float(slope), _p(residual), r_ldc, _p(out_scale),
                       _p(stats), n, di, hi, wi, cin, do, ho, wo, cout, self._stream())
        return y

    def supports_fused_norm_bwd(self, prec=None):
        """The split-bf16 conv kernels can accumulate the InstanceNorm-backward sums in the data-gradient epilogue."""
        return (prec or _PRECISION) != "fp32"

    def in_bwd_apply(self, dy, x, scale, shift, slope, sums, dx_add=None):
        """Second half of in_bwd, given the (S1, S2) sums (from the data-gradient epilogue, conv(..., nb=...))."""
        dy, dy_ldc = cl(dy)
        x, x_ldc = cl(x)
        n, d, h, w, c = x.shape
        dx = torch.empty((n, d, h, w, c), dtype=_f32, device=x.device)
        a_ldc = 0
        if dx_add is not None:
            dx_add, a_ldc = cl(dx_add)
        self._call("cwf_in_bwd_apply", dy.data_ptr(), dy_ldc, x.data_ptr(), x_ldc, scale.data_ptr(), shift.data_ptr(), float(slope),
                   sums.data_ptr(), _p(dx_add), a_ldc, dx.data_ptr(), c, n, d * h * w, c, self._stream())
        return dx

    # ------------------------------------------------------------------ bf16 operand images (16-channel full-resolution layers)
    # The weight gradient of the full-resolution 16-channel 3x3x3 layers takes its operands as bf16 tensors [N,D,H,W,16] when both
    # exist (cwf_wgrad16_bf16: LDS-DMA staging, half the bytes, no conversion): xa16 = bf16(act(IN(x))) is a side output of the
    # layer's own InstanceNorm-backward apply pass, dy16 of the pass that produced the incoming gradient.  Same operand values as
    # the fp32-tensor kernel computes per tile (single-bf16 products), so results do not change.
    # which bf16 images the main-stream InstanceNorm-backward apply pass (and the block tail) write as side outputs
    # (CWF_APPLY_EMITS="xa,dx", "" = none): each costs the main stream 2 B per element; without it a conversion pass in front of the
    # weight gradient costs the side stream 6 B per element.  Measured (plan mode, one box, volumes/s): none 102.7, dx 104.2, xa,dx 106.2.
    import os as _os
    APPLY_EMITS = tuple(t for t in _os.environ.get("CWF_APPLY_EMITS", "dx" if _os.environ.get("CWF_XA16_FWD", "0") != "0" else "xa,dx").split(",") if t)

    def bf16_operands_ok(self, op, cin, cout, nvox):
        import os
        if os.environ.get("CWF_NO_BF16_OPERANDS"):
            return False
        if op != pk.CONV3_S1 or (_WGRAD_PRECISION or _PRECISION) != "bf16":
            return 0
        if cin == 16 and cout == 16 and nvox >= 32768:
            return 16                                      # wgrad16d_kernel
        if cin >= 32 and cin % 16 == 0 and cout % 32 == 0 and not os.environ.get("CWF_NO_BF16_S1D"):
            return 32                                      # wgrad_s1d_kernel (32-channel output groups)
        return 0

    def bf16_dgrad_ok(self, op, cin, cout, nvox):
        """the data gradient of such a layer can read its incoming gradient as a bf16 image (conv(..., x16=))"""
        import os
        if os.environ.get("CWF_NO_BF16_OPERANDS") or os.environ.get("CWF_NO_BF16_DGRAD"):
            return False
        return op == pk.CONV3_S1 and cin == 16 and cout == 16 and nvox >= 32768 and (_DGRAD_PRECISION or _PRECISION) == "bf16"

    def zero16(self, device):
        z = getattr(self, "_zero16", None)
        if z is None:
            z = self._zero16 = {}
        device = torch.device(device)
        t = z.get(device)
        if t is None:
            t = z[device] = torch.zeros(64, dtype=torch.uint8, device=device)
        return t

    def in_bwd_apply16(self, dy, x, scale, shift, slope, sums, dx_add=None, want_dx16=False, want_xa16=False, need_f32=True):
        """in_bwd_apply with the bf16 side outputs: returns (dx, dx16, xa16).  dx16 = bf16(dx); xa16 = bf16(act(x*scale+shift)).
        need_f32=False: dx is an UNWRITTEN carrier (nothing reads the fp32 gradient; autograd wants an fp32 tensor of x's shape)."""
        dy, dy_ldc = cl(dy)
        x, x_ldc = cl(x)
        n, d, h, w, c = x.shape
        dx = torch.empty((n, d, h, w, c), dtype=_f32, device=x.device)
        dx16 = torch.empty((n, d, h, w, c), dtype=torch.bfloat16, device=x.device) if want_dx16 else None
        xa16 = torch.empty((n, d, h, w, c), dtype=torch.bfloat16, device=x.device) if want_xa16 else None
        assert need_f32 or dx16 is not None
        a_ldc = 0
        if dx_add is not None:
            dx_add, a_ldc = cl(dx_add)
        self._call("cwf_in_bwd_apply_ex", dy.data_ptr(), dy_ldc, x.data_ptr(), x_ldc, scale.data_ptr(), shift.data_ptr(), float(slope),
                   sums.data_ptr(), _p(dx_add), a_ldc, dx.data_ptr() if need_f32 else 0, c, _p(dx16), _p(xa16), n, d * h * w, c, self._stream())
        return dx, dx16, xa16

    def to_bf16(self, x, scale=None, shift=None, slope=1.0, out=None):
        """bf16(act(x*scale+shift)) (scale None: bf16(x)) as a [N,D,H,W,C] bfloat16 tensor -- a stream pass, for operands no producer emitted."""
        x, x_ldc = cl(x)
        n, d, h, w, c = x.shape
        y = out if out is not None else torch.empty((n, d, h, w, c), dtype=torch.bfloat16, device=x.device)
        assert y.dtype == torch.bfloat16 and y.is_contiguous() and tuple(y.shape) == (n, d, h, w, c)
        self._call("cwf_to_bf16", x.data_ptr(), x_ldc, _p(scale), _p(shift), float(slope), y.data_ptr(), n, d * h * w, c, self._stream())
        return y

    # CWF_XA16_FWD=1: bf16(act(IN(x))) of the eligible layers during the FORWARD pass, on the weight-gradient side stream (idle then);
    # default: the apply pass / a conversion in front of the weight gradient makes it during backward
    import os as _os2
    # (measured: 105.9 volumes/s against 108.5 with the apply pass writing it -- the conversions slow the forward's own kernels; off)
    xa16_in_forward = _os2.environ.get("CWF_XA16_FWD", "0") != "0"

    def to_bf16_side(self, x, scale, shift, slope):
        if not self.wgrad_async:
            return self.to_bf16(x, scale, shift, slope)
        side = self.wgrad_stream(x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            y = self.to_bf16(x, scale, shift, slope)
        for t in (x, scale, shift):
            if t is not None:
                t.record_stream(side)
        y.record_stream(torch.cuda.current_stream(x.device))      # (consumed on the side stream; freed by whoever drops the last reference)
        return y

    def conv_grouped(self, x_all, cin, wpks, biases, cout, y_all, x_goff, y_goff, w_refs=None, fwd_op=None, prec=None):
        """G = len(wpks) channel-grouped 3x3x3 stride-1 convs in ONE launch (cwf_conv_mfma_bf16_grouped): group q reads channels
        [q*x_goff, q*x_goff + cin) of x_all and writes channels [q*y_goff, q*y_goff + cout) of y_all (same voxel rows).  Data
        gradients: pass the transposed packed weights (spec.packed(True)) and fwd_op.  Exact-fp32 mode: one launch per group."""
        x_all, x_ldc = cl(x_all)
        y, y_ldc = cl(y_all)
        n, d, h, w, _ = x_all.shape
        G = len(wpks)
        mode = prec or ((_DGRAD_PRECISION or _PRECISION) if (fwd_op is not None) else _PRECISION)
        if mode == "fp32":
            for q in range(G):
                self.conv(pk.CONV3_S1, x_all[..., q * x_goff:q * x_goff + cin], wpks[q], None if biases is None else biases[q], cout,
                          out=y[..., q * y_goff:q * y_goff + cout], fwd_op=fwd_op, prec=mode)
            return y_all
        wp = (ctypes.c_void_p * G)(*[t.data_ptr() for t in wpks])
        bp = (ctypes.c_void_p * G)(*[(0 if (biases is None or b is None) else b.data_ptr()) for b in (biases or [None] * G)])
        self._call("cwf_conv_mfma_bf16_grouped", pk.CONV3_S1, 1 if mode == "bf16x3" else 0, x_all.data_ptr(), x_ldc, x_goff,
                   ctypes.addressof(wp), ctypes.addressof(bp), y.data_ptr(), y_ldc, y_goff, G, n, d, h, w, cin, d, h, w, cout, self._stream())
        return y_all


    # Weight gradients are leaves of the backward pass (only the optimizer reads them) while the data gradients form its
    # critical path.  With `wgrad_async` (opt-in: the caller must call join_wgrad_stream() before reading any .grad -- the
    # Trainer does) the weight-gradient kernels and their slab reductions go to a side stream and overlap the dgrad chain.
    def wgrad_stream(self, device):
        st = self._wg_stream.get(device)
        if st is None:
            # A NON-default priority, for the hardware queue it brings rather than for the scheduling: the HIP runtime multiplexes
            # all streams of one priority over a small pool of hardware queues, and once other streams exist (RCCL creates
            # several at communicator init) a default-priority side stream lands on the MAIN stream's queue -- the weight
            # gradients then run strictly between the main stream's kernels (measured: zero overlap, 87.7 -> 78.5 volumes/s
            # with nothing but a one-rank process group alive).  Each priority has its own queue pool.  "low" is also the
            # right scheduling hint: the data-gradient chain on the main stream is the critical path.
            import os
            pr = os.environ.get("CWF_WGRAD_PRIORITY", "low")
            st = _priority_stream(device, pr)
            self._wg_stream[device] = st
        return st

    def join_wgrad_stream(self):
        for st in self._wg_stream.values():
            torch.cuda.current_stream(st.device).wait_stream(st)
        self._wg_keep.clear()                  # (blocks freed now are reused in main-stream order, which follows the side stream)

    def wgrad(self, op, x, in_scale, in_shift, slope, dy, cout, inv_map, has_bias_map, w_numel, w_ref_shape=None, prec=None, allow_async=False):
        if self.wgrad_async and allow_async:
            # also under hipGraph capture: the side stream joins the capture through wait_stream and becomes a parallel branch of
            # the graph; record_stream defers the operands' blocks until the capture ends (no reuse inside the graph)
            side = self.wgrad_stream(x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                out = self._wgrad_impl(op, x, in_scale, in_shift, slope, dy, cout, inv_map, has_bias_map, w_numel, w_ref_shape, prec)
            for t in (x, dy, in_scale, in_shift):
                if t is not None:
                    t.record_stream(side)
            return out
        return self._wgrad_impl(op, x, in_scale, in_shift, slope, dy, cout, inv_map, has_bias_map, w_numel, w_ref_shape, prec)

    # Gradient-sink form (cwf.optim.GradSink, used by the Trainer): the layer's split-K slabs go to a buffer of its own and the
    # reduction of ALL layers of a backward phase is one launch (wgrad_flush) that writes dW / db straight into the flat gradient
    # buffer -- no per-layer reduce launch, no per-layer gradient tensors, no concatenation before the optimizer.
    def wgrad_release(self):
        """Enqueue the weight-gradient launches held back since `wgrad_defer` was set (see Trainer._fwd_bwd: the decoder's
        full-resolution weight gradients wait until backward reaches the GPU-light middle of the model, where they fill idle CUs
        instead of competing with the decoder's own HBM-bound data gradients)."""
        self.wgrad_defer = False
        held, self._wg_held = self._wg_held, []
        if not held:
            return
        if not self.wgrad_async:
            for args in held:
                self.wgrad_to(*args, allow_async=True)
            return
        # ONE hand-over to the side stream for the whole batch (instead of an event + stream switch + four record_stream calls per
        # launch); the operands stay referenced until the side stream is joined (join_wgrad_stream) instead of being recorded on it
        dev = held[0][2].device
        side = self.wgrad_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for args in held:
                self._wgrad_to_impl(*args)
        self._wg_keep.extend(held)

    def channel_sum_to(self, dy, out, allow_async=False):
        """out[c] = sum over (n, voxels) of dy[..., c] (the bias gradient of a layer whose slabs carry no bias row): a full pass over
        dy that only the optimizer waits for -- on the weight-gradient side stream when that is in use."""
        if self.wgrad_async and allow_async:
            side = self.wgrad_stream(dy.device)
            side.wait_stream(torch.cuda.current_stream(dy.device))
            with torch.cuda.stream(side):
                self.stats_channel_sum(self.in_stats(dy), out)
            dy.record_stream(side)
        else:
            self.stats_channel_sum(self.in_stats(dy), out)
            self._wg_sync_needed = self.wgrad_async

    def dy_scale_ok(self, op, cin, cout, nvox):
        """wgrad_to(..., dy_scale=) is implemented for this layer (the full-resolution kernel of the <= 16 -> 16 layers, bf16 modes)"""
        return op == pk.CONV3_S1 and cin <= 16 and cout == 16 and nvox >= 32768 and (_WGRAD_PRECISION or _PRECISION) != "fp32" and \
            not self.bf16_operands_ok(op, cin, cout, nvox)

    def wgrad_to(self, key, op, x, in_scale, in_shift, slope, dy, cout, inv_map, dw_dst, db_dst, prec=None, x16=None, dy16=None, dy_scale=None,
                 allow_async=False):
        """x16 / dy16: bf16 operand images (see bf16_operands_ok).  Given one of them for an eligible layer, the other is made by a
        conversion pass in front of the launch (on the stream the launch runs on); x / dy are then not read."""
        if self.wgrad_async and allow_async and self.wgrad_defer:
            self._wg_held.append((key, op, x, in_scale, in_shift, slope, dy, cout, inv_map, dw_dst, db_dst, prec, x16, dy16, dy_scale))
            return
        if self.wgrad_async and allow_async:
            side = self.wgrad_stream(x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):
                self._wgrad_to_impl(key, op, x, in_scale, in_shift, slope, dy, cout, inv_map, dw_dst, db_dst, prec, x16, dy16, dy_scale)
            for t in (x, dy, in_scale, in_shift, x16, dy16, dy_scale):
                if t is not None:
                    t.record_stream(side)
        else:
            self._wgrad_to_impl(key, op, x, in_scale, in_shift, slope, dy, cout, inv_map, dw_dst, db_dst, prec, x16, dy16, dy_scale)
            self._wg_sync_needed = self.wgrad_async          # a main-stream producer: the side-stream reduce must wait for it
        if len(self._wg_pending) >= self.flush_every:
            # reduce in instalments: the LAST reduce of backward (after the stem's weight gradient) is exposed before the optimizer
            # step -- it should carry a handful of layers, not a whole phase.  (The Trainer switches this off in hipGraph mode:
            # the descriptor tables are keyed by their rows and must already exist when the capture runs.)
            self.wgrad_flush(x.device)

    def _wgrad_to_impl(self, key, op, x, in_scale, in_shift, slope, dy, cout, inv_map, dw_dst, db_dst, prec, x16=None, dy16=None, dy_scale=None):
        x, x_ldc = cl(x)
        dy, dy_ldc = cl(dy)
        n, di, hi, wi, cin = x.shape
        _, do, ho, wo, _ = dy.shape
        # eligible layers always take the bf16-image kernel; an image nobody handed over is made here, i.e. on the stream this launch
        # runs on (normally the side stream, which has slack: the main stream's kernels are the step's critical path)
        use16 = self.bf16_operands_ok(op, cin, cout, do * ho * wo) if prec is None else 0
        nsplit = self.lib.cwf_wgrad_nsplit(op, n, do, ho, wo, cin, cout)
        slab = self.lib.cwf_wgrad_slab_floats(op, cin, cout)
        if nsplit <= 0 or slab <= 0 or slab != inv_map.numel():
            raise _lib.CwfError("cwf_wgrad plan failed (%d, %d, %d)" % (nsplit, slab, inv_map.numel()))
        pk_ = (key, x.device)                 # persistent, also across graph capture: the descriptor tables below stay valid
        part = self._wg_part.get(pk_)
        if part is None or part.numel() < nsplit * slab:
            part = torch.empty(int(nsplit * slab), dtype=_f32, device=x.device)
            self._wg_part[pk_] = part
        mode = prec or _WGRAD_PRECISION or _PRECISION
        if use16:
            if x16 is None:
                x16 = self.to_bf16(x, in_scale, in_shift, slope)
            if dy16 is None:
                dy16 = self.to_bf16(dy)
            assert x16.dtype == torch.bfloat16 and dy16.dtype == torch.bfloat16 and x16.is_contiguous() and dy16.is_contiguous()
            assert tuple(x16.shape) == (n, di, hi, wi, cin) and tuple(dy16.shape) == (n, do, ho, wo, cout)
            used = ctypes.c_int(0)
            if use16 == 16:
                self._call("cwf_wgrad16_bf16", x16.data_ptr(), dy16.data_ptr(), self.zero16(x.device).data_ptr(), part.data_ptr(),
                           n, di, hi, wi, ctypes.addressof(used), self._stream())
            else:
                self._call("cwf_wgrad_s1_bf16", x16.data_ptr(), dy16.data_ptr(), self.zero16(x.device).data_ptr(), part.data_ptr(),
                           n, di, hi, wi, cin, cout, ctypes.addressof(used), self._stream())
            nsplit = used.value
        elif dy_scale is not None:
            used = ctypes.c_int(0)
            self._call("cwf_wgrad_mfma_bf16_dys", op, 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, _p(in_scale), _p(in_shift),
                       float(slope), dy.data_ptr(), dy_ldc, dy_scale.data_ptr(), part.data_ptr(), n, di, hi, wi, cin, do, ho, wo, cout,
                       ctypes.addressof(used), self._stream())
            nsplit = used.value
        elif mode == "fp32":
            self._call("cwf_wgrad_mfma", op, x.data_ptr(), x_ldc, _p(in_scale), _p(in_shift), float(slope),
                       dy.data_ptr(), dy_ldc, part.data_ptr(), n, di, hi, wi, cin, do, ho, wo, cout, self._stream())
        else:
            used = ctypes.c_int(0)
            self._call("cwf_wgrad_mfma_bf16", op, 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, _p(in_scale), _p(in_shift),
                       float(slope), dy.data_ptr(), dy_ldc, part.data_ptr(), n, di, hi, wi, cin, do, ho, wo, cout,
                       ctypes.addressof(used), self._stream())
            nsplit = used.value
        self._wg_pending.append((part.data_ptr(), inv_map.data_ptr(), dw_dst.data_ptr(), _p(db_dst), int(slab), int(nsplit)))


    def wgrad_to_grouped(self, keys, op, xs, dys, cout, inv_maps, dw_dsts, db_dsts, prec=None, allow_async=False):
        """wgrad_to for G same-shape 3x3x3 stride-1 layers without prologue (channel-group views xs[q] / dys[q] of shared buffers):
        ONE slab launch (cwf_wgrad_mfma_bf16_grouped); each layer keeps its own slab buffer and row in the batched reduce."""
        mode = prec or _WGRAD_PRECISION or _PRECISION
        if mode == "fp32" or (self.wgrad_async and allow_async and self.wgrad_defer):
            for q in range(len(keys)):
                self.wgrad_to(keys[q], op, xs[q], None, None, 1.0, dys[q], cout, inv_maps[q], dw_dsts[q], db_dsts[q], prec=prec, allow_async=allow_async)
            return
        if self.wgrad_async and allow_async:
            dev = xs[0].device
            side = self.wgrad_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._wgrad_to_grouped_impl(keys, op, xs, dys, cout, inv_maps, dw_dsts, db_dsts, mode)
            for t in list(xs) + list(dys):
                t.record_stream(side)
        else:
            self._wgrad_to_grouped_impl(keys, op, xs, dys, cout, inv_maps, dw_dsts, db_dsts, mode)
            self._wg_sync_needed = self.wgrad_async

    def _wgrad_to_grouped_impl(self, keys, op, xs, dys, cout, inv_maps, dw_dsts, db_dsts, mode):
        G = len(keys)
        x0, x_ldc = cl(xs[0])
        d0, dy_ldc = cl(dys[0])
        n, di, hi, wi, cin = x0.shape
        _, do, ho, wo, _ = d0.shape
        nsplit = self.lib.cwf_wgrad_nsplit(op, n, do, ho, wo, cin, cout)
        slab = self.lib.cwf_wgrad_slab_floats(op, cin, cout)
        if nsplit <= 0 or slab <= 0 or any(slab != m.numel() for m in inv_maps):
            raise _lib.CwfError("cwf_wgrad plan failed (%d, %d)" % (nsplit, slab))
        parts = []
        for q in range(G):
            assert xs[q].stride() == xs[0].stride() and dys[q].stride() == dys[0].stride() and xs[q].shape == xs[0].shape
            pk_ = (keys[q], x0.device)
            part = self._wg_part.get(pk_)
            if part is None or part.numel() < nsplit * slab:
                part = torch.empty(int(nsplit * slab), dtype=_f32, device=x0.device)
                self._wg_part[pk_] = part
            parts.append(part)
        xp = (ctypes.c_void_p * G)(*[t.data_ptr() for t in xs])
        dp = (ctypes.c_void_p * G)(*[t.data_ptr() for t in dys])
        pp = (ctypes.c_void_p * G)(*[t.data_ptr() for t in parts])
        used = ctypes.c_int(0)
        self._call("cwf_wgrad_mfma_bf16_grouped", op, 1 if mode == "bf16x3" else 0, ctypes.addressof(xp), x_ldc, ctypes.addressof(dp), dy_ldc,
                   ctypes.addressof(pp), G, n, di, hi, wi, cin, do, ho, wo, cout, ctypes.addressof(used), self._stream())
        for q in range(G):
            self._wg_pending.append((parts[q].data_ptr(), inv_maps[q].data_ptr(), dw_dsts[q].data_ptr(), _p(db_dsts[q]), int(slab), int(used.value)))

    def wgrad_flush(self, device=None):
        """One batched reduce for every layer queued by wgrad_to since the last flush (on the weight-gradient side stream when
        that is in use: it follows the queued kernels in stream order)."""
        rows = self._wg_pending
        if not rows:
            return
        self._wg_pending = []
        device = device or torch.device("cuda", _current_device())
        key = (device, tuple(rows))
        table = self._wg_tables.get(key)
        if table is None:                       # static across steps (persistent buffers, static shapes): built once
            import struct
            raw = b"".join(struct.pack("<QQQQqii", r[0], r[1], r[2], r[3], r[4], r[5], 0) for r in rows)
            host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            if torch.cuda.is_current_stream_capturing():
                # (only if no eager step preceded the capture) a pinned source keeps the copy capturable; it must outlive the graph
                host = host.pin_memory()
                self._wg_tables[("pinned",) + key] = host
                table = torch.empty_like(host, device=device)
                table.copy_(host, non_blocking=True)
            else:
                table = host.to(device)
            self._wg_tables[key] = table
        if self.wgrad_async:
            side = self.wgrad_stream(device)
            if getattr(self, "_wg_sync_needed", False):
                side.wait_stream(torch.cuda.current_stream(device))
                self._wg_sync_needed = False
            with torch.cuda.stream(side):
                self._call("cwf_wgrad_reduce_batched", table.data_ptr(), len(rows), self._stream())
        else:
            self._call("cwf_wgrad_reduce_batched", table.data_ptr(), len(rows), self._stream())

    def _wgrad_impl(self, op, x, in_scale, in_shift, slope, dy, cout, inv_map, has_bias_map, w_numel, w_ref_shape=None, prec=None):
        """returns (dW flat [w_numel], db [cout] or None)"""
        x, x_ldc = cl(x)
        dy, dy_ldc = cl(dy)
        n, di, hi, wi, cin = x.shape
        _, do, ho, wo, _ = dy.shape
        nsplit = self.lib.cwf_wgrad_nsplit(op, n, do, ho, wo, cin, cout)
        slab = self.lib.cwf_wgrad_slab_floats(op, cin, cout)
        if nsplit <= 0 or slab <= 0 or slab != inv_map.numel():
            raise _lib.CwfError("cwf_wgrad plan failed (%d, %d, %d)" % (nsplit, slab, inv_map.numel()))
        part = self.workspace("wgrad", nsplit * slab, x.device)
        mode = prec or _WGRAD_PRECISION or _PRECISION
        if mode == "fp32":
            self._call("cwf_wgrad_mfma", op, x.data_ptr(), x_ldc, _p(in_scale), _p(in_shift), float(slope),
                       dy.data_ptr(), dy_ldc, part.data_ptr(), n, di, hi, wi, cin, do, ho, wo, cout, self._stream())
        else:
            used = ctypes.c_int(0)
            self._call("cwf_wgrad_mfma_bf16", op, 1 if mode == "bf16x3" else 0, x.data_ptr(), x_ldc, _p(in_scale), _p(in_shift),
                       float(slope), dy.data_ptr(), dy_ldc, part.data_ptr(), n, di, hi, wi, cin, do, ho, wo, cout,
                       ctypes.addressof(used), self._stream())
            nsplit = used.value
        dw = torch.empty(w_numel, dtype=_f32, device=x.device)
        db = torch.empty(cout, dtype=_f32, device=x.device) if has_bias_map else None
        self._call("cwf_wgrad_reduce", part.data_ptr(), nsplit, slab, inv_map.data_ptr(), dw.data_ptr(), _p(db), self._stream())
        return dw, db

    def gather_batched(self, table, nlayers, max_n, split_bf16=False):
        self._call("cwf_gather_split_bf16" if split_bf16 else "cwf_gather_batched", table.data_ptr(), nlayers, max_n, self._stream())

    # ------------------------------------------------------------------ K3
    def new_stats(self, n, c, device):
        return self._zeros_f64((n, c, 2), device)

    def in_finalize(self, stats, nvox, eps=1e-5):
        n, c, _ = stats.shape
        scale = torch.empty((n, c), dtype=_f32, device=stats.device)
        shift = torch.empty((n, c), dtype=_f32, device=stats.device)
        self._call("cwf_in_finalize", stats.data_ptr(), scale.data_ptr(), shift.data_ptr(), n * c, nvox, eps, self._stream())
        return scale, shift

    def in_stats(self, x):
        x, ldc = cl(x)
        n, d, h, w, c = x.shape
        stats = self.new_stats(n, c, x.device)
        self._call("cwf_in_stats", x.data_ptr(), ldc, stats.data_ptr(), n, d * h * w, c, self._stream())
        return stats

    def norm_act_add(self, x, scale, shift, slope, residual=None, want16=False):
        """want16: also returns bf16(y) (the operand image of a consuming layer's weight gradient): (y, y16)"""
        x, ldc = cl(x)
        n, d, h, w, c = x.shape
        y = torch.empty((n, d, h, w, c), dtype=_f32, device=x.device)
        y16 = torch.empty((n, d, h, w, c), dtype=torch.bfloat16, device=x.device) if want16 else None
        r_ldc = 0
        if residual is not None:
            residual, r_ldc = cl(residual)
        self._call("cwf_norm_act_add_ex", x.data_ptr(), ldc, scale.data_ptr(), shift.data_ptr(), float(slope), _p(residual), r_ldc,
                   y.data_ptr(), c, _p(y16), n, d * h * w, c, self._stream())
        return (y, y16) if want16 else y

    def in_bwd(self, dy, x, scale, shift, slope, dx_add=None):
        """dx for y = act(IN(x)) given dy = dL/dy (full InstanceNorm backward, statistics included)."""
        dy, dy_ldc = cl(dy)
        x, x_ldc = cl(x)
        n, d, h, w, c = x.shape
        v = d * h * w
        sums = self.new_stats(n, c, x.device)
        self._call("cwf_in_bwd_stats", dy.data_ptr(), dy_ldc, x.data_ptr(), x_ldc, scale.data_ptr(), shift.data_ptr(), float(slope),
                   sums.data_ptr(), n, v, c, self._stream())
        dx = torch.empty((n, d, h, w, c), dtype=_f32, device=x.device)
        a_ldc = 0
        if dx_add is not None:
            dx_add, a_ldc = cl(dx_add)
        self._call("cwf_in_bwd_apply", dy.data_ptr(), dy_ldc, x.data_ptr(), x_ldc, scale.data_ptr(), shift.data_ptr(), float(slope),
                   sums.data_ptr(), _p(dx_add), a_ldc, dx.data_ptr(), c, n, v, c, self._stream())
        return dx

    # ------------------------------------------------------------------ K6/K7
    def gemm(self, a, sa, b, sb, c, sc, m, n, k, zb=1, zh=1, bias=None, residual=None, sr=(0, 0, 0), alpha=1.0, act=0,
             accumulate=False, a_off=0, b_off=0, c_off=0, r_off=0):
        """sa = (m, k, zb, zh) element strides of A; sb = (k, n, zb, zh); sc = (m, zb, zh); offsets in elements."""
        self._call("cwf_gemm", a.data_ptr() + 4 * a_off, sa[0], sa[1], sa[2], sa[3],
                   b.data_ptr() + 4 * b_off, sb[0], sb[1], sb[2], sb[3],
                   c.data_ptr() + 4 * c_off, sc[0], sc[1], sc[2],
                   _p(bias), (residual.data_ptr() + 4 * r_off) if residual is not None else 0, sr[0], sr[1], sr[2],
                   m, n, k, zb, zh, float(alpha), act, int(accumulate), self._stream())
        return c

    def layernorm_fwd(self, x, gamma, beta, eps=1e-5):
        rows, e = x.numel() // x.shape[-1], x.shape[-1]
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=_f32, device=x.device)
        rstd = torch.empty(rows, dtype=_f32, device=x.device)
        self._call("cwf_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   rows, e, eps, self._stream())
        return y, mean, rstd

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dgamma, dbeta):
        rows, e = x.numel() // x.shape[-1], x.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        self._call("cwf_layernorm_bwd", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, e, 0, self._stream())
        return dx

    def softmax_rows_(self, s):
        cols = s.shape[-1]
        self._call("cwf_softmax_rows", s.data_ptr(), s.numel() // cols, cols, cols, self._stream())
        return s

    def softmax_rows_bwd_(self, p, dp):
        cols = p.shape[-1]
        self._call("cwf_softmax_rows_bwd", p.data_ptr(), dp.data_ptr(), p.numel() // cols, cols, cols, self._stream())
        return dp

    def gelu_bwd(self, x, dy):
        dx = torch.empty_like(x)
        self._call("cwf_gelu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), self._stream())
        return dx

    def colsum(self, x2d):
        rows, cols = x2d.shape
        out = torch.empty(cols, dtype=_f32, device=x2d.device)
        self._call("cwf_colsum", x2d.data_ptr(), rows, cols, x2d.stride(0), out.data_ptr(), 0, self._stream())
        return out

    # ------------------------------------------------------------------ K4/K5
    def window_to_tokens(self, x, patch):
        x, ldc = cl(x)
        b, d, h, w, c = x.shape
        p0, p1, p2 = patch
        tok = torch.empty((b, (d // p0) * (h // p1) * (w // p2), c * p0 * p1 * p2), dtype=_f32, device=x.device)
        self._call("cwf_window_to_tokens", x.data_ptr(), ldc, tok.data_ptr(), b, d, h, w, c, p0, p1, p2, self._stream())
        return tok

    def tokens_to_window(self, tok, size, channels, patch):
        b = tok.shape[0]
        d, h, w = size
        p0, p1, p2 = patch
        tok = tok.contiguous()
        x = torch.empty((b, d, h, w, channels), dtype=_f32, device=tok.device)
        self._call("cwf_tokens_to_window", tok.data_ptr(), x.data_ptr(), channels, b, d, h, w, channels, p0, p1, p2, 0, self._stream())
        return x

    def window_to_tokens_g(self, x, groups, patch):
        """x [B,D,H,W,groups*C] -> tok [groups,B,T,C*p0*p1*p2] (one launch for the three sub-regions)"""
        x, ldc = cl(x)
        b, d, h, w, ct = x.shape
        c = ct // groups
        p0, p1, p2 = patch
        tok = torch.empty((groups, b, (d // p0) * (h // p1) * (w // p2), c * p0 * p1 * p2), dtype=_f32, device=x.device)
        self._call("cwf_window_to_tokens_g", x.data_ptr(), ldc, tok.data_ptr(), groups, b, d, h, w, c, p0, p1, p2, self._stream())
        return tok

    def tokens_to_window_g(self, tok, size, channels, patch):
        """tok [groups,B,T,E] -> x [B,D,H,W,groups*channels]"""
        g, b = tok.shape[0], tok.shape[1]
        d, h, w = size
        p0, p1, p2 = patch
        tok = tok.contiguous()
        x = torch.empty((b, d, h, w, g * channels), dtype=_f32, device=tok.device)
        self._call("cwf_tokens_to_window_g", tok.data_ptr(), x.data_ptr(), g * channels, g, b, d, h, w, channels, p0, p1, p2, self._stream())
        return x

    def cat3_channels(self, parts, shape, device):
        """[N,D,H,W,C] x 3 (None = zeros) -> [N,D,H,W,3C]"""
        n, d, h, w, c = shape
        parts = [None if t is None else t.contiguous() for t in parts]
        y = torch.empty((n, d, h, w, 3 * c), dtype=_f32, device=device)
        self._call("cwf_cat3_channels", _p(parts[0]), _p(parts[1]), _p(parts[2]), y.data_ptr(), n * d * h * w, c, self._stream())
        return y

    def token_scores(self, feats, query):
        """feats [B,T,E]; query [1,1,E] (shared) or [B,1,E]."""
        b, t, e = feats.shape
        score = torch.empty((b, t), dtype=_f32, device=feats.device)
        qbs = 0 if query.shape[0] == 1 else e
        self._call("cwf_token_scores", feats.data_ptr(), query.data_ptr(), qbs, score.data_ptr(), b, t, e, self._stream())
        return score

    def topk(self, score, k):
        b, t = score.shape
        idx = torch.empty((b, k), dtype=torch.int32, device=score.device)
        self._call("cwf_topk", score.data_ptr(), idx.data_ptr(), b, t, k, self._stream())
        return idx

    def gather_tokens(self, feats, index, head, keep=None, pe_odd=1.0):
        b, t, e = feats.shape
        k = index.shape[1]
        seq = torch.empty((b, k + 1, e), dtype=_f32, device=feats.device)
        hbs = 0 if head.shape[0] == 1 else e
        self._call("cwf_gather_tokens", feats.data_ptr(), index.data_ptr(), head.data_ptr(), hbs, _p(keep), float(pe_odd),
                   seq.data_ptr(), b, t, k, e, self._stream())
        return seq

    def gather_tokens_bwd(self, dseq, index, keep, dfeats, dhead):
        """dfeats [B,T,E] accumulated in place (may be None); dhead [1 or B,1,E] accumulated in place (may be None)."""
        b, k1, e = dseq.shape
        dseq = dseq.contiguous()
        t = dfeats.shape[1] if dfeats is not None else 0
        dhbs = 0 if (dhead is None or dhead.shape[0] == 1) else e
        self._call("cwf_gather_tokens_bwd", dseq.data_ptr(), index.data_ptr(), _p(keep), _p(dfeats), _p(dhead), dhbs,
                   b, t, k1 - 1, e, self._stream())

    def scatter_rows(self, feats, index, rows, gate=None):
        """rows: [B,k,E] view (row stride / batch stride taken from the tensor); gate [B,1,E] view or None."""
        b, t, e = feats.shape
        k = index.shape[1]
        assert rows.stride(2) == 1
        out = torch.empty((b, t, e), dtype=_f32, device=feats.device)
        gbs = gate.stride(0) if gate is not None else 0
        self._call("cwf_scatter_rows", feats.data_ptr(), index.data_ptr(), rows.data_ptr(), rows.stride(1), rows.stride(0),
                   _p(gate), gbs, out.data_ptr(), b, t, k, e, self._stream())
        return out

    def scatter_rows_bwd(self, dout, index, scat, gate, k, need_feats=True, need_rows=True):
        """returns (dfeats [B,T,E] or None, drows [B,k,E] or None, dgate [B,1,E] or None)"""
        b, t, e = dout.shape
        dout = dout.contiguous()
        dfeats = torch.empty((b, t, e), dtype=_f32, device=dout.device) if need_feats else None
        drows = torch.empty((b, k, e), dtype=_f32, device=dout.device) if need_rows else None
        dgate = torch.zeros((b, 1, e), dtype=_f32, device=dout.device) if gate is not None else None
        gbs = gate.stride(0) if gate is not None else 0
        self._call("cwf_scatter_rows_bwd", dout.data_ptr(), index.data_ptr(), _p(scat), _p(gate), gbs,
                   _p(dfeats), 0, _p(drows), e, k * e, _p(dgate), e, b, t, k, e, self._stream())
        return dfeats, drows, dgate

    # ------------------------------------------------------------------ K6/K7, round-2 fused forms (one coupler block = 4 launches)
    def _gemm_ex(self, **kw):
        g = _lib.GemmArgs()
        for k, v in kw.items():
            setattr(g, k, v)
        self._call("cwf_gemm_ex", ctypes.addressof(g), self._stream())

    def _drop_kw(self, prefix, drop, numel, dev):
        if not drop:
            return {}
        off, p, p2 = drop
        return {prefix + "_off": off, prefix + "_n": numel, prefix + "_p": float(p), prefix + "_p2": float(p2), "rng": self.rng(dev).data_ptr()}

    def linear_fwd(self, x, w, bias, out, x2=None, split_n=0, act=0, pre=None, drop=None, residual=None):
        """out = drop(act(x' w^T + bias)) + residual ; x' = x for output columns < split_n, x2 beyond (2-D row-major views)."""
        m, k = x.shape
        grouped = isinstance(w, (list, tuple))          # G weight sets: rows of x / out are G stacked problems (z = group)
        G = len(w) if grouped else 1
        w0 = w[0] if grouped else w
        n = w0.shape[0]
        assert x.stride(1) == 1 and w0.stride(1) == 1 and out.stride(1) == 1 and out.shape == (m, n) and m % G == 0
        mg = m // G
        kw = dict(A=x.data_ptr(), sa_m=x.stride(0), sa_k=1, sa_zb=mg * x.stride(0), B=w0.data_ptr(), sb_k=1, sb_n=w0.stride(0),
                  C=out.data_ptr(), sc_m=out.stride(0), sc_zb=mg * out.stride(0), bias=_p(bias) if not grouped else 0,
                  M=mg, N=n, K=k, ZB=G, ZH=1, alpha=1.0, act=act)
        if grouped:
            assert all(t.stride() == w0.stride() for t in w)
            kw["B_tab"] = _tab(w)
            if bias is not None:
                kw["bias_tab"] = _tab(bias)
        if x2 is not None:
            assert x2.stride() == x.stride() and x2.shape == x.shape
            kw.update(A2=x2.data_ptr(), split_n=split_n)
        if pre is not None:
            assert pre.stride() == out.stride()
            kw["C2"] = pre.data_ptr()
        if residual is not None:
            assert residual.stride(1) == 1
            kw.update(residual=residual.data_ptr(), sr_m=residual.stride(0), sr_zb=mg * residual.stride(0))
        kw.update(self._drop_kw("c_drop", drop, m * n, x.device))
        if drop:
            assert out.stride(0) == n           # the mask index is the element offset inside out
        self._gemm_ex(**kw)
        return out

    def linear_dgrad(self, dy, w, drop=None, out=None):
        """dx = (dy * keep) w ; dy [M,N] (row stride from the view), w [N,K] view."""
        m, n = dy.shape
        grouped = isinstance(w, (list, tuple))
        G = len(w) if grouped else 1
        w0 = w[0] if grouped else w
        k = w0.shape[1]
        assert dy.stride(1) == 1 and w0.stride(1) == 1 and m % G == 0
        mg = m // G
        dx = out if out is not None else torch.empty((m, k), dtype=_f32, device=dy.device)
        kw = dict(A=dy.data_ptr(), sa_m=dy.stride(0), sa_k=1, sa_zb=mg * dy.stride(0), B=w0.data_ptr(), sb_k=w0.stride(0), sb_n=1,
                  C=dx.data_ptr(), sc_m=dx.stride(0), sc_zb=mg * dx.stride(0), M=mg, N=k, K=n, ZB=G, ZH=1, alpha=1.0)
        if grouped:
            assert all(t.stride() == w0.stride() for t in w)
            kw["B_tab"] = _tab(w)
        kw.update(self._drop_kw("a_drop", drop, m * n, dy.device))
        if drop:
            assert dy.stride(0) == n
        self._gemm_ex(**kw)
        return dx

    def linear_wgrad(self, dy, x, dw, dbias=None, x2=None, split_m=0, accumulate=False, drop=None):
        """dw (+)= (dy * keep)^T x' ; dbias (+)= column sums of dy * keep ; x' = x for dw rows < split_m, x2 beyond."""
        m, n = dy.shape
        k = x.shape[1]
        grouped = isinstance(dw, (list, tuple))
        G = len(dw) if grouped else 1
        dw0 = dw[0] if grouped else dw
        assert dy.stride(1) == 1 and x.stride(1) == 1 and dw0.shape == (n, k) and dw0.is_contiguous() and m % G == 0
        mg = m // G
        kw = dict(A=dy.data_ptr(), sa_m=1, sa_k=dy.stride(0), sa_zb=mg * dy.stride(0), B=x.data_ptr(), sb_k=x.stride(0), sb_n=1,
                  sb_zb=mg * x.stride(0), C=dw0.data_ptr(), sc_m=k, M=n, N=k, K=mg, ZB=G, ZH=1, alpha=1.0, accumulate=int(accumulate))
        if grouped:
            assert all(t.shape == dw0.shape and t.is_contiguous() for t in dw)
            kw["C_tab"] = _tab(dw)
        if x2 is not None:
            assert x2.stride() == x.stride()
            kw.update(B2=x2.data_ptr(), split_m=split_m)
        if dbias is not None:
            if grouped:
                kw.update(rowsum_tab=_tab(dbias), rowsum_acc=int(accumulate))
            else:
                kw.update(rowsum=dbias.data_ptr(), rowsum_acc=int(accumulate))
        kw.update(self._drop_kw("a_drop", drop, m * n, dy.device))
        if drop:
            assert dy.stride(0) == n
        self._gemm_ex(**kw)

    def ln_pair_fwd(self, x, x2, perm_T, g1, b1, g2, b2, eps=1e-5):
        """g1 ... b2: tensors, or lists of G tensors (rows = G stacked problems, one LayerNorm parameter set each)"""
        rows, e = x.shape
        ya = torch.empty_like(x)
        yb = torch.empty_like(x) if x2 is not None else None
        stats = torch.empty((2 if x2 is not None else 1, rows, 2), dtype=_f32, device=x.device)
        P, G = _ln_params(g1=g1, b1=b1, g2=g2, b2=b2)
        self._call("cwf_ln_pair_fwd_g", x.data_ptr(), _p(x2), perm_T, ctypes.addressof(P), G, ya.data_ptr(), _p(yb),
                   stats.data_ptr(), rows, e, eps, self._stream())
        return ya, yb, stats

    def ln_pair_bwd(self, dy, da, db, x, x2, perm_T, g1, g2, stats, dg1, db1, dg2, db2, accumulate, want_dx2):
        rows, e = x.shape
        dx = torch.empty_like(x)
        dx2 = torch.empty_like(x) if want_dx2 else None
        P, G = _ln_params(g1=g1, g2=g2, dg1=dg1, db1=db1, dg2=dg2, db2=db2)
        self._call("cwf_ln_pair_bwd_g", _p(dy), da.data_ptr(), _p(db), x.data_ptr(), _p(x2), perm_T, ctypes.addressof(P), G, stats.data_ptr(),
                   dx.data_ptr(), _p(dx2), rows, e, int(accumulate), self._stream())
        return dx, dx2

    def attn_fwd(self, qkv, z, t, heads, drop=None):
        e = qkv.shape[1] // 3
        o = torch.empty((z * t, e), dtype=_f32, device=qkv.device)
        off, p = (drop[0], drop[1]) if drop else (0, 0.0)
        self._call("cwf_attn_fwd", qkv.data_ptr(), qkv.stride(0), o.data_ptr(), e, z, t, e, heads, float((e // heads) ** -0.5),
                   self.rng(qkv.device).data_ptr(), off, float(p), self._stream())
        return o

    def attn_bwd(self, qkv, d_o, z, t, heads, drop=None):
        e = qkv.shape[1] // 3
        assert d_o.is_contiguous()
        dqkv = torch.empty_like(qkv)
        off, p = (drop[0], drop[1]) if drop else (0, 0.0)
        self._call("cwf_attn_bwd", qkv.data_ptr(), qkv.stride(0), d_o.data_ptr(), e, dqkv.data_ptr(), z, t, e, heads,
                   float((e // heads) ** -0.5), self.rng(qkv.device).data_ptr(), off, float(p), self._stream())
        return dqkv

    def gelu_bwd_drop(self, z, dh, drop=None):
        dz = torch.empty_like(z)
        off, p = (drop[0], drop[1]) if drop else (0, 0.0)
        self._call("cwf_gelu_bwd_drop", z.data_ptr(), dh.data_ptr(), dz.data_ptr(), z.numel(), self.rng(z.device).data_ptr(), off, float(p), self._stream())
        return dz

    # ------------------------------------------------------------------ K4/K5, round-2 forms
    def token_scores2(self, feats, q1, q2=None):
        """q1 / q2: [1|B,1,E] tensors, or lists of G shared queries (sample b belongs to group b // (B/G))"""
        b, t, e = feats.shape
        s1 = torch.empty((b, t), dtype=_f32, device=feats.device)
        s2 = torch.empty((b, t), dtype=_f32, device=feats.device) if q2 is not None else None
        if isinstance(q1, (list, tuple)):
            G = len(q1)
            t1 = (ctypes.c_void_p * G)(*[q.data_ptr() for q in q1])
            t2 = (ctypes.c_void_p * G)(*[q.data_ptr() for q in q2]) if q2 is not None else None
            self._call("cwf_token_scores2_g", feats.data_ptr(), ctypes.addressof(t1), ctypes.addressof(t2) if t2 is not None else 0, G, b // G,
                       s1.data_ptr(), _p(s2), b, t, e, self._stream())
            return s1, s2
        self._call("cwf_token_scores2", feats.data_ptr(), q1.data_ptr(), 0 if q1.shape[0] == 1 else e, _p(q2),
                   0 if (q2 is None or q2.shape[0] == 1) else e, s1.data_ptr(), _p(s2), b, t, e, self._stream())
        return s1, s2

    def topk_inv(self, s0, s1, k):
        """-> (index0, inv0, index1, inv1); the second pair is None when s1 is None."""
        b, t = s0.shape
        i32 = torch.int32
        idx0 = torch.empty((b, k), dtype=i32, device=s0.device); inv0 = torch.empty((b, t), dtype=i32, device=s0.device)
        idx1 = inv1 = None
        if s1 is not None:
            idx1 = torch.empty((b, k), dtype=i32, device=s0.device); inv1 = torch.empty((b, t), dtype=i32, device=s0.device)
        self._call("cwf_topk_inv", s0.data_ptr(), idx0.data_ptr(), inv0.data_ptr(), _p(s1), _p(idx1), _p(inv1), b, t, k, self._stream())
        return idx0, inv0, idx1, inv1

    def index_inv(self, index, t):
        b, k = index.shape
        index = index.to(torch.int32).contiguous()
        inv = torch.empty((b, t), dtype=torch.int32, device=index.device)
        self._call("cwf_index_inv", index.data_ptr(), inv.data_ptr(), b, t, k, self._stream())
        return index, inv

    def gather_multi(self, jobs, k, e, p=0.0, pe_odd=1.0):
        """jobs: list of (feats [B,T,E], index [B,k], head [1|B,1,E], out view [B,k+1,E] with contiguous rows, drop_off)."""
        arr = (_lib.GatherJob * len(jobs))()
        b = jobs[0][0].shape[0]
        for i, (feats, index, head, out, off) in enumerate(jobs):
            assert out.stride(2) == 1 and out.stride(1) == e and feats.is_contiguous()
            if isinstance(head, (list, tuple)):              # one shared head per group of b // len(head) samples
                hg = (ctypes.c_void_p * 4)(*([h.data_ptr() for h in head] + [0] * (4 - len(head))))
                arr[i] = _lib.GatherJob(feats.data_ptr(), index.data_ptr(), 0, out.data_ptr(), 0, out.stride(0), feats.shape[1], off, hg, b // len(head))
            else:
                arr[i] = _lib.GatherJob(feats.data_ptr(), index.data_ptr(), head.data_ptr(), out.data_ptr(), 0 if head.shape[0] == 1 else e,
                                        out.stride(0), feats.shape[1], off, (ctypes.c_void_p * 4)(), 0)
        self._call("cwf_gather_multi", ctypes.addressof(arr), len(jobs), b, k, e, float(pe_odd), self.rng(jobs[0][0].device).data_ptr(), float(p), self._stream())

    def scatter_inv(self, feats, inv, rows, gate, want_gated=True, want_scat=False):
        """rows [B,k,E] / gate [B,1,E] views with unit inner stride -> (gated, scat)"""
        b, t, e = feats.shape
        gated = torch.empty_like(feats) if want_gated else None
        scat = torch.empty_like(feats) if want_scat else None
        self._call("cwf_scatter_inv", feats.data_ptr(), inv.data_ptr(), rows.data_ptr(), rows.stride(1), rows.stride(0), _p(gate),
                   gate.stride(0) if gate is not None else 0, _p(gated), _p(scat), b, t, e, self._stream())
        return gated, scat

    def scatter_bwd(self, dgated, dscat, feats, inv, index, rows, gate, dgate_extra, drows, dgate):
        """writes drows [B,k,E] view and dgate [B,1,E] view (both inside the coupler's output gradient)"""
        b, t, e = feats.shape
        k = index.shape[1]
        self._call("cwf_scatter_bwd", _p(dgated), _p(dscat), feats.data_ptr(), inv.data_ptr(), index.data_ptr(), rows.data_ptr(), rows.stride(1),
                   rows.stride(0), _p(gate), gate.stride(0) if gate is not None else 0, _p(dgate_extra),
                   dgate_extra.stride(0) if dgate_extra is not None else 0, drows.data_ptr(), drows.stride(1), drows.stride(0),
                   dgate.data_ptr(), dgate.stride(0), b, t, k, e, self._stream())

    def token_grad(self, dgated, dscat, gate, inv_p, inv_q, dseq_p, dseq_q, k, p=0.0, off_p=0, off_q=0):
        b, t = inv_p.shape
        e = dseq_p.shape[2]
        dfeats = torch.empty((b, t, e), dtype=_f32, device=dseq_p.device)
        self._call("cwf_token_grad", _p(dgated), _p(dscat), _p(gate), gate.stride(0) if gate is not None else 0, inv_p.data_ptr(), _p(inv_q),
                   dseq_p.data_ptr(), dseq_p.stride(0), _p(dseq_q), dseq_q.stride(0) if dseq_q is not None else 0,
                   self.rng(dseq_p.device).data_ptr(), off_p, off_q, float(p), dfeats.data_ptr(), b, t, k, e, self._stream())
        return dfeats

    def head_grad(self, a1, c1, a2, c2, out1=None, out2=None):
        """out1 = sum_b (a1[b] + c1[b]), out2 = sum_b (a2[b] + c2[b]); inputs are [B,E] row views with a common batch stride"""
        b, e = a1.shape
        bs = a1.stride(0)
        assert c1.stride(0) == bs and a2.stride(0) == bs and c2.stride(0) == bs
        if isinstance(out1, (list, tuple)):                  # G groups of b // G samples, one output pair per group
            G = len(out1)
            t1 = (ctypes.c_void_p * G)(*[t.data_ptr() for t in out1]); t2 = (ctypes.c_void_p * G)(*[t.data_ptr() for t in out2])
            self._call("cwf_head_grad_g", a1.data_ptr(), c1.data_ptr(), a2.data_ptr(), c2.data_ptr(), bs, ctypes.addressof(t1), ctypes.addressof(t2),
                       G, b // G, e, self._stream())
            return out1, out2
        o1 = out1 if out1 is not None else torch.empty((1, 1, e), dtype=_f32, device=a1.device)
        o2 = out2 if out2 is not None else torch.empty((1, 1, e), dtype=_f32, device=a1.device)
        self._call("cwf_head_grad", a1.data_ptr(), c1.data_ptr(), a2.data_ptr(), c2.data_ptr(), bs, o1.data_ptr(), o2.data_ptr(), b, e, self._stream())
        return o1, o2

    def sum_groups3(self, x):
        """x [3, ...] contiguous -> x[0] + x[1] + x[2]"""
        assert x.shape[0] == 3 and x.is_contiguous()
        y = torch.empty(x.shape[1:], dtype=_f32, device=x.device)
        n = y.numel()
        self._call("cwf_add3", x.data_ptr(), x.data_ptr() + 4 * n, x.data_ptr() + 8 * n, y.data_ptr(), n, self._stream())
        return y

    def bcast_groups3(self, d):
        """d [...] -> [3, ...] (three copies): the adjoint of sum_groups3 as ONE launch"""
        d = d.contiguous()
        y = torch.empty((3,) + tuple(d.shape), dtype=_f32, device=d.device)
        self._call("cwf_bcast3", d.data_ptr(), y.data_ptr(), d.numel(), self._stream())
        return y

    def stats_channel_sum(self, stats, out):
        """out[c] = sum_n stats[n][c][0]  (bias gradient of a transposed conv from the per-(n,c) sums of dy)"""
        n, c, _ = stats.shape
        self._call("cwf_stats_channel_sum", stats.data_ptr(), out.data_ptr(), n, c, self._stream())
        return out

    def add3(self, a, b, c):
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        y = torch.empty_like(a)
        self._call("cwf_add3", a.data_ptr(), b.data_ptr(), c.data_ptr(), y.data_ptr(), a.numel(), self._stream())
        return y

    # ------------------------------------------------------------------ K8/K10
    def upsample_softmax(self, logit, c, scale):
        """logit [N,d,h,w,>=c] (channel stride from the tensor) -> prob [N,d*s,h*s,w*s,c]"""
        n, d, h, w, _ = logit.shape
        prob = torch.empty((n, d * scale, h * scale, w * scale, c), dtype=_f32, device=logit.device)
        self._call("cwf_upsample_softmax", logit.data_ptr(), logit.stride(3), prob.data_ptr(), n, d, h, w, c, scale, self._stream())
        return prob

    def upsample_softmax_bwd(self, dprob, prob, lo_shape, c, scale, ldc_out):
        n, d, h, w = lo_shape
        dprob = dprob.contiguous()
        dl = torch.zeros((n, d, h, w, ldc_out), dtype=_f32, device=prob.device)
        ws = self.workspace("upsm_bwd", n * d * scale * h * w * c, prob.device)
        self._call("cwf_upsample_softmax_bwd", dprob.data_ptr(), prob.data_ptr(), dl.data_ptr(), ldc_out, n, d, h, w, c, scale, ws.data_ptr(), self._stream())
        return dl

    def channel_softmax(self, logit, c):
        n, d, h, w, _ = logit.shape
        prob = torch.empty((n, d, h, w, c), dtype=_f32, device=logit.device)
        self._call("cwf_channel_softmax", logit.data_ptr(), logit.stride(3), prob.data_ptr(), n * d * h * w, c, self._stream())
        return prob

    def channel_softmax_bwd(self, dprob, prob):
        dprob = dprob.contiguous()
        n, d, h, w, c = prob.shape
        dl = torch.empty_like(prob)
        self._call("cwf_channel_softmax_bwd", dprob.data_ptr(), prob.data_ptr(), dl.data_ptr(), c, n * d * h * w, c, self._stream())
        return dl

    # ------------------------------------------------------------------ K9
    def dice_ce(self, prob, label, posmask):
        """prob [N,D,H,W,C] dense channels-last, label int64 [N,D,H,W] -> (loss [1], coef [N,C,4])"""
        n, d, h, w, c = prob.shape
        v = d * h * w
        sums = self._zeros_f64((n, c, 4), prob.device)
        self._call("cwf_dice_ce_sums", prob.data_ptr(), label.data_ptr(), posmask, sums.data_ptr(), n, v, c, self._stream())
        loss = torch.empty(1, dtype=_f32, device=prob.device)
        coef = torch.empty((n, c, 4), dtype=_f32, device=prob.device)
        self._call("cwf_dice_ce_finalize", sums.data_ptr(), loss.data_ptr(), coef.data_ptr(), n, v, c, self._stream())
        return loss, coef

    def dice_ce_bwd(self, prob, label, posmask, coef, gscale):
        n, d, h, w, c = prob.shape
        dprob = torch.empty_like(prob)
        self._call("cwf_dice_ce_bwd", prob.data_ptr(), label.data_ptr(), posmask, coef.data_ptr(), gscale.data_ptr(), dprob.data_ptr(),
                   n, d * h * w, c, self._stream())
        return dprob

    def head_loss(self, logits, label, posmasks, scale):
        """Fused head -> loss forward: logits = up to 3 low-res [N,d,h,w,ldc] tensors (2 channels used), label int64 [N,D,H,W].
        -> (total [1], loss [nm], coef [nm,N,2,4])"""
        nm = len(logits)
        n, d, h, w, _ = logits[0].shape
        ldc = logits[0].stride(3)                  # (channel slices of one grouped logit buffer: rows ldc floats apart)
        assert all(t.shape == logits[0].shape and t.stride() == logits[0].stride() and t.stride(4) == 1 for t in logits) and label.is_contiguous()
        assert logits[0].stride(2) == w * ldc and logits[0].stride(1) == h * w * ldc and logits[0].stride(0) == d * h * w * ldc
        dev = logits[0].device
        sums = self._zeros_f64((nm, n, 2, 4), dev)
        ptrs = (ctypes.c_void_p * nm)(*[t.data_ptr() for t in logits])
        masks = (ctypes.c_uint32 * nm)(*[int(m) for m in posmasks])
        self._call("cwf_head_loss_sums", ctypes.addressof(ptrs), nm, ldc, ctypes.addressof(masks), label.data_ptr(), sums.data_ptr(), n, d, h, w, scale, self._stream())
        loss = torch.empty(nm, dtype=_f32, device=dev)
        total = torch.empty(1, dtype=_f32, device=dev)
        coef = torch.empty((nm, n, 2, 4), dtype=_f32, device=dev)
        self._call("cwf_dice_ce_finalize_multi", sums.data_ptr(), loss.data_ptr(), coef.data_ptr(), total.data_ptr(), nm, n,
                   d * h * w * scale ** 3, 2, self._stream())
        return total, loss, coef

    def head_loss_bwd(self, logits, label, posmasks, scale, coef, gscale, grouped_out=None):
        """-> list of d(logit) tensors.  grouped_out = (buffer [N,d,h,w,nm*ca], ca): the gradients are written as channel groups of
        that ONE buffer (ca channels each: 2 values + zeroed padding) and the returned tensors are its slices."""
        nm = len(logits)
        n, d, h, w, ca = logits[0].shape
        ldc = logits[0].stride(3)
        dev = logits[0].device
        if grouped_out is None:
            dls = [torch.empty((n, d, h, w, ca), dtype=_f32, device=dev) for _ in logits]
            dl_ca, dl_ldc = ca, ca
        else:
            buf, dl_ca = grouped_out
            dl_ldc = buf.stride(3)
            dls = [buf[..., m * dl_ca:(m + 1) * dl_ca] for m in range(nm)]
        ws = self.workspace("head_loss_bwd", nm * n * d * scale * h * w * 2, dev)
        ptrs = (ctypes.c_void_p * nm)(*[t.data_ptr() for t in logits])
        dptrs = (ctypes.c_void_p * nm)(*[t.data_ptr() for t in dls])
        masks = (ctypes.c_uint32 * nm)(*[int(m) for m in posmasks])
        self._call("cwf_head_loss_bwd_ex", ctypes.addressof(ptrs), nm, ldc, ctypes.addressof(masks), label.data_ptr(), coef.data_ptr(), gscale.data_ptr(),
                   ctypes.addressof(dptrs), dl_ca, dl_ldc, ws.data_ptr(), n, d, h, w, scale, self._stream())
        return dls

    # ------------------------------------------------------------------ N1 (sliding-window inference glue)
    def stitch_windows(self, out8, nb):
        """out8: the model's output for the 8 windows stacked along the batch, logical [8*nb,4,128,128,128] over channels-last memory
        -> stitched [nb,4,240,240,155] (predict_overlap.py:49-58, incl. the depth-axis offset quirk)"""
        w = out8.permute(0, 2, 3, 4, 1)
        if not w.is_contiguous():
            w = w.contiguous()
        assert tuple(w.shape) == (8 * nb, 128, 128, 128, 4)
        y = torch.empty((nb, 4, 240, 240, 155), dtype=_f32, device=out8.device)
        self._call("cwf_stitch_windows", w.data_ptr(), y.data_ptr(), nb, self._stream())
        return y

    def argmax_dice(self, prob, target=None, miou=False):
        """prob [B,4,...] (any strides with one voxel stride) -> (seg int64 [B,...], [WT, TC, ET] Dice tensor or None); with miou=True
        (target required) a third result: the [class 1, 2, 3] IoU tensor of tools.softmax_mIOU_score (utils/tools.py:50-61)"""
        b = prob.shape[0]
        sp = tuple(prob.shape[2:])
        v = 1
        for d in sp:
            v *= d
        if prob.is_contiguous():
            sb, sc, sv = 4 * v, v, 1
        elif prob.permute(0, 2, 3, 4, 1).is_contiguous():
            sb, sc, sv = 4 * v, 1, 4
        else:
            prob = prob.contiguous(); sb, sc, sv = 4 * v, v, 1
        seg = torch.empty((b,) + sp, dtype=torch.int64, device=prob.device)
        counts = None
        if target is not None:
            target = target.contiguous()
            assert target.dtype == torch.int64 and tuple(target.shape) == (b,) + sp
            counts = torch.zeros((6 if miou else 3, 3), dtype=torch.int64, device=prob.device)
        if miou:
            assert target is not None, "mIoU needs a target"
            self._call("cwf_argmax_metrics", prob.data_ptr(), sb, sc, sv, target.data_ptr(), seg.data_ptr(), counts.data_ptr(), b, v, self._stream())
        else:
            self._call("cwf_argmax_dice", prob.data_ptr(), sb, sc, sv, _p(target), seg.data_ptr(), _p(counts), b, v, self._stream())
        dice = None
        if counts is not None:
            c = counts.double()
            dice = (2 * c[:3, 0] + 1e-8) / (c[:3, 1] + c[:3, 2] + 1e-8)       # tools.dice_score (utils/tools.py:44-47)
            if miou:
                iou = (c[3:, 0] + 1e-8) / (c[3:, 1] + c[3:, 2] - c[3:, 0] + 1e-8)   # tools.mIOU (utils/tools.py:50-53): |o & t| / |o | t|
                return seg, dice, iou
        return seg, dice

    # ------------------------------------------------------------------ K11 / misc
    def adam(self, table, ntensors, max_n, lr, beta1, beta2, eps, wd, step, amsgrad, hyper_dev=None, grad_scale=1.0):
        self._call("cwf_adam_amsgrad_scaled", table.data_ptr(), ntensors, max_n, lr, beta1, beta2, eps, wd, step, int(amsgrad),
                   _p(hyper_dev), float(grad_scale), self._stream())

    def dropout_mask(self, shape, p, device, p2=0.0):
        """Pre-scaled keep mask(s) in one launch from the device generator state (capturable: the state advances by a kernel)."""
        m = torch.empty(shape, dtype=_f32, device=device)
        n = m.numel()
        self._call("cwf_dropout_mask_rng", m.data_ptr(), n, float(p), float(p2), self.rng(device).data_ptr(), self.rng_site(n), self._stream())
        return m

    def mul(self, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        self._call("cwf_mul", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), self._stream())
        return y

    def add(self, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        self._call("cwf_add", a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), self._stream())
        return y

    def channel_scale(self, x, s):
        x, ldc = cl(x)
        n, d, h, w, c = x.shape
        y = torch.empty((n, d, h, w, c), dtype=_f32, device=x.device)
        self._call("cwf_channel_scale", x.data_ptr(), ldc, s.data_ptr(), y.data_ptr(), c, n, d * h * w, c, self._stream())
        return y

    def copy_into(self, x, out):
        """out[..., :] = x for channels-last views (zero-copy concat helper)."""
        x, ldc = cl(x)
        n, d, h, w, c = x.shape
        self._call("cwf_copy_strided", x.data_ptr(), ldc, out.data_ptr(), out.stride(3), n * d * h * w, c, self._stream())
        return out


_backend = None


def backend():
    """The kernel backend of the product path: always the HIP library."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def _set_backend_for_testing(obj):
    """TEST-ONLY hook (tests/ inject oracle/kernel_emul.py to exercise the host-side autograd wiring on CPU).
    Nothing in the package calls this."""
    global _backend
    _backend = obj
