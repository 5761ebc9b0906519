"""torch.autograd.Function glue over the HIP kernels (cwf/kernels.py).  Python only sequences launches and owns
tensors; all arithmetic is in csrc/*.hip.  Activations are [N, D, H, W, C] (NDHWC), tokens [B, T, E].

Fusion choices: InstanceNorm statistics of a conv OUTPUT come from that conv's epilogue; the
normalise + (Leaky)ReLU of a conv INPUT is applied in the consumer's staging prologue, in forward, in the weight
gradient (recomputed) and undone analytically in the data gradient (full InstanceNorm backward).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import packing as pk
from .kernels import backend, precision


def active_sink():
    from .optim import active_sink as f
    return f()


# ======================================================================================================
# per-layer conv description + once-per-step weight packing
# ======================================================================================================
class ConvSpec:
    """Static description of one conv layer: op code, channel counts and the index maps (device tensors) for the
    fp32 and the split-bf16 kernel forms."""

    def __init__(self, op, cin, cout):
        self.op, self.cin, self.cout = op, cin, cout
        self.cout_alloc = (cout + 3) // 4 * 4
        self.np_fwd = pk.fwd_map(op, cin, cout)
        self.np_dgrad = pk.dgrad_map(op, cin, cout, self.cout_alloc)
        self.np_fwd16 = pk.fwd_map16(op, cin, cout)
        self.np_dgrad16 = pk.dgrad_map16(op, cin, cout, self.cout_alloc)
        self.np_inv, self.has_bias_map, self.slab = pk.wgrad_inverse_map(op, cin, cout)
        self.dev = None
        self.uses = 0            # forward applications since the last WeightPacker.refresh() (= per model forward)
        self.fwd_map = self.dgrad_map = self.inv_map = self.fwd_map16 = self.dgrad_map16 = None
        self.wpk_f = self.wpk_d = self.wpk16_f = self.wpk16_d = None

    def to(self, device):
        if self.dev == device:
            return self
        t = lambda a: None if a is None else torch.from_numpy(a).to(device)
        self.fwd_map, self.dgrad_map, self.inv_map = t(self.np_fwd), t(self.np_dgrad), t(self.np_inv)
        self.fwd_map16, self.dgrad_map16 = t(self.np_fwd16), t(self.np_dgrad16)
        self.wpk_f = torch.zeros(self.np_fwd.size, dtype=torch.float32, device=device)
        self.wpk_d = torch.zeros(self.np_dgrad.size, dtype=torch.float32, device=device)
        # split-bf16: hi and lo images side by side = 2 bf16 per map entry = one float32 slot per entry
        self.wpk16_f = torch.zeros(self.np_fwd16.size, dtype=torch.float32, device=device)
        self.wpk16_d = torch.zeros(self.np_dgrad16.size, dtype=torch.float32, device=device)
        self.dev = device
        return self

    def packed(self, dgrad=False):
        """The packed weight buffer the active kernel form consumes."""
        if precision() == "fp32":
            return self.wpk_d if dgrad else self.wpk_f
        return self.wpk16_d if dgrad else self.wpk16_f


class FusedConvSpec(ConvSpec):
    """Three convolutions that read the SAME input (the sub-region decouplers conv_semantic_{1,2,4}: 256 -> 128 each, and
    conv_mid_fea_{1,2,4}: 96 -> 32 each; cls_wise_former.py:157-204) as ONE conv with 3 x cout output channels: the input is
    staged once instead of three times and the launch has three times the workgroups.  The parameters stay three separate
    tensors (checkpoint layout): each source fills its own third of the packed operand (map value -2 = "not mine")."""

    def __init__(self, op, cin, cout_each):
        super().__init__(op, cin, 3 * cout_each)
        self.cout_each = cout_each
        taps = {pk.CONV3_S1: 27, pk.CONV3_S2: 27, pk.CONV1: 1}[op]
        self.src_numel = cout_each * cin * taps

        def split(m):
            out = []
            for g in range(3):
                own = (m >= 0) & (m // self.src_numel == g)
                mg = np.where(own, m - g * self.src_numel, -2).astype(np.int32)
                if g == 0:
                    mg = np.where(m == -1, -1, mg).astype(np.int32)      # the padding zeros are written once, by source 0
                out.append(mg)
            return out
        self.np_parts = {k: split(getattr(self, k)) for k in ("np_fwd", "np_dgrad", "np_fwd16", "np_dgrad16")}
        self.np_bias = [np.where(np.arange(3 * cout_each) // cout_each == g, np.arange(3 * cout_each) - g * cout_each, -2).astype(np.int32)
                        for g in range(3)]
        self.part_maps = None
        self.bias_all = None

    def to(self, device):
        if self.dev == device:
            return self
        super().to(device)
        t = lambda a: torch.from_numpy(a).to(device)
        self.part_maps = {k: [t(m) for m in v] for k, v in self.np_parts.items()}
        self.bias_maps = [t(m) for m in self.np_bias]
        self.bias_all = torch.zeros(self.cout, dtype=torch.float32, device=device)
        return self


class WeightPacker:
    """Packs every conv weight into the MFMA B-operand layouts (forward + data-gradient forms) with ONE launch
    per step (cwf_gather_batched / cwf_gather_split_bf16 over a device-resident descriptor table)."""

    def __init__(self):
        self.items = []          # (spec, weight Parameter)
        self.fused = []          # (FusedConvSpec, [3 weights], [3 biases])
        self._key = None
        self._tables = None

    def add(self, spec, weight):
        self.items.append((spec, weight))

    def add_fused(self, spec, weights, biases):
        self.fused.append((spec, list(weights), list(biases)))

    def refresh(self):
        if not self.items:
            return
        for spec, _ in self.items:
            spec.uses = 0
        for spec, _, _ in self.fused:
            spec.uses = 0
        dev = self.items[0][1].device
        key = (dev, tuple(w.data_ptr() for _, w in self.items), tuple(t.data_ptr() for _, ws, bs in self.fused for t in ws + bs))
        if key != self._key:
            r32, r16, rb = [], [], []
            for spec, ws, bs in self.fused:
                spec.to(dev)
                for g, (w, b) in enumerate(zip(ws, bs)):
                    assert w.is_contiguous() and b.is_contiguous()
                    pm = spec.part_maps
                    r32.append([w.data_ptr(), spec.wpk_f.data_ptr(), pm["np_fwd"][g].data_ptr(), pm["np_fwd"][g].numel()])
                    r32.append([w.data_ptr(), spec.wpk_d.data_ptr(), pm["np_dgrad"][g].data_ptr(), pm["np_dgrad"][g].numel()])
                    r16.append([w.data_ptr(), spec.wpk16_f.data_ptr(), pm["np_fwd16"][g].data_ptr(), pm["np_fwd16"][g].numel()])
                    r16.append([w.data_ptr(), spec.wpk16_d.data_ptr(), pm["np_dgrad16"][g].data_ptr(), pm["np_dgrad16"][g].numel()])
                    rb.append([b.data_ptr(), spec.bias_all.data_ptr(), spec.bias_maps[g].data_ptr(), spec.bias_maps[g].numel()])
            for spec, w in self.items:
                spec.to(dev)
                assert w.is_contiguous()
                r32.append([w.data_ptr(), spec.wpk_f.data_ptr(), spec.fwd_map.data_ptr(), spec.fwd_map.numel()])
                r32.append([w.data_ptr(), spec.wpk_d.data_ptr(), spec.dgrad_map.data_ptr(), spec.dgrad_map.numel()])
                r16.append([w.data_ptr(), spec.wpk16_f.data_ptr(), spec.fwd_map16.data_ptr(), spec.fwd_map16.numel()])
                r16.append([w.data_ptr(), spec.wpk16_d.data_ptr(), spec.dgrad_map16.data_ptr(), spec.dgrad_map16.numel()])
            self._tables = {"fp32": (torch.tensor(r32, dtype=torch.int64).to(dev), max(r[3] for r in r32)),
                            "bf16": (torch.tensor(r16, dtype=torch.int64).to(dev), max(r[3] for r in r16)),
                            "bias": (torch.tensor(rb, dtype=torch.int64).to(dev), max(r[3] for r in rb)) if rb else None}
            self._key = key
        kind = "fp32" if precision() == "fp32" else "bf16"
        table, max_n = self._tables[kind]
        backend().gather_batched(table, table.shape[0], max_n, split_bf16=(kind == "bf16"))
        if self._tables["bias"] is not None:      # the fused layers' concatenated biases (fp32 gather in every mode)
            table, max_n = self._tables["bias"]
            backend().gather_batched(table, table.shape[0], max_n, split_bf16=False)


# ======================================================================================================
# conv family
# ======================================================================================================
class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, spec, in_scale, in_shift, slope, residual, out_scale, want_stats, carry, x_link, out=None, res_link=None,
                carry_link=None, y16=None):
        K = backend()
        ctx.set_materialize_grads(False)     # (autograd would zero-FILL a gradient for each of the (scale, shift) outputs: 2 launches per conv)
        ctx.x_link = x_link
        ctx.res_link, ctx.carry_link = res_link, carry_link
        if spec.dev is not None and x.device != spec.dev:
            # packed weights, index maps and kernel attributes of a layer live on ONE device; a multi-device nn.DataParallel replica
            # would launch with another device's pointers (a GPU memory fault): one process per GPU is the supported layout
            raise RuntimeError("this conv layer's packed weights live on %s but its input is on %s: multi-device replicas inside one "
                               "process (nn.DataParallel over several GPUs) are not supported; use one process per GPU (INTEGRATION.md)"
                               % (spec.dev, x.device))
        n = x.shape[0]
        spec.uses += 1
        stats = K.new_stats(n, spec.cout, x.device) if want_stats else None
        # out: a channel slice of a wider buffer (the conv writes its half of a concatenation in place: no copy launch)
        kwy = dict(y16=y16) if y16 is not None else {}       # (a preallocated bf16 image of the output, written by the same launch)
        y = K.conv(spec.op, x, spec.packed(False), b, spec.cout, in_scale, in_shift, slope, residual, out_scale, stats,
                   out=out, w_ref=w, out_channels_alloc=spec.cout_alloc, **kwy)
        ctx.spec, ctx.slope = spec, slope
        ctx.bias_ref = b
        ctx.has_res = residual is not None
        # bf16 operand images for this layer's weight gradient (full-resolution 16-channel layers, HipBackend.bf16_operands_ok)
        ok16 = getattr(K, "bf16_operands_ok", None)
        ctx.use16 = bool(ok16 is not None and ok16(spec.op, spec.cin, spec.cout, y.shape[1] * y.shape[2] * y.shape[3]))
        ctx.x16 = getattr(x, "_cwf16x", None) if (ctx.use16 and in_scale is None) else None     # bf16(x) from x's producer (a block tail)
        if ctx.use16 and ctx.x16 is None and ctx.needs_input_grad[1] and getattr(K, "wgrad_async", False) and \
                (getattr(K, "xa16_in_forward", False) or (getattr(x, "_cwf_wgrad_first", False) and _FIRST_X16_FWD)):   # (wgrad_async: a Trainer step)
            # the weight-gradient operand bf16(act(IN(x))) depends on forward data only: made NOW, on the weight-gradient side stream,
            # which idles during the forward pass -- the backward pass then neither converts it nor waits for this layer's own
            # InstanceNorm-backward apply pass (which matters for the last layers of backward: nothing is left to hide behind)
            ctx.x16 = K.to_bf16_side(x, in_scale, in_shift, slope)
        ctx.up16 = getattr(x, "_cwf_want16", False)     # x's producer is such a layer: hand its gradient on with a bf16 image ("only": nothing else)
        # the LAST layers of backward: nothing is left to hide their weight gradient behind, so it must not wait for this layer's own data
        # gradient + apply pass (which would deliver xa16): it is issued first and converts x itself on the side stream
        ctx.wgrad_first = bool(getattr(x, "_cwf_wgrad_first", False))
        ctx.save_for_backward(x, w, in_scale, in_shift, out_scale)
        # carry: x is handed on as a second output (an alias).  Whatever consumes that alias (a residual connection, a skip
        # connection) sends its gradient back HERE, where it is folded into the kernel that writes dx (dx_add of the
        # InstanceNorm-backward apply, or the residual operand of the data-gradient conv) instead of autograd summing the
        # two gradients of x with a separate full-resolution add launch.
        xc = x if carry else None
        if want_stats:
            sc, sh = K.in_finalize(stats, y.shape[1] * y.shape[2] * y.shape[3])
            ctx.mark_non_differentiable(sc, sh)
            return y, sc, sh, xc
        return y, None, None, xc

    @staticmethod
    def backward(ctx, dy, _a, _b, dcarry):
        K = backend()
        if ctx.carry_link is not None and ctx.carry_link.grads:
            # gradients the consumers of the carried alias handed over directly (CarryLink) join whatever autograd delivered
            for gq in ctx.carry_link.grads:
                dcarry = gq if dcarry is None else K.add(dcarry, gq)
            ctx.carry_link.grads = []
        if dy is None:                       # y itself unused downstream: only the carried alias has a gradient
            return (dcarry,) + (None,) * 15
        x, w, in_scale, in_shift, out_scale = ctx.saved_tensors
        spec = ctx.spec
        dys = None
        if out_scale is not None:
            sink0 = active_sink()
            fold = (sink0 is not None and not ctx.needs_input_grad[0] and not ctx.has_res and spec.uses <= 1 and ctx.needs_input_grad[1] and
                    sink0.view(w) is not None and (ctx.bias_ref is None or sink0.view(ctx.bias_ref) is not None) and
                    getattr(K, "dy_scale_ok", lambda *a: False)(spec.op, spec.cin, spec.cout, dy.shape[1] * dy.shape[2] * dy.shape[3])
                    and dy.shape[-1] == spec.cout and dy.is_contiguous())
            if fold:
                dys = out_scale                       # the weight gradient (the only reader of dy here) scales dy while staging it
                dy = _dy_f32(K, dy)
            else:
                dy = K.channel_scale(_dy_f32(K, dy), out_scale)
        dres = dy if ctx.has_res else None
        dy_private = not ctx.has_res
        if ctx.has_res and ctx.res_link is not None:
            # the residual is a carried alias: its gradient (= dy) goes to the producing conv's backward through the link, not through
            # autograd -- dy stays a tensor only this code references, so the side stream may read it (see CarryLink)
            ctx.res_link.grads.append(dy)
            dres = None
            dy_private = True
        dw = db = dx = None
        sink = active_sink()
        sw = sink.view(w) if (sink is not None and spec.uses <= 1 and ctx.needs_input_grad[1]) else None
        sb = sink.view(ctx.bias_ref) if (sw is not None and ctx.bias_ref is not None) else None
        to_sink = sw is not None and (ctx.bias_ref is None or sb is not None)
        use16 = ctx.use16 and to_sink and out_scale is None
        dy16 = getattr(dy, "_cwf16", None) if out_scale is None else None
        if (_DY16_ON_MAIN and dy16 is None and use16 and ctx.needs_input_grad[0] and not getattr(dy, "_cwf_f32_missing", False) and
                getattr(K, "bf16_dgrad_ok", lambda *a: False)(spec.op, spec.cin, spec.cout, dy.shape[1] * dy.shape[2] * dy.shape[3])
                and dy.shape[-1] == spec.cout):
            # no producer wrote the bf16 image of this gradient (it comes out of a stride-2 data gradient): one conversion pass HERE,
            # on the main stream, serves both the data gradient (LDS-DMA form) and the weight gradient -- left to the weight gradient it
            # runs on the side stream beside the HBM-bound end of backward (480 us instead of 65) and delays the optimizer
            dy16 = K.to_bf16(dy)
        dg16 = dy16 if (dy16 is not None and getattr(K, "bf16_dgrad_ok", lambda *a: False)(
            spec.op, spec.cin, spec.cout, dy.shape[1] * dy.shape[2] * dy.shape[3])) else None     # the data gradient reads the bf16 image
        if getattr(dy, "_cwf_f32_missing", False) and not (use16 and dy16 is not None and (dg16 is not None or not ctx.needs_input_grad[0])
                                                           and not ctx.has_res):
            dy = _dy_f32(K, dy)                         # some consumer below reads the fp32 tensor
        if not use16:
            dy16 = None
        xa16 = ctx.x16 if use16 else None

        def issue_sink_wgrad(xa16_):
            # gradient-sink path (Trainer): slabs now, ONE batched reduce per backward phase writes dW / db into the flat buffer
            dyv = dy[..., :spec.cout] if dy.shape[-1] != spec.cout else dy
            kw = dict(x16=xa16_, dy16=dy16) if (use16 and (xa16_ is not None or dy16 is not None)) else {}
            if dys is not None:
                kw["dy_scale"] = dys
            K.wgrad_to(spec, spec.op, x, in_scale, in_shift, ctx.slope, dyv, spec.cout, spec.inv_map, sw, sb if spec.has_bias_map else None,
                       allow_async=dy_private, **kw)
            sink.mark(w)
            if sb is not None:
                if not spec.has_bias_map:   # ConvTranspose: the bias gradient spans the 8 parity classes
                    if hasattr(K, "channel_sum_to"):
                        K.channel_sum_to(dy, sb, allow_async=dy_private)      # (a full pass over dy: on the weight-gradient side stream)
                    else:
                        K.stats_channel_sum(K.in_stats(dy), sb)
                sink.mark(ctx.bias_ref)

        early = to_sink and use16 and ctx.wgrad_first
        if early:
            issue_sink_wgrad(xa16)
        if ctx.needs_input_grad[0]:
            dxa = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            fused = in_scale is not None and getattr(K, "supports_fused_norm_bwd", lambda: False)()
            kw16 = dict(x16=dg16) if dg16 is not None else {}
            if fused:
                # the InstanceNorm-backward sums come out of the data gradient's epilogue: one pass over (g, x) less
                sums = K.new_stats(x.shape[0], spec.cin, x.device)
                K.conv(pk.dgrad_op(spec.op), dy, spec.packed(True), None, spec.cin, out=dxa, w_ref=w, fwd_op=spec.op,
                       stats=sums, nb=(x, in_scale, in_shift, ctx.slope), **kw16)
                emits = getattr(K, "APPLY_EMITS", ())
                want_xa, want_dx = use16 and "xa" in emits and not early, ctx.up16 and "dx" in emits
                if want_xa or want_dx:
                    # the apply pass has x, its statistics and dx in registers: it also writes this layer's weight-gradient operand
                    # bf16(act(IN(x))) and the bf16 image of dx for the layer that produced x -- and ONLY that image where the layer
                    # that produced x reads nothing else (ctx.up16 == "only", a single-consumer graph)
                    only16 = want_dx and ctx.up16 == "only" and _SINGLE_CONSUMER and to_sink
                    dx, dx16, xa16 = K.in_bwd_apply16(dxa, x, in_scale, in_shift, ctx.slope, sums, dx_add=dcarry,
                                                      want_dx16=want_dx, want_xa16=want_xa, need_f32=not only16)
                    if dx16 is not None:
                        dx._cwf16 = dx16
                        if only16:
                            dx._cwf_f32_missing = True
                else:
                    dx = K.in_bwd_apply(dxa, x, in_scale, in_shift, ctx.slope, sums, dx_add=dcarry)
            elif in_scale is not None:
                K.conv(pk.dgrad_op(spec.op), dy, spec.packed(True), None, spec.cin, out=dxa, w_ref=w, fwd_op=spec.op)
                dx = K.in_bwd(dxa, x, in_scale, in_shift, ctx.slope, dx_add=dcarry)
            else:
                link = ctx.x_link
                if link is not None and not link.shared and getattr(K, "supports_fused_norm_bwd", lambda: False)():
                    # x is the output of a block tail act(IN(g)) + r whose only consumer is this conv: dx (carry included) is the
                    # tail's dL/dy, so the epilogue also accumulates the tail's InstanceNorm-backward sums
                    link.sums = K.new_stats(x.shape[0], spec.cin, x.device)
                    dx = K.conv(pk.dgrad_op(spec.op), dy, spec.packed(True), None, spec.cin, out=dxa, w_ref=w, fwd_op=spec.op, residual=dcarry,
                                stats=link.sums, nb=(link.g, link.scale, link.shift, link.slope), **kw16)
                else:
                    dx = K.conv(pk.dgrad_op(spec.op), dy, spec.packed(True), None, spec.cin, out=dxa, w_ref=w, fwd_op=spec.op, residual=dcarry, **kw16)
        elif dcarry is not None:
            dx = dcarry
        if to_sink:
            if not early:
                issue_sink_wgrad(xa16)
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dyv = dy[..., :spec.cout] if dy.shape[-1] != spec.cout else dy
            # side-stream weight gradients only where nothing on the main stream can touch their operands or results early:
            #  * a weight applied twice in the forward, or a .grad that already exists, makes AccumulateGrad add IN PLACE;
            #  * with a residual, dy itself is handed on as the residual's gradient and the engine may accumulate into it
            #    in place while the side stream still reads it (record_stream guards reuse, not modification)
            once = spec.uses <= 1 and w.grad is None and dy_private
            dwf, db = K.wgrad(spec.op, x, in_scale, in_shift, ctx.slope, dyv, spec.cout, spec.inv_map, spec.has_bias_map, w.numel(), w_ref_shape=w.shape,
                              allow_async=once)
            dw = dwf.view(w.shape)
            if db is None:      # ConvTranspose: bias gradient spans the 8 parity classes
                db = K.stats_channel_sum(K.in_stats(dy), torch.empty(spec.cout, dtype=torch.float32, device=dy.device))
        return dx, dw, db, None, None, None, None, dres, None, None, None, None, None, None, None, None


class _FusedConvFn(torch.autograd.Function):
    """y[..., g*c:(g+1)*c] = conv(x; w_g) + b_g for the three same-input convs of a FusedConvSpec, + InstanceNorm statistics."""

    @staticmethod
    def forward(ctx, x, w0, w1, w2, b0, b1, b2, spec):
        K = backend()
        ctx.set_materialize_grads(False)
        spec.uses += 1
        stats = K.new_stats(x.shape[0], spec.cout, x.device)
        # (w_ref / bias_ref: the ORIGINAL parameters, which every conv call hands along; the kernels read the packed operands)
        y = K.conv(spec.op, x, spec.packed(False), spec.bias_all, spec.cout, None, None, 1.0, None, None, stats, w_ref=(w0, w1, w2), bias_ref=(b0, b1, b2))
        sc, sh = K.in_finalize(stats, y.shape[1] * y.shape[2] * y.shape[3])
        ctx.mark_non_differentiable(sc, sh)
        ctx.spec = spec
        ctx.params = (w0, w1, w2, b0, b1, b2)
        ctx.save_for_backward(x)
        return y, sc, sh

    @staticmethod
    def backward(ctx, dy, _a, _b):
        K = backend()
        (x,) = ctx.saved_tensors
        spec = ctx.spec
        w0, w1, w2, b0, b1, b2 = ctx.params
        dy = dy.contiguous()
        sink = active_sink()
        views = [sink.view(t) for t in ctx.params] if (sink is not None and spec.uses <= 1) else None
        adjacent = views is not None and all(v is not None for v in views) and \
            views[1].data_ptr() == views[0].data_ptr() + 4 * w0.numel() and views[2].data_ptr() == views[1].data_ptr() + 4 * w1.numel() and \
            views[4].data_ptr() == views[3].data_ptr() + 4 * b0.numel() and views[5].data_ptr() == views[4].data_ptr() + 4 * b1.numel()
        grads = (None,) * 6
        if adjacent:           # the three weight (bias) gradients are adjacent slices of the flat buffer: one reduce target each
            K.wgrad_to(spec, spec.op, x, None, None, 1.0, dy, spec.cout, spec.inv_map, views[0], views[3], allow_async=True)
            for t in ctx.params:
                sink.mark(t)
        else:
            n = w0.numel()
            dwf, db = K.wgrad(spec.op, x, None, None, 1.0, dy, spec.cout, spec.inv_map, True, 3 * n,
                              w_ref_shape=(spec.cout,) + tuple(w0.shape[1:]), allow_async=False)
            c = spec.cout_each
            grads = (dwf[:n].view_as(w0), dwf[n:2 * n].view_as(w1), dwf[2 * n:].view_as(w2), db[:c], db[c:2 * c], db[2 * c:])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            K.conv(pk.dgrad_op(spec.op), dy, spec.packed(True), None, spec.cin, out=dx, w_ref=(w0, w1, w2), fwd_op=spec.op)
        return (dx,) + grads + (None,)


def fused_conv3(x, convs, spec):
    """convs: three HipConv-like parameter holders (.weight, .bias) sharing the input x.  Returns (y [N,D,H,W,3*cout], (scale, shift))."""
    y, sc, sh = _FusedConvFn.apply(x, convs[0].weight, convs[1].weight, convs[2].weight, convs[0].bias, convs[1].bias, convs[2].bias, spec)
    return y, (sc, sh)


# A model may declare that every tensor it hands to conv() / norm_act_add() has ONE gradient consumer (ClsWiseFormer does, in its
# forward).  Only then may a backward pass leave the fp32 gradient of such a tensor unwritten and hand on its bf16 image alone
# (autograd would otherwise add the unwritten buffer to another consumer's gradient).
_SINGLE_CONSUMER = False
import os as _os
_DY16_ON_MAIN = _os.environ.get("CWF_DY16_ON_MAIN", "0") != "0"     # (measured 0.3 % slower than leaving the conversion to the side stream)
_FIRST_X16_FWD = _os.environ.get("CWF_FIRST_X16_FWD", "1") != "0"     # the layer behind the stem: its xa16 is made in the forward pass (side stream)


def set_single_consumer_graph(flag: bool):
    global _SINGLE_CONSUMER
    _SINGLE_CONSUMER = bool(flag)


def _dy_f32(K, dy):
    """the fp32 gradient, materialised from its bf16 image if the producer left the fp32 tensor unwritten (rare paths only)"""
    if getattr(dy, "_cwf_f32_missing", False):
        return dy._cwf16.float()
    return dy


class CarryLink:
    """Side channel for the gradient of a carried alias (conv(..., carry=True) hands its input on as a second output; a residual
    connection consumes it).  A conv that takes the alias as its `residual` appends dL/d(residual) -- which IS its incoming dy -- here
    instead of returning it to autograd; the producing conv's backward collects it (and adds whatever reached the alias through
    autograd from other consumers).  The point: dy is then referenced by this code alone, nothing can accumulate into it in place,
    and the residual layer's weight gradient may read it from the side stream like every other layer's (before: the eight EnBlock
    conv2 weight gradients, 1.3 ms per step, ran on the main stream)."""
    __slots__ = ("grads",)

    def __init__(self):
        self.grads = []


def conv(x, w, b, spec, in_norm=None, slope=1.0, residual=None, out_scale=None, want_stats=False, carry=False, out=None, emit16=False):
    """y = conv(act(IN(x)))(+bias)(+residual)(*out_scale).  in_norm = (scale, shift) of x or None.
    Returns (y, (scale_y, shift_y) or None), with carry=True (y, stats, x_alias): use x_alias for every further use of x
    (see _ConvFn.forward).  out: write y into this [N,D,H,W,cout] view (a channel slice of a concatenation buffer, see cat_into)."""
    sc, sh = in_norm if in_norm is not None else (None, None)
    link = getattr(x, "_cwf_link", None)
    if link is not None:
        if link.claimed or in_norm is not None:
            link.shared = True                       # a second consumer (or a normalising one): the tail keeps its own reduction
        link.claimed = True
        if link.shared:
            link = None
    res_link = getattr(residual, "_cwf_carry", None) if residual is not None else None
    carry_link = CarryLink() if (carry and torch.is_grad_enabled()) else None
    # emit16: the output is the un-normalised input of a layer whose weight gradient reads bf16 operand images (a 1x1x1 conv in front of a
    # block's first conv): the same launch writes bf16(y)
    y16 = None
    ok16_ = getattr(backend(), "bf16_operands_ok", None)
    if emit16 and ok16_ is not None and torch.is_grad_enabled() and "xa" in getattr(backend(), "APPLY_EMITS", ()) and out is None and out_scale is None \
            and not _os.environ.get("CWF_NO_PW_Y16"):
        n_, d_, h_, w_ = x.shape[0], *pk.out_dims(spec.op, x.shape[1], x.shape[2], x.shape[3])
        if ok16_(pk.CONV3_S1, spec.cout, spec.cout, d_ * h_ * w_) and (spec.cout_alloc or spec.cout) == spec.cout:
            y16 = torch.empty((n_, d_, h_, w_, spec.cout), dtype=torch.bfloat16, device=x.device)
    y, s1, s2, xc = _ConvFn.apply(x, w, b, spec, sc, sh, float(slope), residual, out_scale, want_stats, carry, link, out, res_link, carry_link, y16)
    if y16 is not None:
        y._cwf16x = y16
    if carry_link is not None and xc is not None:
        xc._cwf_carry = carry_link
    if xc is not None and getattr(x, "_cwf_catbuf", None) is not None:
        xc._cwf_catbuf = x._cwf_catbuf                   # the carried alias of a skip tensor still names its concatenation buffer
    ok16 = getattr(backend(), "bf16_operands_ok", None)
    if ok16 is not None and out_scale is None and torch.is_grad_enabled() and \
            ok16(spec.op, spec.cin, spec.cout, y.shape[1] * y.shape[2] * y.shape[3]):
        # whoever computes dL/dy (an InstanceNorm-backward apply pass) adds its bf16 image: this layer's kernels take it -- both of
        # them ("only") if its data gradient reads bf16 images too
        okd = getattr(backend(), "bf16_dgrad_ok", None)
        # (with a residual the fp32 gradient is needed as well: it is the residual's gradient)
        y._cwf_want16 = "only" if (residual is None and okd is not None and
                                   okd(spec.op, spec.cin, spec.cout, y.shape[1] * y.shape[2] * y.shape[3])) else True
    st = (s1, s2) if want_stats else None
    return (y, st, xc) if carry else (y, st)


class _GroupedConvFn(torch.autograd.Function):
    """G same-shape 3x3x3 stride-1 convs on the channel groups of ONE tensor, one launch each way:
        y_all[..., q*Ca : q*Ca + Cout] = conv_q(x_all[..., q*Cin : (q+1)*Cin]) + b_q          (Ca = Cout rounded up to 4)
    (the three sub-regions' supervision-head layers: 24 tiny convs per step whose launches, not their arithmetic, are the cost).
    Weight gradients stay per layer (their slabs go to the layer's own slice of the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, x_all, specs, *wb):
        K = backend()
        G = len(specs)
        s0 = specs[0]
        if s0.dev is not None and x_all.device != s0.dev:
            raise RuntimeError("grouped conv: packed weights live on %s but the input is on %s (one process per GPU)" % (s0.dev, x_all.device))
        n, d, h, w, _ = x_all.shape
        ca = s0.cout_alloc or s0.cout
        for s_ in specs:
            s_.uses += 1
        y_all = torch.empty((n, d, h, w, G * ca), dtype=torch.float32, device=x_all.device)
        K.conv_grouped(x_all, s0.cin, [s_.packed(False) for s_ in specs], [wb[2 * q + 1] for q in range(G)], s0.cout, y_all,
                       x_goff=s0.cin, y_goff=ca, w_refs=[wb[2 * q] for q in range(G)])
        ctx.specs = specs
        ctx.save_for_backward(x_all, *wb)
        return y_all

    @staticmethod
    def backward(ctx, dy_all):
        K = backend()
        x_all, *wb = ctx.saved_tensors
        specs = ctx.specs
        G = len(specs)
        s0 = specs[0]
        cin, cout = s0.cin, s0.cout
        ca = s0.cout_alloc or cout
        dy_all = dy_all.contiguous()
        sink = active_sink()
        grads = []
        views = None
        if sink is not None and all(spec.uses <= 1 and spec.has_bias_map for spec in specs) and all(ctx.needs_input_grad[2:]):
            views = [(sink.view(wb[2 * q]), sink.view(wb[2 * q + 1])) for q in range(G)]
            if any(v[0] is None or v[1] is None for v in views):
                views = None
        if views is not None:
            # gradient-sink path: the G layers' slabs in ONE launch (their reductions are rows of the phase's batched reduce)
            K.wgrad_to_grouped(list(specs), s0.op, [x_all[..., q * cin:(q + 1) * cin] for q in range(G)],
                               [dy_all[..., q * ca:q * ca + cout] for q in range(G)], cout, [spec.inv_map for spec in specs],
                               [v[0] for v in views], [v[1] for v in views], allow_async=True)
            for q in range(G):
                sink.mark(wb[2 * q]); sink.mark(wb[2 * q + 1])
            grads = [None, None] * G
        for q, spec in enumerate(specs if views is None else ()):
            w, b = wb[2 * q], wb[2 * q + 1]
            xs = x_all[..., q * cin:(q + 1) * cin]
            dys = dy_all[..., q * ca:q * ca + cout]
            sw = sink.view(w) if (sink is not None and spec.uses <= 1 and ctx.needs_input_grad[2 + 2 * q]) else None
            sb = sink.view(b) if (sw is not None and b is not None) else None
            if sw is not None and (b is None or sb is not None):
                K.wgrad_to(spec, spec.op, xs, None, None, 1.0, dys, cout, spec.inv_map, sw, sb if spec.has_bias_map else None, allow_async=True)
                sink.mark(w)
                if sb is not None:
                    sink.mark(b)
                grads += [None, None]
            elif ctx.needs_input_grad[2 + 2 * q]:
                dwf, db = K.wgrad(spec.op, xs, None, None, 1.0, dys, cout, spec.inv_map, spec.has_bias_map, w.numel(), w_ref_shape=w.shape,
                                  allow_async=spec.uses <= 1 and w.grad is None)
                grads += [dwf.view(w.shape), db]
            else:
                grads += [None, None]
        dx_all = None
        if ctx.needs_input_grad[0]:
            dx_all = torch.empty(x_all.shape, dtype=torch.float32, device=x_all.device)
            K.conv_grouped(dy_all, ca, [s_.packed(True) for s_ in specs], None, cin, dx_all, x_goff=ca, y_goff=cin,
                           w_refs=[wb[2 * q] for q in range(G)], fwd_op=pk.CONV3_S1)
        return (dx_all, None) + tuple(grads)


def grouped_conv(x_all, mods):
    """mods: G HipConv-like modules (weight, bias, spec) of one shape, 3x3x3 stride 1; x_all [N,d,h,w,G*Cin] -> [N,d,h,w,G*Ca]."""
    flat = []
    for m in mods:
        flat += [m.weight, m.bias]
    return _GroupedConvFn.apply(x_all, tuple(m.spec for m in mods), *flat)


class NaaLink:
    """Connects a block tail y = act(IN(g)) + x (norm_act_add) with the ONE conv that consumes y: that conv's data gradient
    produces dL/dy, and its epilogue can accumulate the two InstanceNorm-backward sums of the tail (sum dy act', sum dy act' ghat)
    while the values are in registers -- the tail's backward then skips its reduction pass over (dy, g)."""
    __slots__ = ("g", "scale", "shift", "slope", "sums", "claimed", "shared")

    def __init__(self, g, scale, shift, slope):
        self.g, self.scale, self.shift, self.slope = g, scale, shift, slope
        self.sums, self.claimed, self.shared = None, False, False


class _NormActAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, slope, residual, link, want16):
        ctx.slope = slope
        ctx.has_res = residual is not None
        ctx.link = link
        ctx.up16 = getattr(x, "_cwf_want16", False)
        ctx.save_for_backward(x, scale, shift)
        if want16:
            y, y16 = backend().norm_act_add(x, scale, shift, slope, residual, want16=True)
            ctx.mark_non_differentiable(y16)
            return y, y16
        return backend().norm_act_add(x, scale, shift, slope, residual), None

    @staticmethod
    def backward(ctx, dy, _d16):
        x, scale, shift = ctx.saved_tensors
        link = ctx.link
        K = backend()
        if link is not None and link.sums is not None and not link.shared:
            sums, link.sums = link.sums, None        # from the consuming conv's data-gradient epilogue (this backward pass)
            if ctx.up16 and "dx" in getattr(K, "APPLY_EMITS", ()):
                only16 = ctx.up16 == "only" and _SINGLE_CONSUMER and active_sink() is not None
                dx, dx16, _ = K.in_bwd_apply16(dy, x, scale, shift, ctx.slope, sums, want_dx16=True, need_f32=not only16)
                dx._cwf16 = dx16
                if only16:
                    dx._cwf_f32_missing = True
            else:
                dx = K.in_bwd_apply(dy, x, scale, shift, ctx.slope, sums)
        else:
            dx = K.in_bwd(dy, x, scale, shift, ctx.slope)
        return dx, None, None, None, (dy if ctx.has_res else None), None, None


def norm_act_add(x, stats, slope, residual=None, emit16=False):
    """emit16: the consumer of y is a full-resolution 16-channel conv without prologue -- y also leaves as a bf16 image, the operand
    of that conv's weight gradient (HipBackend.bf16_operands_ok)."""
    link = NaaLink(x, stats[0], stats[1], float(slope)) if torch.is_grad_enabled() else None
    ok16 = getattr(backend(), "bf16_operands_ok", None)
    want16 = bool(emit16 and torch.is_grad_enabled() and ok16 is not None and "xa" in getattr(backend(), "APPLY_EMITS", ()) and
                  ok16(pk.CONV3_S1, x.shape[-1], x.shape[-1], x.shape[1] * x.shape[2] * x.shape[3]))
    y, y16 = _NormActAddFn.apply(x, stats[0], stats[1], float(slope), residual, link, want16)
    if link is not None:
        y._cwf_link = link                           # picked up by the conv that takes y as its (un-normalised) input
    if y16 is not None:
        y._cwf16x = y16
    return y


class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        K = backend()
        ca, cb = a.shape[-1], b.shape[-1]
        out = torch.empty(a.shape[:-1] + (ca + cb,), dtype=torch.float32, device=a.device)
        K.copy_into(a, out[..., :ca])
        K.copy_into(b, out[..., ca:])
        ctx.ca = ca
        return out

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.ca], d[..., ctx.ca:]


def cat_channels(a, b):
    return _CatFn.apply(a, b)


class _CatIntoFn(torch.autograd.Function):
    """concat(a, b) where b was WRITTEN by its producer into the upper channels of `buf` (conv(..., out=buf[..., ca:])): only a is
    copied.  The gradient splits into two channel-slice views (no copy), exactly like _CatFn."""
    @staticmethod
    def forward(ctx, a, b, buf):
        ca = a.shape[-1]
        assert b.data_ptr() == buf[..., ca:].data_ptr() and buf.shape[-1] == ca + b.shape[-1] and b.stride(3) == buf.stride(3)
        backend().copy_into(a, buf[..., :ca])
        ctx.ca = ca
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.ca], d[..., ctx.ca:], None


class _CatPrefilledFn(torch.autograd.Function):
    """concat(a, b) where BOTH operands were written by their producers into channel slices of `buf` (a: buf[..., :Ca] -- an encoder
    block wrote its output there, see skip_buffer; b: buf[..., Ca:]): no copy at all.  The gradient splits into two channel-slice views."""
    @staticmethod
    def forward(ctx, a, b, buf):
        ca = a.shape[-1]
        assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf[..., ca:].data_ptr() and buf.shape[-1] == ca + b.shape[-1]
        assert a.stride(3) == buf.stride(3) and b.stride(3) == buf.stride(3)
        ctx.ca = ca
        return alias_channels(buf, 0, buf.shape[-1])

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.ca], d[..., ctx.ca:], None


def alias_channels(buf, c0, c1):
    """buf[..., c0:c1] as a tensor of its OWN (same storage, no autograd view relation, its own version counter): the kernels address memory
    by pointer and stride, and the producers / consumers of the slices are ordered by their autograd Functions; a real view would tie the
    slices' version counters together and make autograd reject the carried aliases of one slice once the other slice is written."""
    n, d, h, w, ct = buf.shape
    t = torch.empty(0, dtype=buf.dtype, device=buf.device)
    t.set_(buf.untyped_storage(), buf.storage_offset() + c0, (n, d, h, w, c1 - c0), buf.stride())
    return t


def skip_buffer(n, d, h, w, c, cb, device):
    """[N,D,H,W,c+cb] buffer for a skip connection: the encoder block that produces the skip tensor writes it into channels [0, c) (conv(...,
    out=alias_channels(buf, 0, c))), the decoder's transposed conv into [c, c+cb) -- the concatenation of cls_wise_former.py:716-729 costs
    no copy (the full-resolution one was 268 MB read + written per step)."""
    return torch.empty((n, d, h, w, c + cb), dtype=torch.float32, device=device)


def cat_buffer(a, cb):
    """an uninitialised [N,D,H,W,Ca+cb] buffer for cat_into; hand buf[..., Ca:] to the producer of the second operand"""
    return torch.empty(a.shape[:-1] + (a.shape[-1] + cb,), dtype=torch.float32, device=a.device)


def cat_into(a, b, buf):
    if a.data_ptr() == buf.data_ptr() and a.stride(3) == buf.stride(3):      # a already lives in buf (skip_buffer)
        return _CatPrefilledFn.apply(a, b, buf)
    return _CatIntoFn.apply(a, b, buf)


# ======================================================================================================
# token path
# ======================================================================================================
class _WindowToTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, patch):
        ctx.patch, ctx.size, ctx.c = patch, tuple(x.shape[1:4]), x.shape[4]
        return backend().window_to_tokens(x, patch)

    @staticmethod
    def backward(ctx, d):
        return backend().tokens_to_window(d, ctx.size, ctx.c, ctx.patch), None


class _TokensToWindowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, size, channels, patch):
        ctx.patch = patch
        return backend().tokens_to_window(tok, size, channels, patch)

    @staticmethod
    def backward(ctx, d):
        return backend().window_to_tokens(d, ctx.patch), None, None, None


def window_to_tokens(x, patch):
    return _WindowToTokensFn.apply(x, tuple(patch))


def tokens_to_window(tok, size, channels, patch):
    return _TokensToWindowFn.apply(tok, tuple(size), channels, tuple(patch))


def dropout_mask(shape, p, device, p2=0.0):
    """Pre-scaled keep mask, one launch (K12); p2 > 0 folds a second independent dropout of the same tensor in."""
    return backend().dropout_mask(tuple(shape), float(p), device, float(p2))



# ======================================================================================================
# heads and losses
# ======================================================================================================
class _UpsampleSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, c, scale):
        prob = backend().upsample_softmax(logit, c, scale)
        ctx.c, ctx.scale, ctx.lo = c, scale, tuple(logit.shape)
        ctx.save_for_backward(prob)
        return prob

    @staticmethod
    def backward(ctx, dprob):
        (prob,) = ctx.saved_tensors
        n, d, h, w, ca = ctx.lo
        return backend().upsample_softmax_bwd(dprob, prob, (n, d, h, w), ctx.c, ctx.scale, ca), None, None


def upsample_softmax(logit, c, scale):
    return _UpsampleSoftmaxFn.apply(logit, c, scale)


class _ChannelSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit):
        prob = backend().channel_softmax(logit, logit.shape[-1])
        ctx.save_for_backward(prob)
        return prob

    @staticmethod
    def backward(ctx, dprob):
        (prob,) = ctx.saved_tensors
        return backend().channel_softmax_bwd(dprob, prob)


def channel_softmax(logit):
    return _ChannelSoftmaxFn.apply(logit)


class LazyProb:
    """A sub-region probability map that has not been written yet: softmax(trilinear_up(logit)) of a supervision head
    (SuperviseLabel.py:58-81) in TRAINING mode.  The package's own losses (utils.tools.get_separate_loss /
    get_edge_separate_loss) recognise it and compute Dice / CE straight from the low-resolution logits (cwf_head_loss_*): the
    [N,2,D,H,W] map, its gradient and four more full-resolution passes never exist.  Anything else that touches it -- indexing,
    a torch function, a tensor method -- materialises the real map through the ordinary upsample_softmax Function (same values,
    autograd intact), so code written against the reference's tensors keeps working."""

    def __init__(self, logit, c, scale, parent=None):
        self.logit, self.c, self.scale = logit, c, scale
        self.parent = parent                       # (grouped logit buffer [N,d,h,w,G*ca], group index, G, ca): head_group_loss
        self._t = None

    def materialize(self):
        if self._t is None:
            self._t = upsample_softmax(self.logit, self.c, self.scale).permute(0, 4, 1, 2, 3)      # logical [N,C,D,H,W]
        return self._t

    @property
    def shape(self):
        n, d, h, w, _ = self.logit.shape
        s = self.scale
        return torch.Size((n, self.c, d * s, h * s, w * s))

    def __getattr__(self, name):                 # tensor attributes / methods (only reached for names not defined above)
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        real = lambda a: a.materialize() if isinstance(a, LazyProb) else a
        return func(*[real(a) for a in args], **{k: real(v) for k, v in (kwargs or {}).items()})


class _HeadGroupLossFn(torch.autograd.Function):
    """sum over the (up to three) sub-region maps of one supervision call of dice_loss + softmax_weighted_loss, from the
    low-resolution logits (tools.py:112-231 on top of SuperviseLabel.py:62-64)."""

    @staticmethod
    def forward(ctx, label, posmasks, scale, *logits):
        K = backend()
        logits = [t.contiguous() for t in logits]
        total, _, coef = K.head_loss(logits, label, posmasks, scale)
        ctx.posmasks, ctx.scale = posmasks, scale
        ctx.save_for_backward(label, coef, *logits)
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        label, coef = ctx.saved_tensors[:2]
        logits = ctx.saved_tensors[2:]
        gs = g.reshape(1).to(torch.float32).contiguous()
        dls = backend().head_loss_bwd(list(logits), label, ctx.posmasks, ctx.scale, coef, gs)
        return (None, None, None) + tuple(dls)


class _HeadGroupLossGFn(torch.autograd.Function):
    """_HeadGroupLossFn for maps whose logits are the channel groups of ONE buffer (grouped head convs): one tensor in, one gradient
    buffer out (cwf_head_loss_bwd_ex writes each map's group, padding zeroed) -- no per-map slices for autograd to reassemble."""

    @staticmethod
    def forward(ctx, label, posmasks, scale, ca, l_all):
        K = backend()
        G = len(posmasks)
        total, _, coef = K.head_loss([l_all[..., q * ca:(q + 1) * ca] for q in range(G)], label, posmasks, scale)
        ctx.posmasks, ctx.scale, ctx.ca = posmasks, scale, ca
        ctx.save_for_backward(label, coef, l_all)
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        label, coef, l_all = ctx.saved_tensors
        G, ca = len(ctx.posmasks), ctx.ca
        gs = g.reshape(1).to(torch.float32).contiguous()
        d_all = torch.empty_like(l_all)
        backend().head_loss_bwd([l_all[..., q * ca:(q + 1) * ca] for q in range(G)], label, ctx.posmasks, ctx.scale, coef, gs, grouped_out=(d_all, ca))
        return None, None, None, None, d_all


def head_group_loss(lazies, label, posmasks):
    """lazies: LazyProb maps of one supervision call (same shape / scale)."""
    label = label.contiguous()
    if label.dtype != torch.int64:
        label = label.long()
    par = [z.parent for z in lazies]
    if all(p is not None for p in par) and all(p[0] is par[0][0] for p in par) and [p[1] for p in par] == list(range(par[0][2])) \
            and len(lazies) == par[0][2] and par[0][0].shape[-1] == par[0][2] * par[0][3]:
        return _HeadGroupLossGFn.apply(label, tuple(int(m) for m in posmasks), lazies[0].scale, par[0][3], par[0][0])
    return _HeadGroupLossFn.apply(label, tuple(int(m) for m in posmasks), lazies[0].scale, *[z.logit for z in lazies])


class _DiceCeFn(torch.autograd.Function):
    """dice_loss + softmax_weighted_loss of one probability map (tools.py:8-34)."""

    @staticmethod
    def forward(ctx, prob_cl, label, posmask):
        loss, coef = backend().dice_ce(prob_cl, label, posmask)
        ctx.posmask = posmask
        ctx.save_for_backward(prob_cl, label, coef)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        prob_cl, label, coef = ctx.saved_tensors
        gs = g.reshape(1).to(torch.float32).contiguous()
        return backend().dice_ce_bwd(prob_cl, label, ctx.posmask, coef, gs), None, None


def to_channels_last_view(prob_ncdhw):
    """[N,C,D,H,W] (any strides) -> dense [N,D,H,W,C] tensor sharing memory when the input is already channels-last."""
    t = prob_ncdhw.permute(0, 2, 3, 4, 1)
    return t if t.is_contiguous() else t.contiguous()


def dice_ce_loss(prob_ncdhw, label, posmask=0):
    """prob [N,C,D,H,W] (C = 4: class = label; C = 2: class = (posmask >> label) & 1), label int64 [N,D,H,W]."""
    label = label.contiguous()
    if label.dtype != torch.int64:
        label = label.long()
    return _DiceCeFn.apply(to_channels_last_view(prob_ncdhw), label, int(posmask))
