"""Host side of the hand-written MFMA convolution (csrc/conv.hip): NHWC bf16 tensors in, NHWC bf16 out.

Layouts: activations [N, H, W, C] contiguous bfloat16; weights "tap-major" [9, Cout, Cin] bfloat16 (tap = 3*r+s).
`pack_weight` / `pack_weight_dgrad` convert a torch conv weight [Cout, Cin, 3, 3] (models/networks.py, MONAI
DynUNet naming) into the forward layout and into the layout whose forward pass IS the data gradient.
"""
import ctypes
import os

import torch

from .. import _native


def _pad32(c):
    return (c + 31) // 32 * 32


# ---- weight gradients on a side stream ---------------------------------------------------------------------------------
# Weight gradients straight into `weight.grad` (round 5). A layer's weight gradient is a leaf of the backward graph: nothing reads it
# before the optimiser step. Inside `direct_weight_grads()` (the trainers wrap `loss.backward()` in it, models/base_model_abc.py
# backward_scope) the 3x3 weight-gradient launches write their result into the parameter's `.grad` in the parameter's own layout
# (csrc/conv.hip octa_conv3x3_nhwc_wgrad_acc) -- ADDED to an existing gradient (a view of the data-parallel gradient arena, a second use
# of the weight), written over a fresh buffer otherwise -- and the Function reports None to autograd: no [9][Cout][Cin] temporary, no
# fill, no layout copy and no accumulation launch per layer (21 copies + 15 fills per U-Net step, ~3 per convolution of the GAN's
# generator). Outside the context (tests, torch.autograd.grad, user code calling
# .backward() directly) and for parameters without a matching float32 `.grad` the gradient is returned to autograd as before.
# (Measured and removed in round 5: the same scope running the weight-gradient launches on a SIDE STREAM beside the data-gradient / norm
# chain -- U-Net step 19.3 against 19.0 ms at B = 4, 38.7 against 35.5 at B = 8, profiles/r05_stream_overlap_ab.log: every kernel of the
# step fills the GPU on its own, co-resident workgroups only evict each other's lines.)
USE_DIRECT_WGRAD = True         # module switch (tests / tools may clear it): every gradient then returns through autograd
# set by direct_weight_grads(); read by the autograd ENGINE's thread (not the caller's: a thread-local would never be seen there), hence
# process-wide -- and hence one backward scope at a time: a second thread entering while one is open is refused (its backward would get
# None for its weight gradients and find them written into .grad), torch.autograd.grad / .backward() outside a scope are unaffected
_WG_ACTIVE = {"on": False, "owner": None}
_WGRAD_TR = 2                   # the C side runs the transposing-read kernels at stride 1 and 2 (csrc/conv.hip use_tr): only they write into .grad
DIRECT_WGRAD_COUNTS = [0, 0]    # weight gradients accumulated in place / returned to autograd


class direct_weight_grads:
    """with direct_weight_grads(device): loss.backward()"""

    def __init__(self, device):
        self.on = USE_DIRECT_WGRAD and torch.device(device).type == "cuda"

    def __enter__(self):
        import threading
        me = threading.get_ident()
        self.prev = (_WG_ACTIVE["on"], _WG_ACTIVE["owner"])
        if self.on:
            if _WG_ACTIVE["on"] and _WG_ACTIVE["owner"] not in (None, me):
                raise RuntimeError("direct_weight_grads(): another thread is inside a backward scope; the in-place weight gradients need one backward at a time "
                                   "(mfma_conv.USE_DIRECT_WGRAD = False returns every gradient through autograd)")
            _WG_ACTIVE["on"], _WG_ACTIVE["owner"] = True, me
        return self

    def __exit__(self, *exc):
        _WG_ACTIVE["on"], _WG_ACTIVE["owner"] = self.prev
        return False


def _wgrad_to(weight, fn, into=None):
    """dW for `weight`: inside direct_weight_grads() `into(grad, accumulate)` writes the gradient into the parameter's float32 `.grad` in
    place -- added to an existing one, written over a fresh buffer when there is none -- and None goes back to autograd; otherwise
    fn()'s tensor does."""
    if into is not None and _WG_ACTIVE["on"] and weight.is_leaf and weight.is_cuda and weight.dtype == torch.float32:
        g = weight.grad
        if g is None:
            g = torch.empty_like(weight, memory_format=torch.contiguous_format)
            into(g, 0)
            weight.grad = g
            DIRECT_WGRAD_COUNTS[0] += 1
            return None
        if g.dtype == torch.float32 and g.is_contiguous() and g.shape == weight.shape:
            into(g, 1)
            DIRECT_WGRAD_COUNTS[0] += 1
            return None
    DIRECT_WGRAD_COUNTS[1] += 1
    return fn()


def _wgrad_acc(x1, x2, dy, grad, accumulate, stride=1, tap_mask=0x1ff):
    """grad [Cout][Cin][3][3] float32 (+)= weight gradient of the 3x3 layer with input x1 (| x2 on the channel axis) and output gradient dy."""
    n, h, w, c1 = x1.shape
    cin = c1 + (x2.shape[3] if x2 is not None else 0)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = _native.lib().octa_conv3x3_nhwc_wgrad_acc(_native.ctx(x1.device.index), p(x1), p(x2), c1, p(dy), p(grad), n, h, w, cin, dy.shape[3], int(stride),
                                                   int(tap_mask), int(accumulate), _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_wgrad_acc")


USE_PACK_PLAN = True


class WeightPackPlan:
    """All KxK convolution weights of a network packed into both kernel layouts by ONE launch per optimiser step
    (csrc/conv.hip octa_pack_conv_weights) instead of ~7 small torch kernels per layer and step. The plan is
    refreshed when any parameter's version counter (or storage) changed; `pack_weight*` below look parameters up here
    and fall back to the torch formulation for tensors that are not registered. Optimisers, `load_state_dict` and every
    in-place op on the parameter move the counter; writes through `parameter.data` do NOT -- call `invalidate()` after those
    (`networks.init_weights` does)."""

    def __init__(self, convs, convts=()):
        """convs: float32 parameters [Cout,Cin,K,K]; convts: ConvTranspose2d(2, 2) parameters [Cin,Cout,2,2]."""
        self.params = [(w, 0) for w in convs] + [(w, 1) for w in convts]
        assert all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w, _ in self.params)
        self.device = self.params[0][0].device
        off, self.geom = 0, []
        for w, kind in self.params:
            a, b = w.shape[0], w.shape[1]
            kk = 9 if kind == 1 else w.shape[2] * w.shape[3]
            bp = _pad32(b)
            n = kk * a * bp
            self.geom.append((off, off + n, a, b, bp, kk, kind))
            off += 2 * n
        self.buf = torch.empty(off, dtype=torch.bfloat16, device=self.device)
        self.fwd = [self.buf[g[0]:g[0] + g[5] * g[2] * g[4]].view(g[5], g[2], g[4]) for g in self.geom]      # [KK][A][BP]
        self.dg = [self.buf[g[1]:g[1] + g[5] * g[2] * g[4]].view(g[5], g[4], g[2]) for g in self.geom]       # [KK][BP][A]
        self.table, self.ptrs, self.versions = None, None, None
        for i, (w, _) in enumerate(self.params):
            w._octa_pack = (self, i)          # found again by pack_weight*(w); lives as long as the parameter

    def ensure(self):
        ptrs = [w.data_ptr() for w, _ in self.params]
        versions = [w._version for w, _ in self.params]
        if ptrs != self.ptrs:
            # storage moved (.to(), .bfloat16(), memory_format change ...): the packing kernel reads contiguous fp32 [A][B][K][K]
            for w, _ in self.params:
                if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                    raise RuntimeError("MFMA weight packing needs contiguous float32 master weights on the GPU; got "
                                       f"{w.dtype}, contiguous={w.is_contiguous()}, device={w.device} (convert activations, not parameters)")
            rows = [[p, g[0], g[1], g[2], g[3], g[4], g[5], g[6]] for p, g in zip(ptrs, self.geom)]
            self.table = torch.tensor(rows, dtype=torch.int64).to(self.device)
            self.ptrs, self.versions = ptrs, None
        if versions != self.versions:
            rc = _native.lib().octa_pack_conv_weights(_native.ctx(self.device.index), ctypes.c_void_p(self.table.data_ptr()), len(self.params),
                                                      ctypes.c_void_p(self.buf.data_ptr()), _native.current_stream_ptr())
            _native.check(rc, "octa_pack_conv_weights")
            self.versions = versions

    def invalidate(self):
        """After writes that bypass the version counter (`.data`)."""
        self.versions = None


def invalidate_all_pack_plans(module):
    """Every cached WeightPackPlan below `module` repacks on its next use: call after writes that do not move the
    parameters' version counters (`.data` writes: dist.broadcast(p.data), EMA, checkpoint surgery)."""
    for m in module.modules():
        plan = getattr(m, "_octa_pack_plan", None)
        if plan is not None:
            plan.invalidate()
    from . import conv_f32
    conv_f32.invalidate_packs()


def plan_for_module(module):
    """One WeightPackPlan over every MFMA-shaped convolution weight of `module` (3x3 / 4x4 Conv2d, 2x2 ConvTranspose2d),
    built on first use and cached on the module; None when nothing qualifies (CPU, non-float32 master weights)."""
    plan = getattr(module, "_octa_pack_plan", None)
    if plan is not None or not USE_PACK_PLAN:
        return plan
    ok = lambda w: w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    convs = [m.weight for m in module.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size in ((3, 3), (4, 4)) and m.groups == 1
             and ok(m.weight)]
    convts = [m.weight for m in module.modules() if isinstance(m, torch.nn.ConvTranspose2d) and m.kernel_size == (2, 2) and m.stride == (2, 2)
              and m.groups == 1 and ok(m.weight)]
    if not convs and not convts:
        return None
    plan = WeightPackPlan(convs, convts)
    object.__setattr__(module, "_octa_pack_plan", plan)
    return plan


def _planned(w, kind):
    if not USE_PACK_PLAN:
        return None, None
    ent = getattr(w, "_octa_pack", None)
    if ent is None or ent[0].params[ent[1]][0] is not w or ent[0].params[ent[1]][1] != kind:
        return None, None
    ent[0].ensure()
    return ent


def slice_major(wt):
    """Tap-major packed weights [KK][Cout][CinP] -> the storage order the kernels read, [CinP/16][KK][Cout][16] (csrc/conv.hip wt_off():
    the 16 input channels of one MFMA K-step are contiguous for all taps and output channels, so a workgroup's weight slice is whole
    cache lines). The nominal shape stays [KK][Cout][CinP] -- the tensor is only ever handed to the kernels."""
    kk, co, cp = wt.shape
    assert cp % 16 == 0, "packed weights: input channels must be a multiple of 16"
    return wt.reshape(kk, co, cp // 16, 16).permute(2, 0, 1, 3).contiguous().view(kk, co, cp)


def tap_major(wt):
    """Inverse of slice_major(): the nominal [KK][Cout][CinP] view of packed weights (tests, debugging)."""
    kk, co, cp = wt.shape
    return wt.reshape(cp // 16, kk, co, 16).permute(1, 2, 0, 3).reshape(kk, co, cp)


def pack_weight(w, cin_pad=None):
    """[Cout, Cin, K, K] -> [K*K, Cout, CinP] bf16 (CinP = cin_pad or Cin; padded columns zero), stored slice-major (slice_major())."""
    co, ci, kk = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    cp = ci if cin_pad is None else cin_pad
    plan, i = _planned(w, 0)
    if plan is not None and plan.geom[i][4] == cp:
        return plan.fwd[i]
    if cp != ci:
        wp = w.new_zeros((co, cp) + tuple(w.shape[2:]))
        wp[:, :ci] = w
        w = wp
    return slice_major(w.permute(2, 3, 0, 1).reshape(kk, co, cp).to(torch.bfloat16))


def pack_weight_dgrad(w, cin_pad=None):
    """[Cout, Cin, K, K] -> [K*K, CinP, Cout] bf16 with the taps flipped: conv(dy, this) = dL/dx."""
    co, ci, kk = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    cp = ci if cin_pad is None else cin_pad
    plan, i = _planned(w, 0)
    if plan is not None and plan.geom[i][4] == cp:
        return plan.dg[i]
    if cp != ci:
        wp = w.new_zeros((co, cp) + tuple(w.shape[2:]))
        wp[:, :ci] = w
        w = wp
    return slice_major(w.flip(2, 3).permute(2, 3, 1, 0).reshape(kk, cp, co).to(torch.bfloat16))


def pack_convt2x2(w):
    """ConvTranspose2d(2, 2) weight [Cin, Cout, 2, 2] as the 3x3 kernel wc[a][b][r][s] = w[a][b][r-1][s-1] (zero at r = 0 or
    s = 0), in both layouts: (pack_weight(wc) [9, Cin, Cout], pack_weight_dgrad(wc) [9, Cout, Cin])."""
    plan, i = _planned(w, 1)
    if plan is not None and plan.geom[i][4] == w.shape[1]:
        return plan.fwd[i], plan.dg[i]
    wc = w.new_zeros((w.shape[0], w.shape[1], 3, 3))
    wc[:, :, 1:, 1:] = w
    return (slice_major(wc.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).to(torch.bfloat16)),
            slice_major(wc.flip(2, 3).permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]).to(torch.bfloat16)))


def conv3x3_nhwc(x, wt, stride=1, in_dilation=1, tap_mask=0x1ff, residual=None):
    """x [N,H,W,Cin] bf16, wt [9,Cout,Cin] bf16 -> [N,Ho,Wo,Cout] bf16 (padding 1). tap_mask bit 3r+s = evaluate tap
    (r, s) of wt (as packed); cleared taps must have zero weights. residual (shape of the result, bf16) is added in the
    kernel's epilogue (octa_conv3x3_nhwc_fwd6)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4
    assert wt.dtype == torch.bfloat16 and wt.is_contiguous() and wt.shape[0] == 9 and wt.shape[2] == x.shape[3]
    n, h, w, cin = x.shape
    cout = wt.shape[1]
    ho = (h * in_dilation - 1) // stride + 1
    wo = (w * in_dilation - 1) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == torch.bfloat16 and residual.is_contiguous()
        rc = _native.lib().octa_conv3x3_nhwc_fwd6(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), None, cin,
                                                  ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(y.data_ptr()), None, cout, n, h, w, cin, cout,
                                                  int(stride), int(in_dilation), int(tap_mask), 1, 0, 0, None, None, None, None, 0.0, None,
                                                  ctypes.c_void_p(residual.data_ptr()), _native.current_stream_ptr())
        _native.check(rc, "octa_conv3x3_nhwc_fwd6")
        return y
    rc = _native.lib().octa_conv3x3_nhwc_fwd2(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), None, cin,
                                              ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(y.data_ptr()), None, cout, n, h, w, cin, cout,
                                              int(stride), int(in_dilation), int(tap_mask), _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_fwd2")
    return y


# The parity-fused kernel for the two layers that double the image (round 5, csrc/conv.hip conv3x3_s2t_kernel). OCTA_S2T=0 keeps the
# zero-insertion form of rounds 1-4 (the stride-1 kernel on a virtually dilated input: 4 x the multiply-adds).
USE_S2T = True


def conv3x3_s2t_nhwc(x, wt, tap_mask=0x1ff, residual=None):
    """x [N,H,W,Cin] bf16 (the small image), wt [9,Cout,Cin] bf16 as conv3x3_nhwc(..., in_dilation=2) takes it -> [N,2H,2W,Cout] bf16."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and wt.dtype == torch.bfloat16 and wt.is_contiguous() and wt.shape[2] == x.shape[3]
    n, h, w, cin = x.shape
    cout = wt.shape[1]
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == torch.bfloat16 and residual.is_contiguous()
    rc = _native.lib().octa_conv3x3_s2t_nhwc(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wt.data_ptr()),
                                             ctypes.c_void_p(y.data_ptr()), n, h, w, cin, cout, int(tap_mask),
                                             ctypes.c_void_p(residual.data_ptr()) if residual is not None else None, _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_s2t_nhwc")
    return y


def conv3x3_nhwc_wgrad(x, dy, tap_mask=0x1ff):
    """x [N,H,W,Cin] bf16, dy [N,H,W,Cout] bf16 (stride-1 layer) -> dW as a torch conv weight gradient
    [Cout, Cin, 3, 3] float32 (taps cleared in tap_mask come back as zero)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and dy.dtype == torch.bfloat16 and dy.is_contiguous()
    n, h, w, cin = x.shape
    assert dy.shape[:3] == x.shape[:3]
    cout = dy.shape[3]
    dw = torch.empty((9, cout, cin), dtype=torch.float32, device=x.device)
    rc = _native.lib().octa_conv3x3_nhwc_wgrad2(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), None, cin,
                                                ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(dw.data_ptr()), n, h, w, cin, cout, int(tap_mask),
                                                _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_wgrad2")
    return dw.view(3, 3, cout, cin).permute(2, 3, 0, 1)


STAT_SLOTS = 16      # slots the tiles of an image spread their statistics atomics over
_STAT_RINGS = {}
_STAT_LOCK = __import__("threading").Lock()                  # two threads running forwards on one stream must not be handed overlapping slices


def _stat_slots(device, n, cout):
    """Zeroed double[STAT_SLOTS][n][cout][2] from a per-(device, stream) ring: ONE fill per lap instead of one per layer. A slice is
    consumed by the two launches enqueued right behind it (convolution, norm apply) on the same stream; the lap's fill is enqueued on
    that stream too, so it is ordered behind every earlier consumer."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    need = STAT_SLOTS * n * cout * 2
    with _STAT_LOCK:
        ring = _STAT_RINGS.get(key)
        if ring is None or ring[0].numel() < 8 * need:
            ring = [torch.zeros(max(4 << 20, 8 * need), dtype=torch.float64, device=device), 0]
            _STAT_RINGS[key] = ring
        if ring[1] + need > ring[0].numel():
            ring[0].zero_()
            ring[1] = 0
        t = ring[0][ring[1]:ring[1] + need].view(STAT_SLOTS, n, cout, 2)
        ring[1] += need
    return t


def _conv_fwd_stats(x1, x2, wt, stride, want_stats):
    """Forward launch over one or two inputs (virtual concatenation); optionally also the InstanceNorm statistics of the result,
    accumulated in the kernel's epilogue: want_stats = True / "slots" -> double [STAT_SLOTS][N][Cout][2] (round 5: octa_conv3x3_nhwc_fwd7),
    "tiles" -> the round-1 per-tile partials float32 [N][tiles][Cout][2] (octa_conv3x3_nhwc_fwd5; kept for its tests)."""
    n, h, w, c1 = x1.shape
    c2 = x2.shape[3] if x2 is not None else 0
    cout = wt.shape[1]
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=x1.device)
    part = None
    if want_stats and want_stats != "tiles":
        part = _stat_slots(x1.device, n, cout)
        pp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        rc = _native.lib().octa_conv3x3_nhwc_fwd7(_native.ctx(x1.device.index), pp(x1), pp(x2), c1, pp(wt), pp(y), n, h, w, c1 + c2, cout, int(stride),
                                                  pp(part), STAT_SLOTS, _native.current_stream_ptr())
        _native.check(rc, "octa_conv3x3_nhwc_fwd7")
        return y, part
    if want_stats:
        tiles = _native.lib().octa_conv_stat_tiles(ho, wo)
        part = torch.empty((n, tiles, cout, 2), dtype=torch.float32, device=x1.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = _native.lib().octa_conv3x3_nhwc_fwd5(_native.ctx(x1.device.index), p(x1), p(x2), c1, p(wt), p(y), None, cout, n, h, w, c1 + c2, cout,
                                              int(stride), 1, 0x1ff, 1, 0, 0, None, None, None, None, 0.0, p(part), _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_fwd5")
    return y, part


# ---- zero-insertion-free stride-2 data gradient / 2x2 transposed convolution: one scattered launch per parity class

# dX[2h+a] receives dY[h + off] * W[r] for (r, off) in: a = 0 -> (1, 0); a = 1 -> (2, 0), (0, +1). The kernel's tap kr reads
# offset kr - 1, so off 0 <-> kr 1 and off +1 <-> kr 2.
_PARITY_TAPS = {0: ((1, 1),), 1: ((2, 1), (0, 2))}   # parity -> ((conv tap r, kernel tap kr), ...)
_parity_tables = {}


def _parity_table(device):
    """src[p][t] = flat conv tap 3r+s feeding kernel tap t of parity class p = 2a+b (or 0 with valid 0), masks[p]."""
    key = str(device)
    if key not in _parity_tables:
        src = torch.zeros((4, 9), dtype=torch.long)
        valid = torch.zeros((4, 9), dtype=torch.bfloat16)
        masks = []
        for a in (0, 1):
            for b in (0, 1):
                m = 0
                for r, kr in _PARITY_TAPS[a]:
                    for s_, ks in _PARITY_TAPS[b]:
                        src[2 * a + b, 3 * kr + ks] = 3 * r + s_
                        valid[2 * a + b, 3 * kr + ks] = 1
                        m |= 1 << (3 * kr + ks)
                masks.append(m)
        _parity_tables[key] = (src.to(device), valid.to(device), masks)
    return _parity_tables[key]


def _fwd3(x, wt, y, cin, cout, mask, a, b):
    n, h, w, _ = x.shape
    rc = _native.lib().octa_conv3x3_nhwc_fwd3(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), None, cin, ctypes.c_void_p(wt.data_ptr()),
                                              ctypes.c_void_p(y.data_ptr()), None, cout, n, h, w, cin, cout, 1, 1, int(mask), 2, int(a), int(b),
                                              _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_fwd3")


def conv3x3_s2_dgrad(dy, weight):
    """Data gradient of a stride-2, padding-1 3x3 layer (weight [Cout, Cin, 3, 3]): dy [N,Ho,Wo,Cout] -> [N,2Ho,2Wo,Cin].
    Four scattered launches (one per output parity) over the taps that reach that parity: no inserted zeros."""
    n, ho, wo, cout = dy.shape
    cin = weight.shape[1]
    src, valid, masks = _parity_table(dy.device)
    w9 = weight.to(torch.bfloat16).permute(2, 3, 1, 0).reshape(9, cin, cout)          # [tap][ci][co]
    wp = (w9[src] * valid[:, :, None, None]).contiguous()                              # [4][9][ci][co]
    dx = torch.empty((n, 2 * ho, 2 * wo, cin), dtype=torch.bfloat16, device=dy.device)
    for p in range(4):
        _fwd3(dy, slice_major(wp[p]), dx, cout, cin, masks[p], p >> 1, p & 1)
    return dx


def conv_transpose_2x2_fwd(x, weight):
    """ConvTranspose2d(k = s = 2) forward, weight [Cin, Cout, 2, 2]: per output parity a 1x1 convolution (kernel tap 4)."""
    n, h, w, cin = x.shape
    cout = weight.shape[1]
    wp = x.new_zeros((4, 9, cout, cin))
    wp[:, 4] = weight.to(torch.bfloat16).permute(2, 3, 1, 0).reshape(4, cout, cin)
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=torch.bfloat16, device=x.device)
    for p in range(4):
        _fwd3(x, slice_major(wp[p]), y, cin, cout, 1 << 4, p >> 1, p & 1)
    return y


# Measured on MI355X (B=4, 1216^2 U-Net step), twice: with the register-staged kernels (32.6 vs 32.1 ms) and again with the
# DMA-staged ones (stride-2 data gradient 22.0 vs 21.8 ms, transposed convolution 22.15 vs 21.8 ms in the same session): the
# four scattered launches are no faster than one launch over the zero-inserted input -- four passes over the input, four
# weight repacks and 64/128-byte scattered stores cost what the skipped multiplications save. The zero-insertion form stays.
USE_PARITY_SCATTER = False          # stride-2 data gradient as four parity classes
USE_PARITY_SCATTER_CONVT = False    # 2x2 transposed convolution forward as four 1x1 parity classes


# ---- autograd bindings ---------------------------------------------------------------------------------------------

def _pad_channels(x, mult=32):
    c = x.shape[-1]
    if c % mult == 0:
        return x
    out = x.new_zeros(x.shape[:-1] + ((c + mult - 1) // mult * mult,))
    out[..., :c] = x
    return out


# A/B switch (development aid). The residual epilogue exists in the DMA-staged kernel only: with OCTA_CONV_GLDS=0 (the
# register-staged kernel) the mailbox is never armed and autograd adds the two skip gradients itself.
USE_SKIP_GRAD_FUSION = True


class SkipGradMailbox:
    """A tensor with two consumers in the U-Net (a skip connection: the next encoder block and the decoder's concatenation)
    receives two gradients that autograd would add in a separate pass over the tensor (`add<bf16>`: 0.32 ms of the step for the
    four skips). The decoder's gradient is always computed first (everything the encoder block feeds lies between them), so the
    decoder's backward POSTS its gradient here and reports None to autograd, and the encoder convolution's backward adds it in the
    epilogue of its data-gradient launch. One mailbox per skip tensor and forward pass (models/networks.py builds them)."""
    __slots__ = ("pending", "armed")

    def __init__(self):
        self.pending = None
        self.armed = False       # set by the encoder convolution's forward when its backward will collect the posting


class _Conv3x3NHWC(torch.autograd.Function):
    """y = conv3x3(x, weight), padding 1, stride 1 or 2. x [N,H,W,Cin] bf16 (Cin padded to 32 internally),
    weight [Cout,Cin,3,3] (any float dtype; master copy), y [N,Ho,Wo,Cout] bf16."""

    @staticmethod
    def forward(ctx, x, weight, stride, want_stats=False, mailbox=None):
        cin = weight.shape[1]
        ctx.mailbox = None
        ctx.set_materialize_grads(False)      # the statistics output is not differentiable: no zero "gradient" of it (a fill launch per layer)
        if mailbox is not None and USE_SKIP_GRAD_FUSION and x.requires_grad and x.shape[-1] % 32 == 0 and not (int(stride) == 2 and USE_PARITY_SCATTER):
            ctx.mailbox = mailbox
            mailbox.armed = True
        xp = _pad_channels(x.contiguous())
        y, part = _conv_fwd_stats(xp, None, pack_weight(weight, xp.shape[-1]), stride, want_stats)
        ctx.save_for_backward(xp, weight)
        ctx.stride, ctx.cin = int(stride), cin
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        xp, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None, None, None
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        cin, st = ctx.cin, ctx.stride
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if st == 2:
                assert xp.shape[1] % 2 == 0 and xp.shape[2] % 2 == 0, "stride-2 layers need even input sizes"
            # channel-padded input: the GAN trains the generator THROUGH the segmentor's first layer
            if st == 2 and USE_PARITY_SCATTER:
                wd = weight
                if xp.shape[-1] != cin:
                    wd = weight.new_zeros((weight.shape[0], xp.shape[-1], 3, 3))
                    wd[:, :cin] = weight
                dx = conv3x3_s2_dgrad(dy, wd)
            else:
                res = None
                if ctx.mailbox is not None:
                    res, ctx.mailbox.pending = ctx.mailbox.pending, None       # the decoder's gradient of the same tensor, if posted
                if st == 2 and USE_S2T:
                    dx = conv3x3_s2t_nhwc(dy, pack_weight_dgrad(weight, xp.shape[-1]), 0x1ff, res)
                else:
                    dx = conv3x3_nhwc(dy, pack_weight_dgrad(weight, xp.shape[-1]), stride=1, in_dilation=st, residual=res)
            if xp.shape[-1] != cin:
                dx = dx[..., :cin].contiguous()
        elif ctx.mailbox is not None and ctx.mailbox.pending is not None:
            raise RuntimeError("skip gradient posted but the encoder convolution computes no input gradient")
        if ctx.needs_input_grad[1]:
            direct = (lambda g, acc: _wgrad_acc(xp, None, dy, g, acc, st)) if (xp.shape[-1] == cin and _WGRAD_TR >= st and (st == 1 or (xp.shape[1] % 2 == 0 and xp.shape[2] % 2 == 0))) else None
            if st == 1:
                dw = _wgrad_to(weight, lambda: conv3x3_nhwc_wgrad(xp, dy)[:, :cin].to(weight.dtype), direct)
            else:
                dw = _wgrad_to(weight, lambda: _s2_wgrad(xp, dy)[:, :cin].to(weight.dtype), direct)
        return dx, dw, None, None, None


class _Conv3x3ReflectNHWC(torch.autograd.Function):
    """nn.ReflectionPad2d(1) + Conv2d(3, padding 0) (the ResNet blocks of the reference's generator, models/networks.py:151-176) in ONE
    launch per pass: the forward and the weight-gradient kernels mirror the image in their halo fetch (no padded tensor, no crop);
    the data gradient is the full (pad 2) convolution of dy with the flipped weights, folded back by the reflection's adjoint."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        x = x.contiguous()
        n, h, w, cin = x.shape
        cout = weight.shape[0]
        ctx.set_materialize_grads(False)
        y = torch.empty((n, h, w, cout), dtype=torch.bfloat16, device=x.device)
        wt = pack_weight(weight, cin)
        part = _stat_slots(x.device, n, cout) if want_stats else None       # InstanceNorm statistics from the epilogue (slot form, as conv3x3)
        rc = _native.lib().octa_conv3x3_nhwc_fwd_pad_s(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wt.data_ptr()),
                                                       ctypes.c_void_p(y.data_ptr()), n, h, w, cin, cout, 1, 1,
                                                       ctypes.c_void_p(part.data_ptr()) if part is not None else None, STAT_SLOTS if part is not None else 0,
                                                       _native.current_stream_ptr())
        _native.check(rc, "octa_conv3x3_nhwc_fwd_pad_s")
        ctx.save_for_backward(x, weight)
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        from . import resample
        x, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        n, h, w, cin = x.shape
        cout = dy.shape[3]
        lib, hctx = _native.lib(), _native.ctx(x.device.index)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wd = pack_weight_dgrad(weight, cin)
            dxp = torch.empty((n, h + 2, w + 2, cin), dtype=torch.bfloat16, device=x.device)
            rc = lib.octa_conv3x3_nhwc_fwd_pad(hctx, ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(wd.data_ptr()), ctypes.c_void_p(dxp.data_ptr()),
                                               n, h, w, cout, cin, 2, 0, _native.current_stream_ptr())
            _native.check(rc, "octa_conv3x3_nhwc_fwd_pad")
            dx = torch.empty_like(x)
            resample._launch("octa_reflect_pad_bwd", dxp, dx, n, h, w, cin, 1)
        if ctx.needs_input_grad[1]:
            def wg():
                dwf = torch.empty((9, cout, cin), dtype=torch.float32, device=x.device)
                rc = lib.octa_conv3x3_nhwc_wgrad_pad(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                                                     ctypes.c_void_p(dwf.data_ptr()), n, h, w, cin, cout, 1, 1, _native.current_stream_ptr())
                _native.check(rc, "octa_conv3x3_nhwc_wgrad_pad")
                return dwf.view(3, 3, cout, cin).permute(2, 3, 0, 1).to(weight.dtype)

            def direct(g, acc):
                rc = lib.octa_conv3x3_nhwc_wgrad_pad_acc(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                                                         ctypes.c_void_p(g.data_ptr()), n, h, w, cin, cout, 1, 1, int(acc), _native.current_stream_ptr())
                _native.check(rc, "octa_conv3x3_nhwc_wgrad_pad_acc")
            dw = _wgrad_to(weight, wg, direct)
        return dx, dw, None


USE_FUSED_REFLECT = True     # module switch (development aid)


def conv3x3_reflect(x, weight, want_stats=False):
    """ReflectionPad2d(1) + 3x3 convolution without padding; x [N,H,W,Cin] bf16 with Cin, Cout multiples of 32. want_stats: returns
    (y, statistics slots for instance_norm_leaky_relu_nhwc(..., partials=)) -- None for the slots where the fused kernel does not apply."""
    if USE_FUSED_REFLECT and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 32 == 0 and weight.shape[0] % 32 == 0 and weight.shape[1] == x.shape[-1]:
        return _Conv3x3ReflectNHWC.apply(x, weight, bool(want_stats))
    from . import resample
    y = conv3x3(resample.reflect_pad(x, 1, "nhwc"), weight, 1)[:, 1:-1, 1:-1, :]
    return (y, None) if want_stats else y


USE_C1_DGRAD = True      # module switch: the one-channel layer's streaming data gradient (round 5)


class _Conv3x3C1(torch.autograd.Function):
    """First layer: one input channel. want_stats: also the InstanceNorm statistics of the result in slot form (double
    [STAT_SLOTS][N][Cout][2], accumulated by the kernel's epilogue: the norm needs no statistics pass). The data gradient -- needed when
    the image is itself a network's output: the segmentor behind the generator in the GAN-seg step -- is the C -> 1 streaming kernel of
    csrc/thin_conv.hip with the flipped taps (round 5; until then such an input took the matrix-core kernels with the image zero-padded to
    32 channels: a 757 MB fill + copy in front of the layer and a 32-channel data gradient + slice copy behind it)."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        x = x.contiguous()
        n, h, w, _ = x.shape
        cout = weight.shape[0]
        ctx.set_materialize_grads(False)
        wf = weight.reshape(cout, 9).float().contiguous()
        y = torch.empty((n, h, w, cout), dtype=torch.bfloat16, device=x.device)
        part = _stat_slots(x.device, n, cout) if want_stats else None
        rc = _native.lib().octa_conv3x3_c1_fwd2(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wf.data_ptr()),
                                                ctypes.c_void_p(y.data_ptr()), n, h, w, cout,
                                                ctypes.c_void_p(part.data_ptr()) if part is not None else None, STAT_SLOTS if part is not None else 0,
                                                _native.current_stream_ptr())
        _native.check(rc, "octa_conv3x3_c1_fwd2")
        ctx.save_for_backward(x)
        ctx.w_shape, ctx.w_dtype = weight.shape, weight.dtype
        ctx.weight_ref = weight
        ctx.wf = wf if ctx.needs_input_grad[0] else None        # the taps as the forward used them (fp32 [Cout][9]): the data gradient's weights
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        (x,) = ctx.saved_tensors
        if dy is None:
            return None, None, None
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        n, h, w, _ = x.shape
        cout = dy.shape[3]
        def wg():
            dw = torch.empty((cout, 9), dtype=torch.float32, device=x.device)
            rc = _native.lib().octa_conv3x3_c1_wgrad(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                                                     ctypes.c_void_p(dw.data_ptr()), n, h, w, cout, _native.current_stream_ptr())
            _native.check(rc, "octa_conv3x3_c1_wgrad")
            return dw.view(ctx.w_shape).to(ctx.w_dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            # dx[n][y][x] = sum_co sum_t dy[n][y + 1 - ky][x + 1 - kx][co] * w[co][t]: the C -> 1 convolution of dy with the flipped kernel, pad K - 1 - 1
            dx = torch.empty((n, h, w, 1), dtype=torch.bfloat16, device=x.device)
            rc = _native.lib().octa_thinconv_squeeze(_native.ctx(x.device.index), ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(ctx.wf.data_ptr()), None,
                                                     ctypes.c_void_p(dx.data_ptr()), n, h, w, cout, 3, 1, 1, _native.current_stream_ptr())
            _native.check(rc, "octa_thinconv_squeeze")
        return dx, _wgrad_to(ctx.weight_ref, wg) if ctx.needs_input_grad[1] else None, None


def conv3x3(x, weight, stride=1, want_stats=False, mailbox=None):
    """want_stats: also return the statistics for instance_norm_leaky_relu_nhwc(..., partials=)."""
    if (x.shape[-1] == 1 and weight.shape[1] == 1 and stride == 1 and weight.shape[0] in (8, 16, 32, 64)
            and x.shape[2] <= 3840 and (USE_C1_DGRAD or not x.requires_grad)):
        if want_stats == "tiles":                        # the per-tile form belongs to the MFMA kernel: the norm runs its own statistics pass
            return _Conv3x3C1.apply(x, weight, False), None
        return _Conv3x3C1.apply(x, weight, bool(want_stats))          # streaming first-layer kernels
    return _Conv3x3NHWC.apply(x, weight, stride, want_stats, mailbox)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class _Conv3x3CatNHWC(torch.autograd.Function):
    """y = conv3x3(cat(x1, x2, channel axis), weight), stride 1, without materialising the concatenation: the
    kernels read the two tensors as one virtual input and the data gradient is written straight into two tensors
    (MONAI UnetUpBlock: conv_block(torch.cat((transp_conv(x), skip), 1)))."""

    @staticmethod
    def forward(ctx, x1, x2, weight, want_stats=False, mailbox=None):
        ctx.set_materialize_grads(False)
        ctx.mailbox = mailbox if (mailbox is not None and mailbox.armed) else None
        x1, x2 = x1.contiguous(), x2.contiguous()
        c1, c2 = x1.shape[3], x2.shape[3]
        assert x2.shape[:3] == x1.shape[:3] and weight.shape[1] == c1 + c2 and c1 % 32 == 0 and c2 % 32 == 0
        y, part = _conv_fwd_stats(x1, x2, pack_weight(weight), 1, want_stats)
        ctx.save_for_backward(x1, x2, weight)
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        x1, x2, weight = ctx.saved_tensors
        if dy is None:
            return None, None, None, None, None
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        n, h, w, c1 = x1.shape
        c2, cout = x2.shape[3], weight.shape[0]
        lib, hctx, st = _native.lib(), _native.ctx(x1.device.index), _native.current_stream_ptr()
        dx1 = dx2 = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
            wd = pack_weight_dgrad(weight)
            rc = lib.octa_conv3x3_nhwc_fwd2(hctx, _p(dy), None, cout, _p(wd), _p(dx1), _p(dx2), c1, n, h, w, cout, c1 + c2, 1, 1, 0x1ff, st)
            _native.check(rc, "octa_conv3x3_nhwc_fwd2 (data gradient)")
        if ctx.needs_input_grad[2]:
            def wg():
                dwf = torch.empty((9, cout, c1 + c2), dtype=torch.float32, device=x1.device)
                rc = lib.octa_conv3x3_nhwc_wgrad2(_native.ctx(x1.device.index), _p(x1), _p(x2), c1, _p(dy), _p(dwf), n, h, w, c1 + c2, cout, 0x1ff,
                                                  _native.current_stream_ptr())
                _native.check(rc, "octa_conv3x3_nhwc_wgrad2")
                return dwf.view(3, 3, cout, c1 + c2).permute(2, 3, 0, 1).to(weight.dtype)
            dw = _wgrad_to(weight, wg, (lambda g, acc: _wgrad_acc(x1, x2, dy, g, acc)) if _WGRAD_TR >= 1 else None)
        if ctx.mailbox is not None and dx2 is not None and ctx.needs_input_grad[1]:
            assert ctx.mailbox.pending is None
            ctx.mailbox.pending, dx2 = dx2, None          # collected by the encoder convolution's data-gradient epilogue
        return dx1, dx2, dw, None, None


def conv3x3_cat(x1, x2, weight, want_stats=False, mailbox=None):
    return _Conv3x3CatNHWC.apply(x1, x2, weight, want_stats, mailbox)


class _InstNormLReLUNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, slope, eps, partials=None):
        x = x.contiguous()
        assert x.dtype == torch.bfloat16 and x.dim() == 4
        B, C = x.shape[0], x.shape[3]
        hw = x.shape[1] * x.shape[2]
        y = torch.empty_like(x)
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        w = weight.float().contiguous() if weight is not None else None
        b = bias.float().contiguous() if bias is not None else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if partials is not None and partials.dtype == torch.float64:       # slot form (octa_conv3x3_nhwc_fwd7)
            rc = _native.lib().octa_instnorm_lrelu_nhwc_fwd_s(_native.ctx(x.device.index), p(x), p(y), p(w), p(b), p(mean), p(rstd), B, C, hw,
                                                              float(slope), float(eps), p(partials), int(partials.shape[0]),
                                                              _native.current_stream_ptr())
        elif partials is not None:
            rc = _native.lib().octa_instnorm_lrelu_nhwc_fwd_p(_native.ctx(x.device.index), p(x), p(y), p(w), p(b), p(mean), p(rstd), B, C, hw,
                                                              float(slope), float(eps), p(partials), int(partials.shape[1]),
                                                              _native.current_stream_ptr())
        else:
            rc = _native.lib().octa_instnorm_lrelu_nhwc_fwd(_native.ctx(x.device.index), p(x), p(y), p(w), p(b), p(mean), p(rstd), B, C, hw,
                                                            float(slope), float(eps), _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_lrelu_nhwc_fwd")
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.slope, ctx.has_w, ctx.has_b = float(slope), weight is not None, bias is not None
        ctx.w_dtype = weight.dtype if weight is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        B, C = x.shape[0], x.shape[3]
        hw = x.shape[1] * x.shape[2]
        dx = torch.empty_like(x)
        if ctx.has_w and ctx.has_b:
            dwb = torch.empty(2 * C, dtype=torch.float32, device=x.device)     # back to back: the C side clears both with one fill
            dw, db = dwb[:C], dwb[C:]
        else:
            dw = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_w else None
            db = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_b else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        rc = _native.lib().octa_instnorm_lrelu_nhwc_bwd(_native.ctx(x.device.index), p(x), p(dy), p(w), p(b), p(mean), p(rstd), p(dx), p(dw),
                                                        p(db), B, C, hw, ctx.slope, _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_lrelu_nhwc_bwd")
        return dx, (dw.to(ctx.w_dtype) if dw is not None else None), (db.to(ctx.w_dtype) if db is not None else None), None, None, None


def instance_norm_leaky_relu_nhwc(x, weight, bias, negative_slope=0.01, eps=1e-5, partials=None):
    """partials: per-tile statistics produced by the convolution that wrote x (conv3x3(..., want_stats=True))."""
    return _InstNormLReLUNHWC.apply(x, weight, bias, negative_slope, eps, partials)


def _s2_wgrad(x_big, dy_small, taps2=((0, 1, 2), (0, 1, 2))):
    """Weight gradient [Cout, Cin, 3, 3] (float32) of a stride-2, padding-1 3x3 conv with input x_big [N,2H,2W,Cin] and
    output gradient dy_small [N,H,W,Cout]; taps2 = the (r, s) taps that are wanted (others come back zero)."""
    x_big, dy_small = x_big.contiguous(), dy_small.contiguous()
    n, h, w, cin = x_big.shape
    cout = dy_small.shape[-1]
    mask = 0
    for r in taps2[0]:
        for s_ in taps2[1]:
            mask |= 1 << (3 * r + s_)
    dw = torch.empty((9, cout, cin), dtype=torch.float32, device=x_big.device)
    rc = _native.lib().octa_conv3x3_nhwc_wgrad4(_native.ctx(x_big.device.index), ctypes.c_void_p(x_big.data_ptr()), None, cin,
                                                ctypes.c_void_p(dy_small.data_ptr()), ctypes.c_void_p(dw.data_ptr()), n, h, w, cin, cout, 2, mask,
                                                None, None, None, None, 0.0, _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_wgrad4")
    return dw.view(3, 3, cout, cin).permute(2, 3, 0, 1)


class _ConvT2x2NHWC(torch.autograd.Function):
    """ConvTranspose2d(kernel 2, stride 2, no bias) as the ADJOINT of a stride-2 3x3 convolution whose taps r, s = 0
    are zero: forward = that convolution's data-gradient kernel (virtual zero insertion), backward = its forward
    kernel (stride 2) and its weight gradient -- all on the MFMA kernels, no GEMM + pixel-shuffle round trip."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        if USE_PARITY_SCATTER_CONVT:
            y = conv_transpose_2x2_fwd(x, weight)
        else:
            # the 3x3 form wc has taps r, s in {1, 2}; flipped for the data-gradient form they sit at r, s in {0, 1}
            if USE_S2T:
                y = conv3x3_s2t_nhwc(x, pack_convt2x2(weight)[1], 0b000011011)
            else:
                y = conv3x3_nhwc(x, pack_convt2x2(weight)[1], stride=1, in_dilation=2, tap_mask=0b000011011)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_nhwc(dy, pack_convt2x2(weight)[0], stride=2, tap_mask=0b110110000)
        if ctx.needs_input_grad[1]:
            dw = _wgrad_to(weight, lambda: _s2_wgrad(dy, x, taps2=((1, 2), (1, 2)))[:, :, 1:, 1:].to(weight.dtype))
        return dx, dw


# The 1x1 "transposed" layer at the U-Net's bottleneck (512 -> 256 at 152^2) runs the repository's own MFMA convolution with only the centre
# tap unmasked: no vendor kernel is left in the training step. (Round 3's route through hipBLASLt -- plain GEMMs through torch, the weight
# gradient split by hand into batched products; 1.4 % faster on the whole step -- left the product in round 6: tools/micro/try_bmm_splitk.py
# keeps the measurement.)


def _t1x1_packs(weight):
    """[9][Cout][Cin] bf16 operands of the tap-masked 3x3 kernel for a ConvTranspose2d(k = 1) weight [Cin, Cout, 1, 1]: forward
    (tap 4 = W^T) and data gradient (tap 4 = W); the masked taps are never fetched. Cached on the parameter: version counter + storage
    + the invalidation epoch that invalidate_all_pack_plans() bumps (writes through `.data` -- dist.broadcast(p.data), EMA, checkpoint
    surgery -- move neither the counter nor the storage)."""
    from . import conv_f32
    key = (conv_f32._EPOCH[0], weight._version, weight.data_ptr())
    c = getattr(weight, "_octa_t1x1", None)
    if c is None or c[0] != key:
        cin, cout = weight.shape[0], weight.shape[1]
        wm = weight.detach().reshape(cin, cout).to(torch.bfloat16)
        fwd = torch.zeros((9, cout, cin), dtype=torch.bfloat16, device=weight.device)
        dg = torch.zeros((9, cin, cout), dtype=torch.bfloat16, device=weight.device)
        fwd[4] = wm.t()
        dg[4] = wm
        c = (key, slice_major(fwd), slice_major(dg))
        weight._octa_t1x1 = c
    return c[1], c[2]


class _ConvT1x1NHWC(torch.autograd.Function):
    """ConvTranspose2d(kernel 1, stride 1, no bias) = a 1x1 convolution with the transposed weight: the MFMA 3x3 kernels with the
    tap mask 0b000010000 (forward, data gradient on the swapped weight, weight gradient), csrc/conv.hip."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        n, h, w, cin = x.shape
        cout = weight.shape[1]
        ctx.w_shape, ctx.w_dtype = weight.shape, weight.dtype
        fwd, dg = _t1x1_packs(weight)
        ctx.save_for_backward(x, dg)
        return conv3x3_nhwc(x, fwd, tap_mask=1 << 4)

    @staticmethod
    def backward(ctx, dy):
        x, wm = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        n, h, w, cin = x.shape
        cout = dy.shape[3]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_nhwc(dy, wm, tap_mask=1 << 4)          # wm: the data-gradient pack [9][Cin][Cout]
        if ctx.needs_input_grad[1]:
            # as a convolution weight [Cout_conv = cout][Cin_conv = cin][3][3] with only tap (1, 1): dW[co][ci] = sum_p dy[p][co] x[p][ci]
            dw = conv3x3_nhwc_wgrad(x, dy, tap_mask=1 << 4)[:, :, 1, 1].t().reshape(ctx.w_shape).to(ctx.w_dtype)
        return dx, dw


def conv_transpose_kxk_nhwc(x, weight, k):
    if k == 2 and x.shape[-1] % 32 == 0 and weight.shape[1] % 32 == 0 and USE_MFMA_CONVT:
        return _ConvT2x2NHWC.apply(x, weight)
    if k == 1 and x.shape[-1] % 32 == 0 and weight.shape[1] % 32 == 0 and USE_MFMA_CONVT:
        return _ConvT1x1NHWC.apply(x, weight)
    from . import networks
    networks._vendor_fallback("DynUNet transposed convolution (NHWC)", f"k = {k}, {x.shape[-1]} -> {weight.shape[1]} channels: GEMM through hipBLASLt "
                              "(the MFMA kernels need k in (1, 2) and channel counts % 32 == 0)")
    return _conv_transpose_kxk_gemm(x, weight, k)


USE_MFMA_CONVT = True


def _conv_transpose_kxk_gemm(x, weight, k):
    """ConvTranspose2d with kernel == stride == k (no padding, no bias) on NHWC bf16: every input pixel owns a
    disjoint k x k output patch, so it is one GEMM [N*H*W, Cin] x [Cin, k*k*Cout] (hipBLASLt through torch) plus a
    pixel shuffle. weight: the torch parameter [Cin, Cout, k, k]."""
    n, h, w, cin = x.shape
    cout = weight.shape[1]
    wm = weight.permute(0, 2, 3, 1).reshape(cin, k * k * cout).to(torch.bfloat16)
    y = torch.matmul(x.reshape(n * h * w, cin), wm)
    if k == 1:
        return y.view(n, h, w, cout)
    return y.view(n, h, w, k, k, cout).permute(0, 1, 3, 2, 4, 5).reshape(n, h * k, w * k, cout)


class _Head1NHWC(torch.autograd.Function):
    """1x1 convolution to ONE channel with bias on the streaming HIP kernels (csrc/conv.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        n, h, w, c = x.shape
        wv = weight.reshape(-1).float().contiguous()
        y = torch.empty((n, h, w, 1), dtype=torch.bfloat16, device=x.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        bv = bias.float().contiguous() if bias is not None else None      # read on the device: no host round trip in the step
        rc = _native.lib().octa_head1_nhwc_fwd_b(_native.ctx(x.device.index), p(x), p(wv), p(bv) if bv is not None else None,
                                                 n * h * w, c, p(y), _native.current_stream_ptr())
        _native.check(rc, "octa_head1_nhwc_fwd_b")
        ctx.save_for_backward(x, wv)
        ctx.w_shape, ctx.w_dtype, ctx.has_bias = weight.shape, weight.dtype, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wv = ctx.saved_tensors
        dy = dy.contiguous().to(torch.bfloat16)
        n, h, w, c = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(1, dtype=torch.float32, device=x.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = _native.lib().octa_head1_nhwc_bwd(_native.ctx(x.device.index), p(x), p(dy), p(wv), n * h * w, c, p(dx), p(dw), p(db),
                                               _native.current_stream_ptr())
        _native.check(rc, "octa_head1_nhwc_bwd")
        return dx, dw.view(ctx.w_shape).to(ctx.w_dtype), (db.to(ctx.w_dtype) if ctx.has_bias else None)


class _InstNormLReLUHead1NHWC(torch.autograd.Function):
    """InstanceNorm(affine) + LeakyReLU + 1x1 convolution to one channel (+ bias) in one pair of passes each way
    (csrc/norm.hip octa_instnorm_lrelu_head1_nhwc_*): the normalised tensor and its gradient never reach HBM."""

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, eps, head_w, head_b, partials=None):
        x = x.contiguous()
        assert x.dtype == torch.bfloat16 and x.dim() == 4
        B, H, W, C = x.shape
        f32 = dict(dtype=torch.float32, device=x.device)
        g = gamma.float().contiguous() if gamma is not None else None
        bt = beta.float().contiguous() if beta is not None else None
        hw_ = head_w.reshape(-1).float().contiguous()
        hb = head_b.float().contiguous() if head_b is not None else None
        mean, rstd = torch.empty(B * C, **f32), torch.empty(B * C, **f32)
        logits = torch.empty((B, H, W, 1), dtype=torch.bfloat16, device=x.device)
        if partials is not None and partials.dtype == torch.float64:       # statistics of x in slot form from the convolution that wrote it
            rc = _native.lib().octa_instnorm_lrelu_head1_nhwc_fwd_s(_native.ctx(x.device.index), _p(x), _p(g), _p(bt), _p(hw_), _p(hb), _p(mean), _p(rstd),
                                                                    _p(logits), B, C, H * W, float(slope), float(eps), _p(partials),
                                                                    int(partials.shape[0]), _native.current_stream_ptr())
        else:
            rc = _native.lib().octa_instnorm_lrelu_head1_nhwc_fwd(_native.ctx(x.device.index), _p(x), _p(g), _p(bt), _p(hw_), _p(hb), _p(mean), _p(rstd),
                                                                  _p(logits), B, C, H * W, float(slope), float(eps), _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_lrelu_head1_nhwc_fwd")
        ctx.save_for_backward(x, g, bt, hw_, mean, rstd)
        ctx.slope = float(slope)
        ctx.meta = (gamma is not None, beta is not None, head_b is not None, head_w.shape, head_w.dtype,
                    gamma.dtype if gamma is not None else None)
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, g, bt, hw_, mean, rstd = ctx.saved_tensors
        has_g, has_b, has_hb, hw_shape, hw_dtype, g_dtype = ctx.meta
        dl = dl.contiguous()
        if dl.dtype != torch.bfloat16:
            dl = dl.to(torch.bfloat16)
        B, H, W, C = x.shape
        f32 = dict(dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dgb = torch.empty(2 * C, **f32)                   # back to back: one fill clears both
        dg, db = dgb[:C], dgb[C:]
        dhw, dhb = torch.empty(C, **f32), torch.empty(1, **f32)
        rc = _native.lib().octa_instnorm_lrelu_head1_nhwc_bwd(_native.ctx(x.device.index), _p(x), _p(dl), _p(g), _p(bt), _p(hw_), _p(mean), _p(rstd),
                                                              _p(dx), _p(dg), _p(db), _p(dhw), _p(dhb), B, C, H * W, ctx.slope,
                                                              _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_lrelu_head1_nhwc_bwd")
        return (dx, dg.to(g_dtype) if has_g else None, db.to(g_dtype) if has_b else None, None, None,
                dhw.view(hw_shape).to(hw_dtype), dhb.to(hw_dtype) if has_hb else None, None)


USE_FUSED_NORM_HEAD = True     # module switch (development aid)


def norm_lrelu_head1_ok(c, head_weight):
    return USE_FUSED_NORM_HEAD and head_weight.shape[0] == 1 and head_weight.shape[1] == c and c in (8, 16, 32, 64, 128, 256)


def instance_norm_leaky_relu_head1_nhwc(x, gamma, beta, negative_slope, eps, head_weight, head_bias, partials=None):
    """x [N,H,W,C] bf16 (raw convolution output) -> logits [N,H,W,1] bf16 = head(lrelu(instance_norm(x))). partials: the slot-form
    statistics of x from the convolution that wrote it (conv3x3(..., want_stats=True))."""
    return _InstNormLReLUHead1NHWC.apply(x, gamma, beta, negative_slope, eps, head_weight, head_bias, partials)


def conv1x1_bias_nhwc(x, weight, bias):
    """1x1 convolution head: weight [Cout, Cin, 1, 1], bias [Cout] -> [N,H,W,Cout] bf16."""
    if weight.shape[0] == 1 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 256 and 256 % (x.shape[-1] // 8) == 0:
        return _Head1NHWC.apply(x, weight, bias)
    from . import networks
    networks._vendor_fallback("DynUNet output block (NHWC)", f"{x.shape[-1]} -> {weight.shape[0]} channels: GEMM through hipBLASLt (the streaming head covers one output channel)")
    n, h, w, cin = x.shape
    y = torch.matmul(x.reshape(n * h * w, cin), weight.reshape(weight.shape[0], cin).t().to(torch.bfloat16))
    if bias is not None:
        y = y + bias.to(torch.bfloat16)
    return y.view(n, h, w, weight.shape[0])


# ---- normalise-on-load path: the normalised activations never go to HBM ---------------------------------------------
# A "lazy" activation is a triple (t, scale, shift): t aliases the RAW output of a convolution, and every consumer reads
# it as lrelu(t * scale + shift). For autograd t IS the normalised activation: the convolutions return dL/d(normalised)
# for it, contributions of several consumers are summed by autograd, and _LazyNorm.backward turns the sum into the
# gradient of the raw tensor with the usual InstanceNorm+LeakyReLU backward kernels.

LRELU_SLOPE = 0.01


class _LazyNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, slope, eps):
        x = x.contiguous()
        B, C = x.shape[0], x.shape[3]
        hw = x.shape[1] * x.shape[2]
        f32 = dict(dtype=torch.float32, device=x.device)
        mean, rstd, scale, shift = (torch.empty(B * C, **f32) for _ in range(4))
        w = weight.float().contiguous() if weight is not None else None
        b = bias.float().contiguous() if bias is not None else None
        rc = _native.lib().octa_instnorm_nhwc_stats(_native.ctx(x.device.index), _p(x), _p(w), _p(b), _p(mean), _p(rstd), _p(scale), _p(shift),
                                                    B, C, hw, float(eps), _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_nhwc_stats")
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.slope, ctx.has_w, ctx.has_b = float(slope), weight is not None, bias is not None
        ctx.w_dtype = weight.dtype if weight is not None else None
        y = x.view(x.shape)          # same storage: consumers apply scale / shift / LeakyReLU while loading
        ctx.mark_non_differentiable(scale, shift)
        return y, scale, shift

    @staticmethod
    def backward(ctx, dy, _ds, _dh):
        return _InstNormLReLUNHWC.backward(ctx, dy)[:5]


def lazy_norm(x, weight, bias, negative_slope=LRELU_SLOPE, eps=1e-5):
    """-> (t, scale, shift): see the section comment."""
    return _LazyNorm.apply(x, weight, bias, negative_slope, eps)


class _Materialise(torch.autograd.Function):
    """y = lrelu(t * scale + shift) as a real tensor, for the consumers that are not normalise-on-load kernels."""

    @staticmethod
    def forward(ctx, t, scale, shift, slope):
        t = t.contiguous()
        y = torch.empty_like(t)
        B, C = t.shape[0], t.shape[3]
        rc = _native.lib().octa_scale_shift_lrelu_nhwc(_native.ctx(t.device.index), _p(t), _p(y), _p(scale), _p(shift), B, C,
                                                       t.shape[1] * t.shape[2], float(slope), _native.current_stream_ptr())
        _native.check(rc, "octa_scale_shift_lrelu_nhwc")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None      # t stands for the normalised activation in the graph


def materialise(lazy, slope=LRELU_SLOPE):
    t, sc, sh = lazy
    return t if sc is None else _Materialise.apply(t, sc, sh, slope)


class _ConvLazy(torch.autograd.Function):
    """conv3x3 over one or two (virtually concatenated) lazy inputs, stride 1 or 2 (stride 2: single input)."""

    @staticmethod
    def forward(ctx, x1, sc1, sh1, x2, sc2, sh2, weight, stride, slope):
        x1 = x1.contiguous()
        n, h, w, c1 = x1.shape
        c2 = 0
        if x2 is not None:
            x2 = x2.contiguous()
            c2 = x2.shape[3]
        cin, cout = c1 + c2, weight.shape[0]
        assert weight.shape[1] == cin and c1 % 32 == 0 and c2 % 32 == 0 and cout % 32 == 0
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=x1.device)
        wt = pack_weight(weight)
        rc = _native.lib().octa_conv3x3_nhwc_fwd4(_native.ctx(x1.device.index), _p(x1), _p(x2), c1, _p(wt), _p(y), None, cout, n, h, w, cin, cout,
                                                  int(stride), 1, 0x1ff, 1, 0, 0, _p(sc1), _p(sh1), _p(sc2), _p(sh2), float(slope),
                                                  _native.current_stream_ptr())
        _native.check(rc, "octa_conv3x3_nhwc_fwd4")
        ctx.save_for_backward(x1, sc1, sh1, x2, sc2, sh2, weight)
        ctx.stride, ctx.slope = int(stride), float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, sc1, sh1, x2, sc2, sh2, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        n, h, w, c1 = x1.shape
        c2 = x2.shape[3] if x2 is not None else 0
        cin, cout, st = c1 + c2, weight.shape[0], ctx.stride
        lib, hctx, stream = _native.lib(), _native.ctx(x1.device.index), _native.current_stream_ptr()
        dx1 = dx2 = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
            dx1 = torch.empty_like(x1)
            dx2 = torch.empty_like(x2) if x2 is not None else None
            wd = pack_weight_dgrad(weight)
            rc = lib.octa_conv3x3_nhwc_fwd4(hctx, _p(dy), None, cout, _p(wd), _p(dx1), _p(dx2), c1, n, dy.shape[1], dy.shape[2], cout, cin, 1, st,
                                            0x1ff, 1, 0, 0, None, None, None, None, 0.0, stream)
            _native.check(rc, "octa_conv3x3_nhwc_fwd4 (data gradient)")
        if ctx.needs_input_grad[6]:
            if st == 1:
                dwf = torch.empty((9, cout, cin), dtype=torch.float32, device=x1.device)
                rc = lib.octa_conv3x3_nhwc_wgrad3(hctx, _p(x1), _p(x2), c1, _p(dy), _p(dwf), n, h, w, cin, cout, 0x1ff, _p(sc1), _p(sh1), _p(sc2),
                                                  _p(sh2), ctx.slope, stream)
                _native.check(rc, "octa_conv3x3_nhwc_wgrad3")
                dw = dwf.view(3, 3, cout, cin).permute(2, 3, 0, 1).to(weight.dtype)
            else:
                xm = x1 if sc1 is None else _Materialise.apply(x1, sc1, sh1, ctx.slope)   # parity planes need the tensor itself
                dw = _s2_wgrad(xm, dy).to(weight.dtype)
        return dx1, None, None, dx2, None, None, dw, None, None


def conv3x3_lazy(a, weight, stride=1, b=None, slope=LRELU_SLOPE):
    """a, b: lazy activations (t, scale, shift) (scale None = plain tensor); b is virtually concatenated after a."""
    t2, s2, h2 = b if b is not None else (None, None, None)
    return _ConvLazy.apply(a[0], a[1], a[2], t2, s2, h2, weight, stride, slope)


# ---- 4x4 stride-1 convolution (PatchGAN inner layers) -----------------------------------------------------------------

def conv4x4_nhwc(x, wt, pad):
    """x [N,H,W,Cin] bf16, wt [16,Cout,Cin] bf16 (tap 4r+s) -> [N,H+2pad-3,W+2pad-3,Cout] bf16 (csrc/conv.hip, KS = 4)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and wt.dtype == torch.bfloat16 and wt.is_contiguous()
    n, h, w, cin = x.shape
    cout = wt.shape[1]
    y = torch.empty((n, h + 2 * pad - 3, w + 2 * pad - 3, cout), dtype=torch.bfloat16, device=x.device)
    rc = _native.lib().octa_conv4x4_nhwc_fwd(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wt.data_ptr()),
                                             ctypes.c_void_p(y.data_ptr()), n, h, w, cin, cout, int(pad), _native.current_stream_ptr())
    _native.check(rc, "octa_conv4x4_nhwc_fwd")
    return y


class _Conv4x4NHWC(torch.autograd.Function):
    """Conv2d(Cin, Cout, 4, stride 1, padding 1) without bias on NHWC bf16: forward, data gradient and weight gradient on
    the K = 4 instantiations of the MFMA kernels (the weight gradient holds 16 accumulator tiles = 256 AGPRs per wave)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        return conv4x4_nhwc(x, pack_weight(weight), 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # dx = full correlation of dy with the flipped kernel, channels transposed: padding 3 - 1 = 2
            dx = conv4x4_nhwc(dy, pack_weight_dgrad(weight), 2)
        if ctx.needs_input_grad[1]:
            n, h, w, cin = x.shape
            cout = dy.shape[3]

            def wg():
                d = torch.empty((16, cout, cin), dtype=torch.float32, device=x.device)
                rc = _native.lib().octa_conv4x4_nhwc_wgrad(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                                                           ctypes.c_void_p(d.data_ptr()), n, h, w, cin, cout, _native.current_stream_ptr())
                _native.check(rc, "octa_conv4x4_nhwc_wgrad")
                return d.view(4, 4, cout, cin).permute(2, 3, 0, 1).to(weight.dtype)
            dw = _wgrad_to(weight, wg)
        return dx, dw


def conv4x4(x, weight):
    """weight: the torch parameter [Cout, Cin, 4, 4]; stride 1, padding 1, no bias."""
    return _Conv4x4NHWC.apply(x, weight)
