"""Networks on the hot path (reference: models/networks.py; DynUNet is imported from MONAI there,
networks.py:6,1010 -- MONAI is not installed here, so the architecture is restated in plain torch).

DynUNet as configured by configs/config_ves_seg-S.yml:6-13 (spatial_dims 2, kernel 3, strides
[1,2,2,2,1], upsample kernels [1,2,2,2], filters [32,64,128,256,512], instance norm (affine) +
LeakyReLU(0.01), bias-free convs, 1x1 head with bias, no deep supervision, no residual blocks).
Module and parameter names follow MONAI's DynUNet (input_block / downsamples / bottleneck / upsamples /
output_block with conv1.conv, norm1, transp_conv.conv, conv_block...) so that reference checkpoints
(`{'epoch','model','optimizer','config'}`, utils/visualizer.py:225-238) load; MONAI's state_dict also
carries the same tensors a second time under `skip_layers.*` -- those aliases are emitted on save and
ignored on load. Parity with MONAI itself is UNPINNED (no MONAI, no checkpoint in the container);
numerics are pinned against this module run in fp32 on the CPU.
"""
from typing import Sequence

import torch
import torch.nn as nn

from . import resample


# CUDA tensors go through the fused HIP InstanceNorm+LeakyReLU kernels (csrc/norm.hip); set to False to run the
# plain torch modules (used by the parity tests as the reference).
USE_FUSED_NORM = True
# CUDA inputs run channels-last in bf16 through the hand-written MFMA convolution (csrc/conv.hip) and the NHWC
# norm kernels; False = torch/MIOpen modules (the fp32 reference path of the parity tests)
USE_MFMA_CONV = True
USE_F32_MFMA = True      # fp32 forward passes that record no gradient run csrc/conv_f32.hip (exact-fp32 MFMA) instead of the vendor library
# normalise-on-load (the normalised activations never go to HBM: models/mfma_conv.py) is implemented and tested but
# measured SLOWER on MI355X (37.5 vs 31.9 ms per step): every output-channel block of a layer re-stages and
# re-normalises the same input tile, which costs more VALU work than the two HBM passes it saves
USE_LAZY_NORM = False      # module switch (tests set it): normalise-on-load, re-measured in rounds 4 and 5 on the DMA-staged kernels and slower
# InstanceNorm statistics accumulated in the convolution's epilogue instead of a statistics pass over the stored tensor. Rounds 1-4: per-tile
# partials + a fold launch (conv.hip _fwd5), measured slower every time (33.6 vs 32.1, 20.0 vs 18.4 ms). Round 5: slot form (conv.hip
# octa_conv3x3_nhwc_fwd7 -- the sums ride in the bf16 conversion loop as v_dot2c_f32_bf16, the waves meet behind the output tile in LDS under
# the barrier the tile needs anyway, one pair of double atomics per channel and tile into 16 slots; the apply pass adds the slots): same box,
# B = 4 at 1216^2: 18.7 -> 18.5 ms per step (profiles/r05_unet_epilogue_stats_ab.log). Default ON; networks.USE_EPILOGUE_STATS = False runs the statistics pass.
import os as _os
USE_EPILOGUE_STATS = True      # module switch; the slots live in the DMA-staged kernel's epilogue

# ---- which kernels ran ------------------------------------------------------------------------------------------------
# The product path of a CUDA forward is the hand-written HIP kernels (csrc/conv.hip, conv_f32.hip, norm.hip, thin_conv.hip,
# blur.hip). Whenever a CUDA forward is about to leave them for the torch modules (MIOpen / hipBLASLt) WITHOUT having been told to
# -- shapes the kernels do not cover, fp32 passes that record gradients (`General.amp: false` training), fp32 passes of the GAN
# networks -- `_vendor_fallback` counts it, warns ONCE per (place, reason), and raises under OCTA_STRICT=1 (tests/conftest.py and
# bench.py set it: the measured and the tested path is the HIP path or nothing). Deliberate use of the torch modules as the
# REFERENCE of a parity test is not a fallback: `with vendor_reference():` (or USE_MFMA_CONV = False) says so.
import collections as _collections
import contextlib as _contextlib
import warnings as _warnings

PATH_COUNTS = _collections.Counter()     # 'mfma' (bf16 NHWC passes), 'f32_mfma' (exact-fp32 convolutions), 'vendor' (fallbacks)
_WARNED = set()
_REFERENCE_DEPTH = [0]


class VendorFallbackError(RuntimeError):
    pass


@_contextlib.contextmanager
def vendor_reference():
    """The torch modules (vendor libraries on the GPU) are wanted here: the reference side of a parity test, a timing comparison."""
    _REFERENCE_DEPTH[0] += 1
    try:
        yield
    finally:
        _REFERENCE_DEPTH[0] -= 1


def _vendor_fallback(where, why):
    if _REFERENCE_DEPTH[0] > 0 or not USE_MFMA_CONV:
        return
    PATH_COUNTS['vendor'] += 1
    msg = (f"{where}: this CUDA forward leaves the hand-written HIP kernels for the torch modules (MIOpen / hipBLASLt): {why}. "
           "Results stay correct, speed and the no-vendor-kernel claim do not; OCTA_STRICT=1 turns this into an error.")
    if _os.environ.get('OCTA_STRICT', '0') == '1':
        raise VendorFallbackError(msg)
    if (where, why) not in _WARNED:
        _WARNED.add((where, why))
        _warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _why_not_own_kernels(c, x):
    if x.dtype == torch.float32 and torch.is_grad_enabled() and (x.requires_grad or c.weight.requires_grad):
        return "fp32 pass that records gradients (the exact-fp32 MFMA kernels are forward-only; train with General.amp: true)"
    if x.dtype == torch.float32:
        return "fp32 layer shape outside csrc/conv_f32.hip (kernel/stride 1/1, 3/1, 3/2, 4/1, 7/1, transposed 1/1, 2/2)"
    return f"{x.dtype} modules path: the bf16 NHWC path did not apply ({_LAST_MFMA_REFUSAL[0] or 'not a bf16 / bf16-autocast pass'})"


_LAST_MFMA_REFUSAL = [None]


class _Conv(nn.Module):
    """MONAI `Convolution(conv_only=True)`: a container whose only child is `conv`."""

    def __init__(self, cin, cout, kernel, stride, bias=False, transposed=False):
        super().__init__()
        if transposed:
            pad = int((kernel - stride + 1) / 2)
            self.conv = nn.ConvTranspose2d(cin, cout, kernel, stride, padding=pad,
                                           output_padding=2 * pad + stride - kernel, bias=bias)
        else:
            self.conv = nn.Conv2d(cin, cout, kernel, stride, padding=int((kernel - stride + 1) / 2), bias=bias)

    def forward(self, x):
        c = self.conv
        if USE_F32_MFMA and x.is_cuda and x.dtype == torch.float32:
            # fp32 without gradients (test.py / validate.py, as the reference runs them: no autocast): exact-fp32 MFMA convolution
            from . import conv_f32
            if conv_f32.applies(c, x):
                PATH_COUNTS['f32_mfma'] += 1
                return conv_f32.forward(c, x)
        if x.is_cuda:
            _vendor_fallback(f"DynUNet {type(c).__name__}({c.in_channels}->{c.out_channels}, k{c.kernel_size[0]}, s{c.stride[0]})",
                             _why_not_own_kernels(c, x))
        return c(x)


class UnetBasicBlock(nn.Module):
    def __init__(self, cin, cout, kernel, stride):
        super().__init__()
        self.conv1 = _Conv(cin, cout, kernel, stride)
        self.conv2 = _Conv(cout, cout, kernel, 1)
        self.lrelu = nn.LeakyReLU(0.01, inplace=True)
        self.norm1 = nn.InstanceNorm2d(cout, affine=True)
        self.norm2 = nn.InstanceNorm2d(cout, affine=True)

    def _norm_act(self, norm, x):
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and USE_FUSED_NORM:
            from .fused_ops import instance_norm_leaky_relu
            return instance_norm_leaky_relu(x, norm.weight, norm.bias, self.lrelu.negative_slope, norm.eps)
        if x.is_cuda and USE_FUSED_NORM:
            _vendor_fallback("DynUNet InstanceNorm + LeakyReLU", f"dtype {x.dtype} (csrc/norm.hip covers float32 and bfloat16)")
        return self.lrelu(norm(x))

    def forward(self, x):
        x = self._norm_act(self.norm1, self.conv1(x))
        return self._norm_act(self.norm2, self.conv2(x))


class UnetUpBlock(nn.Module):
    def __init__(self, cin, cout, kernel, up_kernel):
        super().__init__()
        self.transp_conv = _Conv(cin, cout, up_kernel, up_kernel, transposed=True)
        self.conv_block = UnetBasicBlock(cout + cout, cout, kernel, 1)

    def forward(self, x, skip):
        return self.conv_block(torch.cat((self.transp_conv(x), skip), dim=1))


class UnetOutBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv(cin, cout, 1, 1, bias=True)

    def forward(self, x):
        return self.conv(x)


class DynUNet(nn.Module):
    def __init__(self, spatial_dims: int = 2, in_channels: int = 1, out_channels: int = 1,
                 kernel_size: Sequence[int] = (3, 3, 3, 3, 3), strides: Sequence[int] = (1, 2, 2, 2, 1),
                 upsample_kernel_size: Sequence[int] = (1, 2, 2, 2, 1), filters: Sequence[int] = None, **unused):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("only the 2-D DynUNet of the OCTA configs is implemented")
        ks = [k if isinstance(k, int) else k[0] for k in kernel_size]
        st = [s if isinstance(s, int) else s[0] for s in strides]
        up = [u if isinstance(u, int) else u[0] for u in upsample_kernel_size]
        if filters is None:
            filters = [min(2 ** (5 + i), 512) for i in range(len(st))]
        self.filters = list(filters)
        self.input_block = UnetBasicBlock(in_channels, filters[0], ks[0], st[0])
        self.downsamples = nn.ModuleList(
            [UnetBasicBlock(filters[i - 1], filters[i], ks[i], st[i]) for i in range(1, len(st) - 1)])
        self.bottleneck = UnetBasicBlock(filters[-2], filters[-1], ks[-1], st[-1])
        inp, out = filters[1:][::-1], filters[:-1][::-1]
        kern, ups = ks[1:][::-1], up[::-1]
        self.upsamples = nn.ModuleList([UnetUpBlock(i, o, k, u) for i, o, k, u in zip(inp, out, kern, ups)])
        self.output_block = UnetOutBlock(filters[0], out_channels)
        self._register_state_dict_hook(self._add_skip_layer_aliases)
        self._register_load_state_dict_pre_hook(self._drop_skip_layer_aliases)

    # MONAI keeps the same modules a second time inside the recursive `skip_layers` wrapper
    def _alias_prefixes(self):
        depth = len(self.downsamples) + 1
        m = {"input_block.": "skip_layers.downsample.", f"upsamples.{depth - 1}.": "skip_layers.upsample."}
        nl = "skip_layers."
        for i in range(len(self.downsamples)):
            nl += "next_layer."
            m[f"downsamples.{i}."] = nl + "downsample."
            m[f"upsamples.{depth - 2 - i}."] = nl + "upsample."
        m["bottleneck."] = nl + "next_layer."
        return m

    @staticmethod
    def _add_skip_layer_aliases(module, state_dict, prefix, local_metadata):
        for src, dst in module._alias_prefixes().items():
            for k in [k for k in state_dict if k.startswith(prefix + src)]:
                state_dict[prefix + dst + k[len(prefix + src):]] = state_dict[k]
        return state_dict

    def _drop_skip_layer_aliases(self, state_dict, prefix, *args):
        for k in [k for k in state_dict if k.startswith(prefix + "skip_layers.")]:
            del state_dict[k]

    # ---- channels-last bf16 path on the hand-written kernels (same parameters, same state dict) ----
    @staticmethod
    def _basic_block_nhwc(blk, x, skip=None, mb_recv=None, mb_send=None, head=None):
        """x, skip and the result are lazy activations (tensor, scale, shift) -- the normalised tensors are applied by the
        consuming kernels while loading (mfma_conv.py, normalise-on-load), or plain tensors with scale None."""
        from . import mfma_conv as mc
        c1, c2 = blk.conv1.conv, blk.conv2.conv
        st = c1.stride[0]
        if not USE_LAZY_NORM:
            xt, sk = x[0], (skip[0] if skip is not None else None)
            if sk is not None and xt.shape[-1] % 32 == 0 and sk.shape[-1] % 32 == 0 and st == 1:
                # conv over the virtual concatenation (x, skip); mb_send: the skip's gradient rides in the encoder's data-gradient launch
                y = mc.conv3x3_cat(xt, sk, c1.weight, USE_EPILOGUE_STATS, mb_send)
            else:
                if sk is not None:
                    xt = torch.cat((xt, sk), dim=-1)
                y = mc.conv3x3(xt, c1.weight, st, USE_EPILOGUE_STATS, mb_recv if sk is None else None)
            y, part = y if USE_EPILOGUE_STATS else (y, None)
            y = mc.instance_norm_leaky_relu_nhwc(y, blk.norm1.weight, blk.norm1.bias, blk.lrelu.negative_slope, blk.norm1.eps, part)
            fused_head = head is not None and mc.norm_lrelu_head1_ok(c2.weight.shape[0], head.weight)
            want2 = USE_EPILOGUE_STATS
            y = mc.conv3x3(y, c2.weight, 1, want2)
            y, part = y if want2 else (y, None)
            if fused_head:
                # last block: norm + activation + the 1x1 output convolution in one pair of passes, nothing normalised goes to HBM (the
                # norm's statistics from conv2's epilogue: no pass over the raw tensor in front of it either)
                return mc.instance_norm_leaky_relu_head1_nhwc(y, blk.norm2.weight, blk.norm2.bias, blk.lrelu.negative_slope, blk.norm2.eps,
                                                              head.weight, head.bias, part)
            y = mc.instance_norm_leaky_relu_nhwc(y, blk.norm2.weight, blk.norm2.bias, blk.lrelu.negative_slope, blk.norm2.eps, part)
            if head is not None:
                return mc.conv1x1_bias_nhwc(y, head.weight, head.bias)
            return (y, None, None)
        if x[0].shape[-1] % 32 != 0 or (skip is not None and (skip[0].shape[-1] % 32 != 0 or st != 1)):
            # channel-padded first layer (or an odd split): through the single-input binding on a real tensor
            xt = mc.materialise(x)
            if skip is not None:
                xt = torch.cat((xt, mc.materialise(skip)), dim=-1)
            y = mc.conv3x3(xt, c1.weight, st)
        else:
            y = mc.conv3x3_lazy(x, c1.weight, st, skip, blk.lrelu.negative_slope)
        y = mc.lazy_norm(y, blk.norm1.weight, blk.norm1.bias, blk.lrelu.negative_slope, blk.norm1.eps)
        y = mc.conv3x3_lazy(y, c2.weight, 1, None, blk.lrelu.negative_slope)
        return mc.lazy_norm(y, blk.norm2.weight, blk.norm2.bias, blk.lrelu.negative_slope, blk.norm2.eps)

    def _mfma_path_ok(self, x):
        why = self._mfma_refusal(x)
        _LAST_MFMA_REFUSAL[0] = why
        return why is None

    def _mfma_refusal(self, x):
        """None when the bf16 NHWC path on the hand-written kernels applies, else the reason it does not."""
        if not (USE_MFMA_CONV and x.is_cuda and x.dim() == 4):
            return "USE_MFMA_CONV off, CPU tensor or not [N, C, H, W]"
        # bf16 arithmetic only where the caller asked for it (bf16 input or bf16 autocast, as the trainers do);
        # fp32 inputs keep the fp32 modules (logits within 1e-4 of the CPU reference)
        if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)):
            return "not a bf16 / bf16-autocast pass"
        blocks = [self.input_block, *self.downsamples, self.bottleneck, *[u.conv_block for u in self.upsamples]]
        for b in blocks:
            for c in (b.conv1.conv, b.conv2.conv):
                if c.kernel_size != (3, 3) or c.stride[0] not in (1, 2) or c.out_channels % 32 or c.bias is not None:
                    return f"convolution {c.in_channels}->{c.out_channels} k{c.kernel_size} s{c.stride} bias={c.bias is not None}: the MFMA path needs 3x3, stride 1 / 2, out_channels % 32 == 0, no bias"
        for u in self.upsamples:
            t = u.transp_conv.conv
            if t.kernel_size != t.stride or t.kernel_size[0] not in (1, 2) or t.bias is not None:
                return f"transposed convolution k{t.kernel_size} s{t.stride}: the MFMA path needs kernel == stride in (1, 2), no bias"
        down = 1
        for d in self.downsamples:
            down *= d.conv1.conv.stride[0]
        down *= self.bottleneck.conv1.conv.stride[0]      # a strided bottleneck halves the map once more: odd sizes take the torch path
        if x.shape[2] % down or x.shape[3] % down:
            return f"input {x.shape[2]}x{x.shape[3]} is not divisible by the total stride {down}"
        return None

    def _forward_nhwc(self, x):
        from . import mfma_conv as mc
        mc.plan_for_module(self)          # all layers' weights packed by one launch per optimiser step
        y = (x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous(), None, None)
        skips = [self._basic_block_nhwc(self.input_block, y)]
        boxes = [mc.SkipGradMailbox()]                       # one per skip tensor: decoder gradient -> encoder data-gradient epilogue
        for d in self.downsamples:
            skips.append(self._basic_block_nhwc(d, skips[-1], mb_recv=boxes[-1]))
            boxes.append(mc.SkipGradMailbox())
        y = self._basic_block_nhwc(self.bottleneck, skips[-1], mb_recv=boxes[-1])
        o = self.output_block.conv.conv
        last = len(self.upsamples) - 1
        for i, (u, s, box) in enumerate(zip(self.upsamples, skips[::-1], boxes[::-1])):
            t = u.transp_conv.conv
            up = mc.conv_transpose_kxk_nhwc(mc.materialise(y), t.weight, t.kernel_size[0])
            y = self._basic_block_nhwc(u.conv_block, (up, None, None), s, mb_send=box, head=o if (i == last and not USE_LAZY_NORM) else None)
        if USE_LAZY_NORM:
            y = mc.conv1x1_bias_nhwc(mc.materialise(y), o.weight, o.bias)
        return y.permute(0, 3, 1, 2)

    def forward(self, x):
        if self._mfma_path_ok(x):
            PATH_COUNTS['mfma'] += 1
            return self._forward_nhwc(x)
        skips = [self.input_block(x)]
        for d in self.downsamples:
            skips.append(d(skips[-1]))
        y = self.bottleneck(skips[-1])
        for u, s in zip(self.upsamples, skips[::-1]):
            y = u(y, s)
        return self.output_block(y)


def init_weights(net: nn.Module, init_type='normal', init_gain=0.02, debug=False, nonlinearity='leaky_relu'):
    """Same rule as models/networks.py:152-184: every module whose class name contains Conv or Linear."""
    def init_func(m):
        name = m.__class__.__name__
        if hasattr(m, 'weight') and (name.find('Conv') != -1 or name.find('Linear') != -1):
            if init_type == 'normal':
                nn.init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                nn.init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in', nonlinearity=nonlinearity)
            elif init_type == 'orthogonal':
                nn.init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError(f'initialization method [{init_type}] is not implemented')
        elif name.find('BatchNorm') != -1:
            nn.init.normal_(m.weight.data, 1.0, init_gain)
            nn.init.constant_(m.bias.data, 0.0)
    net.apply(init_func)
    # writes through `.data` do not move the parameters' version counters: packed copies of the weights (mfma_conv.WeightPackPlan)
    # of this network or of any sub-network that already ran must be rebuilt
    for m in net.modules():
        plan = getattr(m, "_octa_pack_plan", None)
        if plan is not None:
            plan.invalidate()


# ---------------------------------------------------------------------------------------------------
# Contrast-adaptation GAN (reference models/networks.py:186-289 blur filters / Down-/Upsample, :291-443
# ResnetBlock / ResnetGenerator, :445-506 NLayerDiscriminator and the resnetGenerator9 / patchGAN70x70
# factories). Restated so that state_dict keys (`model.<i>.weight`, blur buffers `model.<i>.filt`) and
# parameter counts (11,365,633 / 2,762,689) match reference checkpoints.

_BINOMIAL = {1: [1.], 2: [1., 1.], 3: [1., 2., 1.], 4: [1., 3., 3., 1.], 5: [1., 4., 6., 4., 1.],
             6: [1., 5., 10., 10., 5., 1.], 7: [1., 6., 15., 20., 15., 6., 1.]}


def _blur_kernel(size):
    a = torch.tensor(_BINOMIAL[size])
    k = a[:, None] * a[None, :]
    return k / k.sum()


class Downsample(nn.Module):
    """Anti-aliased stride-2 subsampling: reflect pad, depth-wise binomial blur, stride."""

    def __init__(self, channels, filt_size=3, stride=2):
        super().__init__()
        lo, hi = int((filt_size - 1) / 2), int(-(-(filt_size - 1) // 2))
        self.pad = nn.ReflectionPad2d([lo, hi, lo, hi])
        self.stride, self.channels, self.filt_size = stride, channels, filt_size
        self.register_buffer("filt", _blur_kernel(filt_size)[None, None].repeat(channels, 1, 1, 1))

    def forward(self, x, layout="nchw"):
        if x.is_cuda and self.filt_size == 3 and self.stride == 2:
            return resample.blur_down(x, layout)        # one streaming HIP kernel per direction (csrc/blur.hip)
        assert layout == "nchw"
        return nn.functional.conv2d(self.pad(x), self.filt, stride=self.stride, groups=x.shape[1])


class Upsample(nn.Module):
    """Anti-aliased x2 upsampling: replicate pad, depth-wise transposed binomial blur (x stride^2), crop."""

    def __init__(self, channels, filt_size=4, stride=2):
        super().__init__()
        self.odd = filt_size % 2 == 1
        self.pad_size = int((filt_size - 1) / 2)
        self.stride, self.filt_size = stride, filt_size
        self.pad = nn.ReplicationPad2d([1, 1, 1, 1])
        self.register_buffer("filt", (_blur_kernel(filt_size) * stride ** 2)[None, None].repeat(channels, 1, 1, 1))

    def forward(self, x, layout="nchw"):
        if x.is_cuda and self.filt_size == 4 and self.stride == 2:
            return resample.blur_up(x, layout)
        assert layout == "nchw"
        y = nn.functional.conv_transpose2d(self.pad(x), self.filt, stride=self.stride, padding=1 + self.pad_size,
                                           groups=x.shape[1])[:, :, 1:, 1:]
        return y if self.odd else y[:, :, :-1, :-1]


def _f32_inference_applies(x):
    """A plain fp32 CUDA forward that records no gradient and runs outside autocast: what test.py / validate.py and the frozen generator
    of the data pipeline do in the reference (test.py:75-82, docker/dockershell.sh:14-16)."""
    return (USE_MFMA_CONV and USE_F32_MFMA and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled()
            and not torch.is_grad_enabled())


def _run_f32(mods, x, who):
    """The GAN networks' module lists in fp32 on own kernels (round 6): every convolution on the exact-fp32 MFMA kernel (csrc/conv_f32.hip:
    1x1, 3x3, 4x4, 7x7, bias included), InstanceNorm2d(affine=False) fused with the ReLU / LeakyReLU behind it (csrc/norm.hip, fp32 NCHW),
    reflection pads and anti-aliased resampling on csrc/blur.hip (their modules' own GPU path); residual sums, the first layer's LeakyReLU
    and the Sigmoid are element-wise torch ops. No vendor-library kernel (MIOpen) runs."""
    from . import conv_f32
    from .fused_ops import instance_norm_leaky_relu
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, ResnetBlock):
            x = x + _run_f32(list(m.conv_block), x, who)
            i += 1
        elif isinstance(m, nn.Conv2d):
            if conv_f32.applies(m, x):
                x = conv_f32.forward(m, x)
            else:
                _vendor_fallback(who, f"fp32 layer {i} ({m.kernel_size}, stride {m.stride}, padding {m.padding}) is outside csrc/conv_f32.hip")
                x = m(x)
            i += 1
        elif isinstance(m, nn.InstanceNorm2d) and not m.affine and not m.track_running_stats:
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            slope = 0.0 if isinstance(nxt, nn.ReLU) else (float(nxt.negative_slope) if isinstance(nxt, nn.LeakyReLU) else None)
            x = instance_norm_leaky_relu(x.contiguous(), None, None, 1.0 if slope is None else slope, m.eps)
            i += 1 if slope is None else 2
        else:
            x = m(x)
            i += 1
    return x


class ReflectionPad2d(nn.ReflectionPad2d):
    """nn.ReflectionPad2d whose GPU path is the gather kernels of csrc/blur.hip (torch's backward scatters with atomics)."""

    def forward(self, x):
        p = self.padding
        if x.is_cuda and x.dim() == 4 and p[0] == p[1] == p[2] == p[3]:
            return resample.reflect_pad(x, p[0])
        return super().forward(x)


class ResnetBlock(nn.Module):
    def __init__(self, dim, use_bias=True):
        super().__init__()
        inorm = lambda: nn.InstanceNorm2d(dim, affine=False, track_running_stats=False)
        self.conv_block = nn.Sequential(ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, bias=use_bias), inorm(), nn.ReLU(True),
                                        ReflectionPad2d(1), nn.Conv2d(dim, dim, 3, bias=use_bias), inorm())

    def forward(self, x):
        return x + self.conv_block(x)


class ResnetGenerator(nn.Module):
    """7x7 stem, two blur-downsampling stages, n_blocks residual blocks at 4*ngf, two blur-upsampling stages,
    7x7 head + sigmoid; instance norm without affine, hence biased convolutions."""

    def __init__(self, input_nc=1, output_nc=1, ngf=64, n_blocks=9):
        super().__init__()
        inorm = lambda c: nn.InstanceNorm2d(c, affine=False, track_running_stats=False)
        m = [ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7, bias=True), inorm(ngf), nn.ReLU(True)]
        for i in range(2):
            c = ngf * 2 ** i
            m += [nn.Conv2d(c, 2 * c, 3, padding=1, bias=True), inorm(2 * c), nn.ReLU(True), Downsample(2 * c)]
        m += [ResnetBlock(4 * ngf) for _ in range(n_blocks)]
        for i in range(2):
            c = ngf * 2 ** (2 - i)
            m += [Upsample(c), nn.Conv2d(c, c // 2, 3, padding=1, bias=True), inorm(c // 2), nn.ReLU(True)]
        m += [ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Sigmoid()]
        self.model = nn.Sequential(*m)

    # ---- all 3x3 layers channels-last in bf16 on the MFMA convolution (csrc/conv.hip) ----------------------------------
    # ReflectionPad2d(1) + Conv2d(3, padding 0): mfma_conv.conv3x3_reflect (round 3; rounds 1-2 ran the zero-padded kernel on a
    # reflect-padded copy and cropped the result by one pixel).
    # Every 3x3 convolution is followed by InstanceNorm WITHOUT affine, which subtracts the convolution bias again: it is
    # not added here (its gradient is identically zero in the reference too).
    @staticmethod
    def _resblock_nhwc(blk, x):
        from . import mfma_conv as mc
        c1, n1, c2, n2 = blk.conv_block[1], blk.conv_block[2], blk.conv_block[5], blk.conv_block[6]
        # reflection fused into the convolution's halo fetch; the norms' statistics come from the convolutions' epilogues (round 5: 18 statistics
        # launches per generator pass gone)
        h, part = mc.conv3x3_reflect(x, c1.weight, True) if USE_EPILOGUE_STATS else (mc.conv3x3_reflect(x, c1.weight), None)
        h = mc.instance_norm_leaky_relu_nhwc(h, None, None, 0.0, n1.eps, part)    # InstanceNorm + ReLU
        h, part = mc.conv3x3_reflect(h, c2.weight, True) if USE_EPILOGUE_STATS else (mc.conv3x3_reflect(h, c2.weight), None)
        h = mc.instance_norm_leaky_relu_nhwc(h, None, None, 1.0, n2.eps, part)    # InstanceNorm, no activation
        return x + h

    @staticmethod
    def _is_conv_norm_relu(mods, i):
        """Conv2d(3x3, padding 1, stride 1) -> InstanceNorm2d(affine=False) -> ReLU with MFMA-sized channel counts."""
        if i + 2 >= len(mods):
            return False
        c, n, r = mods[i], mods[i + 1], mods[i + 2]
        return (isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride == (1, 1) and c.groups == 1
                and c.in_channels % 32 == 0 and c.out_channels % 32 == 0 and isinstance(n, nn.InstanceNorm2d) and not n.affine
                and isinstance(r, nn.ReLU))

    def forward(self, x):
        use_mfma = (USE_MFMA_CONV and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)
        if not use_mfma:
            if _f32_inference_applies(x):
                PATH_COUNTS['f32'] = PATH_COUNTS.get('f32', 0) + 1
                return _run_f32(list(self.model), x, "ResnetGenerator")
            if x.is_cuda:
                _vendor_fallback("ResnetGenerator", f"{x.dtype} pass that is neither bf16 autocast nor gradient-free fp32 (the exact-fp32 MFMA kernels are forward-only)")
            return self.model(x)
        PATH_COUNTS['mfma'] += 1
        from . import mfma_conv as mc
        mc.plan_for_module(self)
        mods = list(self.model)
        nhwc = False                                     # layout of x between layers
        i = 0
        from . import thin_conv as tc
        while i < len(mods):
            m = mods[i]
            # 7x7 stem (1 -> ngf, behind ReflectionPad2d, in front of InstanceNorm + ReLU) and head (ngf -> 1, + Sigmoid): streaming
            # kernels (csrc/thin_conv.hip); the stem's bias is subtracted again by the norm, like every 3x3 layer's
            if (not nhwc and i + 3 < len(mods) and isinstance(m, ReflectionPad2d) and isinstance(mods[i + 1], nn.Conv2d) and mods[i + 1].in_channels == 1
                    and mods[i + 1].padding == (0, 0) and tc.supported(mods[i + 1]) and isinstance(mods[i + 2], nn.InstanceNorm2d)
                    and not mods[i + 2].affine and isinstance(mods[i + 3], nn.ReLU) and x.shape[1] == 1):
                xp = resample.reflect_pad(x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous(), m.padding[0], "nhwc")[..., 0]
                x = mc.instance_norm_leaky_relu_nhwc(tc._ConvFrom1.apply(xp, mods[i + 1].weight, None, 0, 1.0), None, None, 0.0, mods[i + 2].eps)
                nhwc = True
                i += 4
                continue
            if (nhwc and i + 1 < len(mods) and isinstance(m, ReflectionPad2d) and isinstance(mods[i + 1], nn.Conv2d) and mods[i + 1].out_channels == 1
                    and mods[i + 1].padding == (0, 0) and tc.supported(mods[i + 1])):
                x = tc.conv_to_1(resample.reflect_pad(x, m.padding[0], "nhwc"), mods[i + 1])[:, None]      # [N, 1, H, W]
                nhwc = False
                i += 2
                continue
            on_path = (self._is_conv_norm_relu(mods, i) or (isinstance(m, ResnetBlock) and m.conv_block[1].in_channels % 32 == 0)
                       or (nhwc and isinstance(m, (Downsample, Upsample))))
            if on_path and not nhwc:
                x, nhwc = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous(), True
            elif not on_path and nhwc:
                x, nhwc = x.permute(0, 3, 1, 2).contiguous(), False
            if not on_path:
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, ResnetBlock, nn.InstanceNorm2d)):
                    _vendor_fallback("ResnetGenerator", f"layer {i} ({type(m).__name__}) has no HIP kernel for its shape")
                x = m(x)
                i += 1
            elif isinstance(m, ResnetBlock):
                x = self._resblock_nhwc(m, x)
                i += 1
            elif isinstance(m, (Downsample, Upsample)):
                x = m(x, "nhwc")
                i += 1
            else:
                x = mc.instance_norm_leaky_relu_nhwc(mc.conv3x3(x, m.weight, 1), None, None, 0.0, mods[i + 1].eps)
                i += 3
        return x.permute(0, 3, 1, 2).contiguous() if nhwc else x


class NLayerDiscriminator(nn.Module):
    """PatchGAN with 4x4 stride-1 convolutions followed by blur-downsampling (anti-aliased variant)."""

    def __init__(self, input_nc=1, ndf=64, n_layers=3):
        super().__init__()
        inorm = lambda c: nn.InstanceNorm2d(c, affine=False, track_running_stats=False)
        seq = [nn.Conv2d(input_nc, ndf, 4, 1, 1), nn.LeakyReLU(0.2, True), Downsample(ndf)]
        prev = 1
        for n in range(1, n_layers):
            mult = min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * prev, ndf * mult, 4, 1, 1, bias=True), inorm(ndf * mult), nn.LeakyReLU(0.2, True),
                    Downsample(ndf * mult)]
            prev = mult
        mult = min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * prev, ndf * mult, 4, 1, 1, bias=True), inorm(ndf * mult), nn.LeakyReLU(0.2, True),
                nn.Conv2d(ndf * mult, 1, 4, 1, 1)]
        self.model = nn.Sequential(*seq)

    # ---- inner layers channels-last in bf16: 4x4 convolutions on the MFMA kernel (csrc/conv.hip, KS = 4), InstanceNorm +
    # LeakyReLU(0.2) and blur-downsampling on the NHWC streaming kernels. A convolution bias in front of an InstanceNorm
    # without affine is subtracted again by the norm and is not added here (its gradient is zero in the reference too).
    # The 1 -> ndf stem (+ bias, LeakyReLU) and the ndf*8 -> 1 head are not matrix-core shaped: streaming kernels (csrc/thin_conv.hip).
    def forward(self, x):
        use_mfma = (USE_MFMA_CONV and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)
        if not use_mfma:
            if _f32_inference_applies(x):
                PATH_COUNTS['f32'] = PATH_COUNTS.get('f32', 0) + 1
                return _run_f32(list(self.model), x, "NLayerDiscriminator")
            if x.is_cuda:
                _vendor_fallback("NLayerDiscriminator", f"{x.dtype} pass that is neither bf16 autocast nor gradient-free fp32 (the exact-fp32 MFMA kernels are forward-only)")
            return self.model(x)
        PATH_COUNTS['mfma'] += 1
        from . import mfma_conv as mc
        from . import thin_conv as tc
        mc.plan_for_module(self)
        mods = list(self.model)
        nhwc = False
        i = 0
        while i < len(mods):
            m = mods[i]
            fused = (isinstance(m, nn.Conv2d) and m.kernel_size == (4, 4) and m.stride == (1, 1) and m.padding == (1, 1)
                     and m.in_channels % 32 == 0 and m.out_channels % 32 == 0 and i + 2 < len(mods)
                     and isinstance(mods[i + 1], nn.InstanceNorm2d) and not mods[i + 1].affine and isinstance(mods[i + 2], nn.LeakyReLU))
            if (not nhwc and isinstance(m, nn.Conv2d) and m.in_channels == 1 and tc.supported(m) and i + 1 < len(mods)
                    and isinstance(mods[i + 1], nn.LeakyReLU) and x.shape[1] == 1):
                x = tc.conv_from_1(x[:, 0], m, mods[i + 1].negative_slope)          # stem: bias + LeakyReLU fused (csrc/thin_conv.hip)
                nhwc = True
                i += 2
                continue
            if nhwc and isinstance(m, nn.Conv2d) and m.out_channels == 1 and tc.supported(m):
                x = tc.conv_to_1(x, m)[:, None]                                     # head: [N, 1, Ho, Wo]
                nhwc = False
                i += 1
                continue
            on_path = fused or (nhwc and isinstance(m, Downsample))
            if on_path and not nhwc:
                x, nhwc = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous(), True
            elif not on_path and nhwc:
                # NCHW copy (37x37x512: nothing): a channels-last VIEW sends torch to MIOpen's NHWC asm solvers, whose data-gradient
                # kernel runs without its workspace in immediate mode and faults (gan_seg_trainer.py)
                x, nhwc = x.permute(0, 3, 1, 2).contiguous(), False
            if fused:
                x = mc.instance_norm_leaky_relu_nhwc(mc.conv4x4(x, m.weight), None, None, mods[i + 2].negative_slope, mods[i + 1].eps)
                i += 3
            elif on_path:
                x = m(x, "nhwc")
                i += 1
            else:
                if isinstance(m, (nn.Conv2d, nn.InstanceNorm2d)):
                    _vendor_fallback("NLayerDiscriminator", f"layer {i} ({type(m).__name__}) has no HIP kernel for its shape")
                x = m(x)
                i += 1
        return x.permute(0, 3, 1, 2).contiguous() if nhwc else x


def resnetGenerator9():
    return ResnetGenerator(1, 1, ngf=64, n_blocks=9)


def patchGAN70x70():
    return NLayerDiscriminator(1, ndf=64, n_layers=3)


from .gan_seg_model import GanSegModel  # noqa: E402  (it receives MODEL_DICT through its constructor, like the reference's)

# reference models/networks.py:1009-1026 restricted to the hot path: the segmentation network, the contrast-adaptation
# generator / discriminator and the joint model; the paper's other GAN baselines are out of scope
MODEL_DICT = {"DynUNet": DynUNet, "resnetGenerator9": resnetGenerator9, "patchGAN70x70": patchGAN70x70, "GanSegModel": GanSegModel}
