"""tools/make_monai_manifest.py -- writes tests/golden/monai_dynunet_S_manifest.json.

MONAI (pinned >= 1.5.0 by the reference's pyproject.toml:11; imported at models/networks.py:6 and built at :1010) is neither in this
image nor under /root/reference, so DynUNet's numerics cannot be pinned against it. What CAN be pinned is the
checkpoint contract: the keys and shapes of `monai.networks.nets.DynUNet(...).state_dict()` for the S configuration
(configs/config_ves_seg-S.yml:6-13: spatial_dims 2, in/out channels 1, kernel_size [3]*5, strides [1,2,2,2,1], upsample_kernel_size
[1,2,2,2,1], default filters, instance norm, no deep supervision, no residual blocks). This script derives them WITHOUT importing the
repository's own network -- from the module structure of MONAI's published source, restated here as a small description:

  monai/networks/nets/dynunet.py, class DynUNet.__init__: attributes are assigned in the order input_block, downsamples, bottleneck,
    upsamples, output_block, (deep_supervision_heads only with deep_supervision=True), then skip_layers = create_skips(0, [input_block] +
    list(downsamples), upsamples[::-1], bottleneck); default filters = [min(2 ** (5 + i), 320 if spatial_dims == 3 else 512) ...];
    get_downsamples: strides[1:-1] / kernel_size[1:-1]; get_bottleneck: filters[-2] -> filters[-1] with strides[-1]; get_upsamples: inp =
    filters[1:][::-1], out = filters[:-1][::-1], kernels kernel_size[1:][::-1], upsample_kernel_size[::-1] (one per up block).
  monai/networks/nets/dynunet.py, class DynUNetSkipLayer.__init__: attributes downsample, next_layer, upsample (super_head / heads hold no
    parameters without deep supervision); create_skips returns the bottleneck block itself at the bottom of the recursion.
  monai/networks/blocks/dynunet_block.py: UnetBasicBlock (conv1, conv2, lrelu, norm1, norm2), UnetUpBlock (transp_conv, conv_block =
    UnetBasicBlock(out + out, out)), UnetOutBlock (conv, built with bias=True); get_conv_layer -> Convolution(..., bias=False,
    conv_only=True) whose only child is registered as "conv" (monai/networks/blocks/convolutions.py: self.add_module("conv", conv)).
  norm_name ("INSTANCE", {"affine": True}) -> torch.nn.InstanceNorm2d(affine=True): weight + bias, no running statistics.

(Restated from the published MONAI 1.x sources as the author knows them; there is no network here to quote line numbers from. The
parameter count it implies, 7 368 769, is the one SURVEY.md a18 records for the reference's S configuration.)

  python tools/make_monai_manifest.py
"""
import json
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "monai_dynunet_S_manifest.json")


def conv(prefix, cin, cout, k, transposed=False, bias=False):
    shape = [cin, cout, k, k] if transposed else [cout, cin, k, k]
    rows = [(prefix + ".conv.weight", shape)]
    if bias:
        rows.append((prefix + ".conv.bias", [cout]))
    return rows


def basic_block(prefix, cin, cout, k):
    return (conv(prefix + ".conv1", cin, cout, k) + conv(prefix + ".conv2", cout, cout, k)
            + [(prefix + ".norm1.weight", [cout]), (prefix + ".norm1.bias", [cout]), (prefix + ".norm2.weight", [cout]), (prefix + ".norm2.bias", [cout])])


def up_block(prefix, cin, cout, k, up_k):
    return conv(prefix + ".transp_conv", cin, cout, up_k, transposed=True) + basic_block(prefix + ".conv_block", cout + cout, cout, k)


def main():
    spatial_dims, in_ch, out_ch = 2, 1, 1
    kernel_size, strides, up_kernel = [3, 3, 3, 3, 3], [1, 2, 2, 2, 1], [1, 2, 2, 2, 1]
    filters = [min(2 ** (5 + i), 320 if spatial_dims == 3 else 512) for i in range(len(strides))]
    blocks = {"input_block": basic_block("input_block", in_ch, filters[0], kernel_size[0])}
    downs = []
    for i, (cin, cout, k) in enumerate(zip(filters[:-2], filters[1:-1], kernel_size[1:-1])):
        downs.append(basic_block(f"downsamples.{i}", cin, cout, k))
    bottleneck = basic_block("bottleneck", filters[-2], filters[-1], kernel_size[-1])
    inp, out = filters[1:][::-1], filters[:-1][::-1]
    ups = [up_block(f"upsamples.{i}", ci, co, k, uk) for i, (ci, co, k, uk) in enumerate(zip(inp, out, kernel_size[1:][::-1], up_kernel[::-1]))]
    output = conv("output_block.conv", filters[0], out_ch, 1, bias=True)
    rows = blocks["input_block"] + [r for d in downs for r in d] + bottleneck + [r for u in ups for r in u] + output

    # skip_layers: the same modules once more under the recursive wrapper (shared tensors; state_dict() lists both names)
    def re_prefix(block_rows, old, new):
        return [(new + name[len(old):], shape) for name, shape in block_rows]

    def skips(prefix, down_list, up_list):
        if not down_list:
            return re_prefix(bottleneck, "bottleneck", prefix[:-1])          # the bottleneck block itself stands at `next_layer`
        (d_old, d_rows), (u_old, u_rows) = down_list[0], up_list[0]
        return (re_prefix(d_rows, d_old, prefix + "downsample") + skips(prefix + "next_layer.", down_list[1:], up_list[1:])
                + re_prefix(u_rows, u_old, prefix + "upsample"))

    down_list = [("input_block", blocks["input_block"])] + [(f"downsamples.{i}", d) for i, d in enumerate(downs)]
    up_list = [(f"upsamples.{i}", u) for i, u in enumerate(ups)][::-1]
    rows += skips("skip_layers.", down_list, up_list)
    n_params = 0
    seen = set()
    for name, shape in rows:
        if not name.startswith("skip_layers."):
            n = 1
            for s in shape:
                n *= s
            n_params += n
        assert name not in seen
        seen.add(name)
    manifest = {"source": "MONAI >= 1.5.0 monai.networks.nets.DynUNet, S configuration of configs/config_ves_seg-S.yml:6-13 (derivation: tools/make_monai_manifest.py)",
                "ctor": {"spatial_dims": spatial_dims, "in_channels": in_ch, "out_channels": out_ch, "kernel_size": kernel_size, "strides": strides,
                         "upsample_kernel_size": up_kernel, "norm_name": ["INSTANCE", {"affine": True}]},
                "n_parameters": n_params, "keys": [[n, s] for n, s in rows]}
    with open(OUT, "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", os.path.abspath(OUT), len(rows), "keys,", n_params, "parameters")


if __name__ == "__main__":
    main()
