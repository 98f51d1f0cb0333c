"""GPU: DynUNet forward parity (fp32, same weights) between cuda and cpu within the 1e-4 of north_star,
and one bf16 training step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dynunet_logits_match_cpu_fp32():
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(0)
    net = networks.DynUNet()
    networks.init_weights(net, init_type="kaiming", nonlinearity="leaky_relu")
    x = torch.rand(2, 1, 96, 96)
    with torch.no_grad():
        ref = net(x)
        got = net.cuda()(x.cuda()).cpu()
    assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4), float((got - ref).abs().max())


def test_training_step_bf16_runs_and_learns():
    from tests.test_models import CFG
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    torch.manual_seed(1)
    tr = SegmentationTrainer(CFG, "cuda")
    x = torch.rand(2, 1, 128, 128, device="cuda")
    y = (x > 0.7).float()
    first = None
    for i in range(30):
        _, losses = tr.perform_training_step({"image": x, "label": y})
        v = float(losses["DiceBCELoss"])
        assert v == v
        first = v if first is None else first
    assert v < first
