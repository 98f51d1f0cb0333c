"""Joint contrast adaptation + segmentation (reference models/gan_seg_model.py:12-196, config configs/config_gan_ves_seg.yml):
generator G (synthetic -> real contrast), PatchGAN D, segmentor S, three Adam optimisers (G, D: betas (0.5, 0.999); S:
(0.9, 0.999)). Per step (gan_seg_model.py:110-173):

  D:    fake_B = G(real_A), idt_B = G(real_B);  loss_D = (LSGAN(D(fake_B.detach()), 0) + LSGAN(D(real_B), 1)) / 2
  G+S:  loss_G = LSGAN(D(fake_B), 1) [+ L1(idt_B, real_B)];  S sees the images bilinearly upsampled to `upshape`;
        loss_S = DiceBCE(S(fake_B), real_A_seg);  loss_S_idt = DiceBCE(S(idt_B), threshold_0.5(S(real_B)));
        loss_GS = loss_G + (loss_S + loss_S_idt) / 2

MI355X formulation: bf16 autocast; passes that share weights run as ONE launch sequence over the concatenated batch --
G(real_A | real_B), D(fake_B | real_B), S(idt_B | fake_B): every network normalises per sample (InstanceNorm), so each
half is what the reference's separate call gives, with half the launches on the launch-bound 76x76 stages; the
pseudo-label pass S(real_B) runs under no_grad (the reference thresholds it in place, which detaches it just the same).
Gradients of D after its backward, and of G and S after theirs, are averaged across ranks with one all-reduce per
flat gradient arena (base_model_abc.GradArena)."""
from typing import Any, Callable, Dict, Tuple

import torch
from torch import nn

from ..utils.enums import Phase
from .base_model_abc import BaseModelABC, LossValues
from .lambda_model import decollate_batch
from .losses import get_loss_function_by_name
from .model_interface_abc import Output


_D_SIDE = {}          # device index -> (side stream of the discriminator chain, its octa context)


class GanSegModel(BaseModelABC):
    def __init__(self, MODEL_DICT: dict, model_g: dict, model_d: dict, model_s: dict, compute_identity=True, compute_identity_seg=True,
                 phase: Phase = Phase.TRAIN, inference: str = None, upshape: Tuple[int, int] = (1216, 1216), **kwargs):
        super().__init__(optimizer_mapping={"optimizer_G": ["generator"], "optimizer_D": ["discriminator"], "optimizer_S": ["segmentor"]},
                         optimizer_configs={"optimizer_S": {"betas": (0.9, 0.999)}}, **kwargs)
        self.segmentor: nn.Module = None
        self.generator: nn.Module = None
        self.discriminator: nn.Module = None
        model_g, model_d, model_s = dict(model_g), dict(model_d), dict(model_s)
        if phase == Phase.TRAIN or inference in ("S", "segmentor"):
            self.segmentor = MODEL_DICT[model_s.pop("name")](**model_s)
        if phase == Phase.TRAIN or inference in ("G", "generator"):
            self.generator = MODEL_DICT[model_g.pop("name")](**model_g)
        if phase == Phase.TRAIN:
            self.discriminator = MODEL_DICT[model_d.pop("name")](**model_d)
        self.compute_identity = compute_identity
        self.compute_identity_seg = compute_identity_seg
        self.criterionIdt = torch.nn.L1Loss()
        self.upshape = tuple(upshape)

    def initialize_model_and_optimizer(self, init_mini_batch: dict, init_weights: Callable, config: dict, args, scaler,
                                       phase: Phase = Phase.TRAIN):
        if phase != Phase.TEST:
            self.loss_name_dg = config[Phase.TRAIN]["loss_dg"]
            self.loss_name_s = config[Phase.TRAIN]["loss_s"]
            self.dg_loss = get_loss_function_by_name(self.loss_name_dg, config)
            self.s_loss = get_loss_function_by_name(self.loss_name_s, config)
        super().initialize_model_and_optimizer(init_mini_batch, init_weights, config, args, scaler, phase)

    def _up(self, x):
        if x.is_cuda:          # csrc/augment.hip both ways (round 4: the ATen interpolate kernels were the last vendor kernels of the up-sampling)
            from ..data.gpu_augment import BilinearResize
            return BilinearResize.apply(x, tuple(self.upshape))
        return torch.nn.functional.interpolate(x, size=self.upshape, mode="bilinear")

    # OCTA_GAN_STREAMS=0: the whole step on the current stream (the order of the reference's optimize_parameters)
    def _side_stream(self, device):
        import os
        if device.type != "cuda" or os.environ.get("OCTA_GAN_STREAMS", "1") == "0":
            return None, None, None
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx not in _D_SIDE:                                   # one per device for the life of the process (like utils/aside.py)
            from .. import _native
            _D_SIDE[idx] = (torch.cuda.Stream(device=idx), _native.new_ctx(idx))
        return _D_SIDE[idx][0], _D_SIDE[idx][1], torch.cuda.current_stream(device)

    def forward(self, input: torch.Tensor):
        if self.segmentor is not None:
            return self.segmentor(self._up(input))
        return self.generator(input)

    def inference(self, mini_batch: Dict[str, Any], post_transformations: Dict[str, Callable], device: torch.device = "cpu",
                  phase: Phase = Phase.TEST) -> Tuple[Output, Dict[str, torch.Tensor]]:
        assert phase == Phase.VALIDATION or phase == Phase.TEST, "This forward function only supports val and test. Use perform_step for training"
        input: torch.Tensor = mini_batch["image"].to(device=device, non_blocking=True)
        pred = self.forward(input)
        losses = dict()
        outputs: Output = {"prediction": [post_transformations["prediction"](i) for i in decollate_batch(pred[0:1, 0:1])]}
        if self.segmentor is not None and phase == Phase.VALIDATION:
            labels: torch.Tensor = mini_batch["label"].to(device=device, non_blocking=True)
            outputs["label"] = [post_transformations["label"](i) for i in decollate_batch(labels[0:1, 0:1])]
            losses[self.loss_name_s] = self.s_loss(pred.float(), labels.float())
        return outputs, losses

    def perform_training_step(self, mini_batch: Dict[str, Any], scaler, post_transformations: Dict[str, Callable],
                              device: torch.device = "cpu") -> Tuple[Output, Dict[str, float]]:
        real_A: torch.Tensor = mini_batch["real_A"].to(device, non_blocking=True)
        real_B: torch.Tensor = mini_batch["real_B"].to(device, non_blocking=True)
        real_A_seg: torch.Tensor = mini_batch["real_A_seg"].to(device, non_blocking=True)
        nA = real_A.shape[0]
        side, sctx, main = self._side_stream(real_A.device)
        # ---- generator forward (root of both updates)
        with self.autocast():
            if self.compute_identity_seg or self.compute_identity:
                g_both = self.generator(torch.cat((real_A, real_B), dim=0))
                fake_B, idt_B = g_both[:nA], g_both[nA:]
            else:
                fake_B, idt_B = self.generator(real_A), None
        # ---- discriminator update, then D(fake_B) with the updated, frozen discriminator: a chain of ~250 short dependent launches
        # (4x4 convolutions at 304^2 and below) that nothing of the segmentor's passes depends on. With two streams it runs BESIDE
        # the segmentor's pseudo-label pass and its forward pass over (idt_B | fake_B) -- MFMA-heavy launches that leave the gaps
        # between D's launches unused otherwise. Measured: 55.3 -> 54.0 ms (profiles/r05_stream_overlap_ab.log). Starting the pseudo-label
        # pass on the side stream beside the GENERATOR's forward as well was measured too and bought nothing more (54.9 against 56.1 with one
        # stream on a slower box: the same 1.2 ms) -- the segmentor's launches fill the GPU whoever runs next to them.
        def d_chain():
            self.zero_grads("optimizer_D")
            with self.autocast():
                self.discriminator.requires_grad_(True)
                d_both = self.discriminator(torch.cat((fake_B.detach(), real_B), dim=0)).float()
                l_fake = self.dg_loss(d_both[:nA], False)
                l_real = self.dg_loss(d_both[nA:], True)
                l_d = 0.5 * (l_fake + l_real)
            with self.backward_scope():
                l_d.backward()
            self.exchange_gradients("optimizer_D")
            self.optimizer_D.step()
            with self.autocast():
                self.discriminator.requires_grad_(False)
                p_fake = self.discriminator(fake_B)
                l_g = self.dg_loss(p_fake.float(), True)
            return l_fake, l_real, l_g

        if side is not None:
            from .. import _native
            side.wait_stream(main)
            for t in (fake_B, real_B):
                t.record_stream(side)
            with torch.cuda.stream(side), _native.use_ctx(sctx):
                loss_D_fake, loss_D_real, loss_G = d_chain()
        else:
            loss_D_fake, loss_D_real, loss_G = d_chain()
        # ---- segmentor passes (current stream)
        self.zero_grads("optimizer_G")
        self.zero_grads("optimizer_S")
        with self.autocast():
            with torch.no_grad():
                real_B_seg = (self.segmentor(self._up(real_B)) > 0.5).float()          # pseudo-labels (gan_seg_model.py:133-134)
            if self.compute_identity_seg:
                both = self.segmentor(torch.cat((self._up(idt_B), self._up(fake_B)), dim=0))
                idt_B_seg, fake_B_seg = both[:real_B.shape[0]], both[real_B.shape[0]:]
            else:
                idt_B_seg, fake_B_seg = None, self.segmentor(self._up(fake_B))
            if side is not None:
                main.wait_stream(side)
                for t in (loss_D_fake, loss_D_real, loss_G):
                    t.record_stream(main)
            zero = torch.zeros((), device=real_A.device)
            loss_G_idt = self.criterionIdt(idt_B.float(), real_B.float()) if self.compute_identity else zero
            loss_G = loss_G + loss_G_idt
            loss_S = self.s_loss(fake_B_seg.float(), real_A_seg.float())
            if self.compute_identity_seg:
                loss_S_idt = self.s_loss(idt_B_seg.float(), real_B_seg)
                loss_SS = 0.5 * (loss_S + loss_S_idt)
            else:
                loss_S_idt = zero
                loss_SS = loss_S
            loss_GS = loss_G + loss_SS
        with self.backward_scope():
            loss_GS.backward()
        if side is not None:
            main.wait_stream(side)          # the discriminator branch of the backward pass ran on the side stream
        self.discriminator.requires_grad_(True)
        self.exchange_gradients("optimizer_G", "optimizer_S")
        self.optimizer_G.step()
        self.optimizer_S.step()
        pt = post_transformations or {}
        pp = pt.get("prediction") or (lambda t: t)
        pl = pt.get("label") or (lambda t: t)
        outputs: Output = {
            "prediction": [pp(i) for i in decollate_batch(fake_B_seg[0:1, 0:1].detach())],
            "label": [pl(i) for i in decollate_batch(real_A_seg[0:1, 0:1])],
            "fake_B": fake_B[0:1, 0:1].detach(),
            "idt_B": None if idt_B is None else idt_B[0:1, 0:1].detach(),
            "real_B_seg": real_B_seg,
        }
        losses = LossValues({"S": loss_S.detach(), "D_fake": loss_D_fake.detach(), "D_real": loss_D_real.detach(), "G": loss_G.detach(),
                             "G_idt": loss_G_idt.detach(), "S_idt": loss_S_idt.detach()})
        return outputs, losses

    def plot_sample(self, visualizer, mini_batch: Dict[str, Any], outputs: Output, *, suffix: str = ""):
        if "fake_B" in outputs:
            return visualizer.plot_gan_seg_sample(mini_batch["real_A"][0], outputs["fake_B"][0], outputs["prediction"][0], mini_batch["real_B"][0],
                                                  None if outputs["idt_B"] is None else outputs["idt_B"][0], outputs["real_B_seg"][0],
                                                  path_A=mini_batch["real_A_path"][0], path_B=mini_batch["real_B_path"][0], suffix=suffix)
        return visualizer.plot_sample(mini_batch["image"][0], outputs["prediction"][0], outputs["label"][0],
                                      path=mini_batch["image_path"][0], suffix=suffix)
