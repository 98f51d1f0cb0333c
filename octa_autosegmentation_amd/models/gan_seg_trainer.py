"""Joint contrast-adaptation + segmentation step (reference models/gan_seg_model.py:12-173, config
configs/config_gan_ves_seg.yml): generator G, PatchGAN D, segmentor S with three Adam optimisers
(G, D: betas (0.5, 0.999); S: (0.9, 0.999)), LSGAN + identity-L1 for G, DiceBCE for S on
bilinearly upsampled fake / identity images, pseudo-labels from thresholding S(real_B) at 0.5.
bf16 autocast; with torch.distributed initialised each optimiser's gradients are averaged with one flat
RCCL all-reduce (D after its backward; G and S together after theirs)."""
import itertools
import os

# The four layers of G and D that are not matrix-core shaped (7x7 stem / head of the generator, 4x4 stem / head of the
# PatchGAN) stay torch convolutions. With the GEMM solvers off (segmentation_trainer.py sets MIOPEN_DEBUG_CONV_GEMM=0 for
# its fp32 reference path) MIOpen's immediate mode answers the PatchGAN head's data gradient (512 -> 1 channels, NHWC bf16)
# with its asm implicit-GEMM NHWC kernel WITHOUT the 11 MB workspace that kernel needs ("workspace required: 11214848,
# provided ptr: 0") and the kernel faults a few steps later (rocgdb: igemm_bwd_gtcx35_nhwc_bf16_... memory violation in the
# on-the-fly GAN-seg loop). That solver is switched off here, before MIOpen reads its environment.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC", "0")    # same "provided ptr: 0" warning for this one

import torch
import torch.distributed as dist

from .losses import get_loss_function_by_name
from .networks import MODEL_DICT, init_weights


def _flat_allreduce(params):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


class GanSegTrainer:
    optimizer_mapping = {"optimizer_G": ["generator"], "optimizer_D": ["discriminator"], "optimizer_S": ["segmentor"]}

    def __init__(self, config, device, upshape=(1216, 1216)):
        m = config["General"]["model"]
        self.device = torch.device(device)
        mk = lambda d: MODEL_DICT[dict(d).pop("name")](**{k: v for k, v in d.items() if k != "name"})
        self.generator = mk(m["model_g"]).to(self.device)
        self.discriminator = mk(m["model_d"]).to(self.device)
        self.segmentor = mk(m["model_s"]).to(self.device)
        init_weights(self.generator, "kaiming", nonlinearity="relu")
        init_weights(self.discriminator, "kaiming", nonlinearity="leaky_relu")
        init_weights(self.segmentor, "kaiming", nonlinearity="leaky_relu")
        self.compute_identity = m.get("compute_identity", True)
        self.compute_identity_seg = m.get("compute_identity_seg", True)
        self.upshape = tuple(m.get("upshape", upshape))
        tr = config["Train"]
        lr = tr["lr"]
        self.optimizer_G = torch.optim.Adam(self.generator.parameters(), lr=lr, betas=(0.5, 0.999))
        self.optimizer_D = torch.optim.Adam(self.discriminator.parameters(), lr=lr, betas=(0.5, 0.999))
        self.optimizer_S = torch.optim.Adam(self.segmentor.parameters(), lr=lr, betas=(0.9, 0.999))
        self.dg_loss = get_loss_function_by_name(tr.get("loss_dg", "LSGANLoss"), config)
        self.s_loss = get_loss_function_by_name(tr.get("loss_s", "DiceBCELoss"), config)
        self.l1 = torch.nn.L1Loss()
        self.amp = bool(config["General"].get("amp", True)) and self.device.type == "cuda"
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in itertools.chain(self.generator.parameters(), self.discriminator.parameters(), self.segmentor.parameters()):
                dist.broadcast(p.data, src=0)

    def _up(self, x):
        return torch.nn.functional.interpolate(x, size=self.upshape, mode="bilinear")

    def perform_training_step(self, mini_batch, scaler=None, post_transformations=None, device=None):
        real_A = mini_batch["real_A"].to(self.device, non_blocking=True)
        real_B = mini_batch["real_B"].to(self.device, non_blocking=True)
        real_A_seg = mini_batch["real_A_seg"].to(self.device, non_blocking=True)
        ac = lambda: torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=self.amp)
        # ---- D step
        self.optimizer_D.zero_grad(set_to_none=True)
        with ac():
            # G(real_A) and G(real_B), D(fake_B) and D(real_B): one pass each over the concatenated batch (both networks normalise per
            # sample, so the halves are what the reference's separate calls give; the 76x76 residual stages are launch-bound at B = 4)
            if self.compute_identity or self.compute_identity_seg:
                g_both = self.generator(torch.cat((real_A, real_B), dim=0))
                fake_B, idt_B = g_both[:real_A.shape[0]], g_both[real_A.shape[0]:]
            else:
                fake_B, idt_B = self.generator(real_A), None
            self.discriminator.requires_grad_(True)
            d_both = self.discriminator(torch.cat((fake_B.detach(), real_B), dim=0)).float()
            loss_D_fake = self.dg_loss(d_both[:fake_B.shape[0]], False)
            loss_D_real = self.dg_loss(d_both[fake_B.shape[0]:], True)
            loss_D = 0.5 * (loss_D_fake + loss_D_real)
        loss_D.backward()
        _flat_allreduce(list(self.discriminator.parameters()))
        self.optimizer_D.step()
        # ---- G + S step
        self.optimizer_G.zero_grad(set_to_none=True)
        self.optimizer_S.zero_grad(set_to_none=True)
        with ac():
            self.discriminator.requires_grad_(False)
            pred_fake_B = self.discriminator(fake_B)
            with torch.no_grad():                      # only its thresholded, detached output is used (gan_seg_model.py: pseudo-labels)
                real_B_seg = self.segmentor(self._up(real_B))
            if self.compute_identity_seg:
                # S(idt_B) and S(fake_B) as ONE pass over the concatenated batch: InstanceNorm is per sample, so every logit and every
                # gradient is what the reference's two passes give (fp32 summation order of the weight gradients aside), with half
                # the launches, one weight-gradient pass and no second gradient accumulation
                both = self.segmentor(torch.cat((self._up(idt_B), self._up(fake_B)), dim=0))
                idt_B_seg, fake_B_seg = both[:idt_B.shape[0]], both[idt_B.shape[0]:]
            else:
                idt_B_seg, fake_B_seg = None, self.segmentor(self._up(fake_B))
            pseudo = (real_B_seg > 0.5).float()
            loss_G = self.dg_loss(pred_fake_B.float(), True)
            loss_G_idt = self.l1(idt_B.float(), real_B.float()) if self.compute_identity else torch.zeros((), device=self.device)
            loss_G = loss_G + loss_G_idt
            loss_S = self.s_loss(fake_B_seg.float(), real_A_seg.float())
            if self.compute_identity_seg:
                loss_S_idt = self.s_loss(idt_B_seg.float(), pseudo)
                loss_SS = 0.5 * (loss_S + loss_S_idt)
            else:
                loss_S_idt = torch.zeros((), device=self.device)
                loss_SS = loss_S
            loss_GS = loss_G + loss_SS
        loss_GS.backward()
        _flat_allreduce(list(itertools.chain(self.generator.parameters(), self.segmentor.parameters())))
        self.optimizer_G.step()
        self.optimizer_S.step()
        outputs = {"prediction": fake_B_seg[0:1, 0:1].detach(), "label": real_A_seg[0:1, 0:1], "fake_B": fake_B[0:1, 0:1].detach(),
                   "idt_B": None if idt_B is None else idt_B[0:1, 0:1].detach(), "real_B_seg": pseudo}
        losses = {"S": loss_S, "D_fake": loss_D_fake, "D_real": loss_D_real, "G": loss_G, "G_idt": loss_G_idt, "S_idt": loss_S_idt}
        return outputs, losses
