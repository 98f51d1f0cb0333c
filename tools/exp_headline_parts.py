"""GPU only (diagnostic): the headline leg of bench.py with parts of the rasterisation left out -- what each part displaces in the
generator's cycle. usage: python tools/exp_headline_parts.py {full|nolabel|nodither|nodraw|simonly} [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:] + ["--no-train", "--no-files", "--no-pmc", "--no-cpu-baseline"]
import torch
from octa_autosegmentation_amd import pipeline
from octa_autosegmentation_amd.vessel_graph_generation import tree2img

TG = pipeline.TripleGenerator
_render, _plan = TG._render, TG._plan


def fake_out(self, res):
    B = self.batch
    return dict(result=res, image=torch.zeros((B, *self.image_res), dtype=torch.uint8, device=self.device),
                label=torch.zeros((B, *self.label_res), dtype=torch.uint8, device=self.device), label_grey=None)


if mode == "nolabel":
    def r(self, res, want_label, plans=None):
        out = _render(self, res, False, plans)
        out["label"] = torch.zeros((self.batch, *self.label_res), dtype=torch.uint8, device=self.device); out["label_grey"] = None
        return out
    TG._render = r
    TG._plan = lambda self, res, want_label: _plan(self, res, False)
elif mode == "nodither":
    tree2img.binarize_label_device = lambda grey: grey
elif mode == "nodraw":
    TG._render = lambda self, res, want_label, plans=None: fake_out(self, res)
elif mode == "simonly":
    TG._render = lambda self, res, want_label, plans=None: fake_out(self, res)
    TG._plan = lambda self, res, want_label: None
    TG.plan_ahead = False
import runpy
runpy.run_path(sys.argv[0], run_name="__main__")
