"""GPU only: which samples of a 512-sample launch are the slow ones, and in which phase (the launch lasts as long as its slowest sample)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
NAMES = ["sample", "assign_art", "pre_art", "seq_art", "satisfy_art", "mailbox", "assign_ven", "pre_ven", "seq_ven", "satisfy_ven",
         "candidates*", "kd_total*", "pairs+ven*", "pair_sort*", "set_replay*", "compact*"]
B = 512
sim = greenhouse.BatchSimulator(configs.load_generator_config(), B)
res = sim.run(np.arange(B) + 5000)
st = res.stats.astype(np.float64)
ph = st[:, 8:24] * 1e-5
tot = ph[:, :10].sum(axis=1)
print("per-sample total: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f  min %.1f; kernel %.1f ms" % (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.min(), res.timing['kernel_b_ms']))
mean = ph.mean(axis=0)
for k in np.argsort(-tot)[:8]:
    d = ph[k] - mean
    top = np.argsort(-d[:10])[:3]
    print("sample %3d total %.1f (+%.1f): " % (k, tot[k], tot[k] - tot.mean()) + ", ".join("%s +%.1f" % (NAMES[j], d[j]) for j in top) + "  | draws %d murray steps %d bif %d respec %d nodes %d + %d" % tuple(int(st[k, c]) for c in (1, 2, 3, 4, 5, 6)))
seq = ph[:, 3] + ph[:, 8]
for name, col in (("py draws", 1), ("murray steps", 2), ("bifurcations", 3), ("re-speculations", 4), ("arterial nodes", 5), ("venous nodes", 6)):
    print("corr(ordered passes, %s) = %.3f   corr(total, %s) = %.3f   mean %.0f  max %.0f" % (name, np.corrcoef(seq, st[:, col])[0, 1], name, np.corrcoef(tot, st[:, col])[0, 1], st[:, col].mean(), st[:, col].max()))
print("murray(seq) timer: mean %.1f max %.1f ms; corr with ordered passes %.3f" % ((st[:, 31] * 1e-5).mean(), (st[:, 31] * 1e-5).max(), np.corrcoef(seq, st[:, 31])[0, 1]))
sim.close()
