"""GPU: the mailbox round trip seen from the device (diagnostic build -DOCTA_SIM_PROF_MAIL: python tools/build_sim_variant.py profmail
-DOCTA_SIM_PROF_MAIL, then OCTA_HIP_LIB=gpurun_variants/liboctahip_profmail.so python tools/sim_mailbox_profile.py [batch ...]): per round
trip, the time spent publishing the request (fences + stores), the polls and the wait for the host's answer."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octa_autosegmentation_amd.utils import configs  # noqa: E402
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse  # noqa: E402

for B in [int(a) for a in sys.argv[1:]] or [512]:
    sim = greenhouse.BatchSimulator(configs.load_generator_config(), B)
    for rep in range(2):
        res = sim.run(np.arange(B) + 5000 + 1000 * rep)
    st = res.stats.astype(np.float64)
    kd = st[:, 24:32]
    rt = kd[:, 2].sum()
    print(f"B={B}: kernel {res.timing['kernel_b_ms']:.1f} ms; per round trip: wait {st[:, 13].sum() / rt * 1e-2:.1f} us, publish {kd[:, 0].sum() / rt * 1e-2:.2f} us, "
          f"polls {kd[:, 1].sum() / rt:.1f}; round trips per sample {rt / B:.1f}, wait per sample {st[:, 13].mean() * 1e-5:.2f} ms")
    sim.close()
