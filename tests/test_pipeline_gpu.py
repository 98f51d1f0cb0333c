"""GPU: one batch through simulator -> image / label, every stage checked against the oracle."""
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu


def test_triples_match_oracle(hip_lib_built):
    import torch
    from octa_autosegmentation_amd import graph_io, pipeline
    from oracle import octa_oracle, sim_oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"] = 30
    cfg["Greenhouse"]["modes"][1]["I"] = 20
    gen = pipeline.TripleGenerator(cfg, 3)
    out = gen.generate([0, 1, 2])
    torch.cuda.synchronize()
    res = out["result"]
    for k in range(3):
        name = f"run_s{k}_30_20"
        edges = res.sample_edges(k)
        assert graph_io.edges_to_csv_text(edges).encode() == g[name + "_csv"].tobytes()
        e_or, info = sim_oracle.simulate(cfg, k)
        na = info["n_art_edges"]
        img = np.maximum(octa_oracle.rasterize(e_or[:na], [304, 304]), octa_oracle.rasterize(e_or[na:], [304, 304]))
        assert (out["image"][k].cpu().numpy() == img).all()
        grey = octa_oracle.rasterize(graph_io.edges_as_read_back(e_or), [1216, 1216])
        assert (out["label_grey"][k].cpu().numpy() == grey).all()
        assert (out["label"][k].cpu().numpy() == octa_oracle.fs_dither(grey)).all()
    gen.close()


def test_loader_image_mode_is_one_render_of_the_read_back_graph(hip_lib_built):
    """image_mode="loader" (the on-the-fly training loops): the image is what LoadGraphAndFilterByRandomRadiusd renders from the CSV file
    (data_transforms.py:376-386) -- ONE rasterisation of all edges as read back from the text, radius window included -- not the
    max of separate arterial / venous renders the generator CLI writes."""
    import torch
    from octa_autosegmentation_amd import graph_io, pipeline
    from octa_autosegmentation_amd.utils import configs
    from oracle import octa_oracle
    cfg = configs.load_generator_config()
    cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = 30, 20
    gen = pipeline.TripleGenerator(cfg, 2, label_min_radius=0.0033, image_mode="loader", image_min_radius=0.001)
    out = gen.generate([4, 5])
    torch.cuda.synchronize()
    for k in range(2):
        e = graph_io.edges_as_read_back(out["result"].sample_edges(k))
        assert (out["image"][k].cpu().numpy() == octa_oracle.rasterize(e[e[:, 6] >= 0.001], [304, 304])).all()
        assert (out["label_grey"][k].cpu().numpy() == octa_oracle.rasterize(e[e[:, 6] >= 0.0033], [1216, 1216])).all()
    gen.close()


def test_device_read_back_matches_host_emulation(hip_lib_built):
    import torch
    from octa_autosegmentation_amd import graph_io
    rng = np.random.default_rng(3)
    n = 20000
    e = np.zeros((n, 7))
    e[:, 0:3] = rng.uniform(-0.01, 1.02, (n, 3)); e[:, 3:6] = rng.uniform(-0.01, 1.02, (n, 3))
    e[:, 2] = rng.uniform(-2e-3, 0.0131, n); e[:, 5] = rng.uniform(-2e-3, 0.0131, n)
    e[::7, 2] = rng.uniform(-9e-5, 9e-5, n)[::7]; e[::11, 0] = 0.0; e[::13, 4] = 1 - 1e-6
    e[5, 0:3] = [0.5, 0.25, 0.125]; e[6, 3:6] = [1e-5, 2e-6, 3e-9]; e[7, 0:3] = [0.99999999499, 0.123456785, 0.000123456789]
    e[:, 6] = rng.uniform(1e-4, 1e-2, n)
    got = graph_io.edges_as_read_back_device(torch.from_numpy(e).cuda()).cpu().numpy()
    want = graph_io.edges_as_read_back(e)
    assert (got == want).all(), np.nonzero((got != want).any(axis=1))[0][:10]


@pytest.mark.parametrize("gated", [False, True, "planned"])
def test_three_steps_in_flight_are_independent(hip_lib_built, gated):
    """bench.py keeps several steps in flight from a thread pool: each thread must get its own scratch context
    and stream, and every image / label must still be the oracle's -- with all persistent kernels resident at once and with
    bench.py's default, one persistent kernel at a time (TripleGenerator.sim_gate) while the other launches are rasterised; "planned":
    the gated form with both rasterisations planned ahead (octa_rasterize_2d_plan / _draw on two contexts per slot)."""
    import threading
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from octa_autosegmentation_amd import graph_io, pipeline
    from oracle import octa_oracle, sim_oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"] = 14
    cfg["Greenhouse"]["modes"][1]["I"] = 8
    B, n_fly = 4, 3
    gens = [pipeline.TripleGenerator(cfg, B) for _ in range(n_fly)]
    if gated:
        gate = pipeline.SimGate()
        for gen in gens:
            gen.sim_gate = gate
            gen.plan_ahead = gated == "planned"
    streams = [torch.cuda.Stream() for _ in range(n_fly)]
    dev = torch.cuda.current_device()

    def work(slot):
        torch.cuda.set_device(dev)
        outs = []
        with torch.cuda.stream(streams[slot]):
            for rep in range(3):
                seeds = np.arange(B) + 10 * slot + 100 * rep
                out = gens[slot].generate(seeds)
                streams[slot].synchronize()
                outs.append((seeds, out["image"].cpu().numpy(), out["label"].cpu().numpy(), out["result"]))
        return outs

    with ThreadPoolExecutor(n_fly) as ex:
        results = list(ex.map(work, range(n_fly)))
    for outs in results:
        for seeds, image, label, res in outs[-1:]:
            for k, s in enumerate(seeds):
                e, info = sim_oracle.simulate(cfg, int(s))
                na = info["n_art_edges"]
                assert graph_io.edges_to_csv_text(res.sample_edges(k)) == sim_oracle.edges_to_csv_text(e)
                img = np.maximum(octa_oracle.rasterize(e[:na], [304, 304]), octa_oracle.rasterize(e[na:], [304, 304]))
                assert (image[k] == img).all()
                grey = octa_oracle.rasterize(graph_io.edges_as_read_back(e), [1216, 1216])
                assert (label[k] == octa_oracle.fs_dither(grey)).all()
    for gen in gens:
        gen.close()


def test_on_the_fly_training_runs(hip_lib_built):
    """train_synthetic.py: generator thread (simulate + rasterise) -> GPU augmentation -> training step, a few steps."""
    import train_synthetic
    res = train_synthetic.run(steps=4, batch=2, gen_batch=8, warmup=1, log=False)
    assert res["value"] > 0 and np.isfinite(res["last_loss"]) and res["n_gpus"] == 1


def test_on_the_fly_gan_seg_training_runs(hip_lib_built):
    """train_synthetic.py --gan (BASELINE configs[4]): the same stream feeding the joint GAN + segmentation step. 20 steps: the
    MIOpen solver fault this loop exposed (models/gan_seg_trainer.py) needed about twelve steps to show."""
    import train_synthetic
    res = train_synthetic.run(steps=20, batch=2, gen_batch=16, warmup=1, log=False, gan=True)
    assert res["value"] > 0 and np.isfinite(res["first_loss"]) and np.isfinite(res["last_loss"]) and res["n_gpus"] == 1


def test_order_gate_waits_for_a_launch_on_the_device_and_gives_up_at_its_timeout():
    """csrc/order.hip: the one-wave gate kernel a rasterisation's stream waits in until the NEXT persistent-kernel launch is resident (round 6:
    the order is kept on the device; round 5 polled a counter from Python and slept). A ticket that has been launched passes at once with its
    workgroups signed in; a ticket nobody launches is given up after the timeout; a launch made WHILE the gate waits releases it. The gate
    and the launch sit on streams of different queue priority, as in pipeline.TripleGenerator: two streams of EQUAL priority may share a
    hardware queue, and the spinning gate would then hold the very launch it waits for behind itself until its time-out (seen in the full
    test suite, where earlier tests had used up the queues)."""
    import threading
    import time
    import torch
    from octa_autosegmentation_amd import _native
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"] = 6
    cfg["Greenhouse"]["modes"][1]["I"] = 3
    lib, ctx = _native.lib(), _native.ctx(0)
    sim = greenhouse.BatchSimulator(cfg, 8, 0)
    out = torch.zeros(3, dtype=torch.int32, device="cuda")
    wait = lambda ticket, timeout_us, st: _native.check(lib.octa_order_wait_launch(ctx, ticket, timeout_us, 100, out.data_ptr(), st.cuda_stream), "octa_order_wait_launch")
    hi = torch.cuda.Stream(priority=-1)

    def run(first_seed):
        torch.cuda.set_device(0)
        with torch.cuda.stream(hi):
            sim.run(np.arange(8) + first_seed)
    try:
        run(11)
        t_mine = lib.octa_sim_launch_count()
        st = torch.cuda.Stream(priority=0)
        wait(t_mine, 2_000_000, st)                         # already launched (and over): passes at once, all 8 workgroups had signed in
        st.synchronize()
        state, ticks, signed = out.cpu().tolist()
        assert state == 1 and signed == 8 and ticks < 100_000, (state, ticks, signed)
        t0 = time.time()
        wait(t_mine + 1, 30_000, st)                        # nobody launches ticket + 1: given up after 30 ms
        st.synchronize()
        state, ticks, _ = out.cpu().tolist()
        assert state == 2 and 2_500_000 <= ticks <= 5_000_000 and time.time() - t0 < 1.0, (state, ticks)
        wait(t_mine + 1, 5_000_000, st)                     # ... and a launch made while the gate waits releases it
        th = threading.Thread(target=lambda: (time.sleep(0.05), run(19)))
        th.start()
        st.synchronize()
        th.join()
        state, ticks, signed = out.cpu().tolist()
        assert state == 1 and signed >= 6 and 3_000_000 <= ticks <= 200_000_000, (state, ticks, signed)
        assert lib.octa_sim_launch_count() == t_mine + 1
    finally:
        sim.close()
