"""CPU: the bookkeeping of pipeline.SimGate (round 6) -- which events a launch is ordered behind, who counts as a waiting successor, and how a
rasterisation without successor takes the gate. The device side (the gate kernel, the sign-in block) is tests/test_pipeline_gpu.py."""
import threading
import time

from octa_autosegmentation_amd.pipeline import SimGate


class _Stream:
    def __init__(self):
        self.waited = []

    def wait_event(self, ev):
        self.waited.append(ev)


def test_a_launch_waits_for_the_kernels_that_must_be_off_the_gpu_when_it_is_placed():
    """Export / planning events of the launch before, rasterisations older than the last one, a rasterisation that went ahead without a
    successor: each exactly once; the LAST rasterisation is not waited for (its gate kernel holds it back until this launch is resident)."""
    g = SimGate()
    with g:
        s1 = _Stream()
        n1 = g.order_launch(s1)
        assert n1 == 1 and s1.waited == []
        g.after_launch("export1", "plan1")
    g.note_render(n1, "render1")
    with g:
        s2 = _Stream()
        n2 = g.order_launch(s2)
        assert n2 == 2 and s2.waited == ["export1", "plan1"]        # render1 belongs to the launch before: held back by its own gate kernel
        g.after_launch("export2", None)
    g.note_render(n2, "render2")
    g.set_barrier("render_without_successor")
    with g:
        s3 = _Stream()
        assert g.order_launch(s3) == 3
        assert s3.waited == ["export2", "render1", "render_without_successor"]
    with g:
        s4 = _Stream()
        g.order_launch(s4)
        assert s4.waited == ["render2"]                              # everything else was handed out once


def test_waiting_counts_launchers_and_a_render_gives_way_to_them():
    g = SimGate()
    assert g.enter_for_render() is True and g.waiting() == 0         # nobody around: the rasterisation takes the gate
    got = []
    th = threading.Thread(target=lambda: (g.__enter__(), got.append("launch"), g.__exit__()))
    th.start()
    t0 = time.time()
    while g.waiting() == 0 and time.time() - t0 < 2.0:
        time.sleep(0.001)
    assert g.waiting() == 1                                          # a launcher blocked at the gate IS the successor
    g.__exit__()
    th.join(2.0)
    assert got == ["launch"] and g.waiting() == 0
    # a launch holds the gate: a rasterisation does not wait for it -- that launch is its successor after all
    with g:
        res = []
        th = threading.Thread(target=lambda: res.append(g.enter_for_render()))
        th.start()
        th.join(2.0)
        assert res == [False]
