"""End-to-end synthetic OCTA triples on one MI355X: graph (CSV rows) + 304x304 image + 1216x1216 label.

Mirrors what the reference produces with
    generate_vessel_graph.py --config_file docker/vessel_graph_gen_docker_config.yml     (graph + image)
    visualize_vessel_graphs.py --resolution 1216,1216,16 --binarize                      (label)
(docker/dockershell.sh:10-17), for a whole batch of seeded samples at once:
  * graphs: BatchSimulator (HIP simulator, bit-exact CSV rows),
  * image : arterial and venous edges rasterised separately at 304x304 and max-combined
            (generate_vessel_graph.py:79-86),
  * label : all edges, positions as read back from the CSV text, rasterised at 1216x1216 and
            Floyd-Steinberg binarised (visualize_vessel_graphs.py:95-99).
"""
import ctypes

import numpy as np

from . import _native, graph_io
from .vessel_graph_generation import greenhouse, tree2img


class SimGate:
    """The lock the generators of one device share so that ONE persistent kernel runs at a time, with what the device-side ordering needs:
    how many threads are waiting for it (is a successor launch coming?) and the events the NEXT launch has to be ordered behind.

    Why a launch must find the GPU free of every LDS-using workgroup (round 6, found with per-sample start times in bench.py's log): two
    simulator workgroups fill a CU's 160 KB of LDS exactly (80 KB each), and the hardware allocates LDS contiguously. A workgroup with ANY
    LDS that is resident while a launch is placed -- the edge-export kernel of the launch before (57 KB), a scan kernel of the rasteriser's
    planning step (2 KB), a render workgroup -- pushes the first simulator workgroup of that CU to an odd offset; when the small one leaves,
    neither hole fits the second simulator workgroup, and the CU runs ONE sample at a time for the whole launch: the launch "lasts two
    samples" (660 instead of 400 ms; rounds 4-5 called it the headline's slow mode and blamed the dispatcher). So every launch waits, on
    its own stream, for: the export (and planning) of the launch before, every rasterisation up to the one before last (the last one is
    held back by the gate kernel until this launch is resident), and a rasterisation that went ahead without a successor. Reentrant use is
    not supported."""

    def __init__(self, timeout_us=50000, settle_us=200):
        import threading
        self._lock = threading.Lock()
        self._meta = threading.Lock()
        self._waiting = 0
        self._holder = None
        self._barrier = None
        self._seq = 0                      # launches made through this gate
        self._drain = []                   # events of the launch before: edge export, rasteriser planning
        self._renders = []                 # (launch number, completion event) of rasterisations not yet known to be over
        self.timeout_us, self.settle_us = int(timeout_us), int(settle_us)

    def __enter__(self):
        """Taken for a LAUNCH: counted as waiting while blocked."""
        with self._meta:
            self._waiting += 1
        self._lock.acquire()
        with self._meta:
            self._waiting -= 1
            self._holder = "launch"
        return self

    def __exit__(self, *exc):
        self._holder = None
        self._lock.release()
        return False

    def enter_for_render(self):
        """Taken to enqueue a rasterisation that has no successor launch. Returns True with the gate held, or False when a LAUNCH holds it or
        is waiting for it (that launch is the successor after all); waits only for another rasterisation's enqueue (milliseconds)."""
        while True:
            if self._lock.acquire(blocking=False):
                with self._meta:
                    if self._waiting > 0:              # a launch arrived first in spirit: let it have the gate
                        self._lock.release()
                        return False
                    self._holder = "render"
                return True
            if self._holder != "render":
                return False
            import time
            time.sleep(0.0002)                         # another slot is enqueueing its rasterisation with the gate held

    def waiting(self):
        with self._meta:
            return self._waiting

    def locked(self):
        return self._lock.locked()

    def order_launch(self, stream):
        """Called with the gate held, before a launch is enqueued on `stream`: the stream waits for everything that must be off the GPU when
        the launch's workgroups are placed. Returns this launch's number."""
        with self._meta:
            self._seq += 1
            n = self._seq
            drain, self._drain = self._drain, []
            barrier, self._barrier = self._barrier, None
            old = [ev for k, ev in self._renders if k <= n - 2]
            self._renders = [(k, ev) for k, ev in self._renders if k > n - 2]
        for ev in drain + old + ([barrier] if barrier is not None else []):
            stream.wait_event(ev)
        return n

    def after_launch(self, *events):
        """Called with the gate held, after the launch: what the NEXT launch waits for (edge export, planning kernels)."""
        with self._meta:
            self._drain = [ev for ev in events if ev is not None]

    def note_render(self, n, event):
        with self._meta:
            self._renders.append((n, event))

    def set_barrier(self, event):
        with self._meta:
            self._barrier = event

    def take_barrier(self):
        with self._meta:
            ev, self._barrier = self._barrier, None
            return ev


class TripleGenerator:
    def __init__(self, config, batch, device_index=None, label_resolution=(1216, 1216), label_min_radius=0.0, image_mode="cli",
                 image_min_radius=0.0):
        """image_mode "cli": arterial and venous edges rendered separately and max-combined, as generate_vessel_graph.py:79-86 writes
        art_ven_img_gray.png; "loader": ONE render of the whole edge list read back from the CSV text with `image_min_radius`, as the
        training loader's LoadGraphAndFilterByRandomRadiusd does (data_transforms.py:376-386) -- what a network trained by train.py sees."""
        import torch
        self.config = config
        self.batch = int(batch)
        self.device = torch.device("cuda", torch.cuda.current_device() if device_index is None else device_index)
        self.sim = greenhouse.BatchSimulator(config, batch, self.device.index)
        self._ctx = _native.new_ctx(self.device.index)   # this slot's rasteriser scratch
        self._ctx_label = _native.new_ctx(self.device.index)   # a second one: image and label rasterisation are planned together (_plan)
        g, o = config["Greenhouse"], config.get("output", {})
        shape = np.array([g["SimulationSpace"][k] for k in ("no_voxel_x", "no_voxel_y", "no_voxel_z")])
        vol = [int(d) for d in shape * o.get("image_scale_factor", 304)]          # generate_vessel_graph.py:43
        self.proj_axis = int(o.get("proj_axis", 2))
        self.image_res = [v for i, v in enumerate(vol) if i != self.proj_axis]     # :81-82
        self.label_res = list(label_resolution)
        # the training configs render the label with min_radius[1] = 0.0033 (configs/config_ves_seg-S.yml:38)
        self.label_min_radius = float(label_min_radius)
        assert image_mode in ("cli", "loader")
        self.image_mode, self.image_min_radius = image_mode, float(image_min_radius)
        self.time_render = False

    def close(self):
        self.sim.close()
        for name in ("_ctx", "_ctx_label"):
            if getattr(self, name) is not None:
                _native.free_ctx(getattr(self, name))
                setattr(self, name, None)

    # plan both rasterisations (octa_rasterize_2d_plan: every host wait of the rasteriser) right after the simulator call, while the successor
    # launch is still being prepared on the host and the GPU is free; what is enqueued behind the gate kernel is then never cut in two by a
    # wait. Round 6 default: with the device-side ordering 1155 - 1162 against 1139 - 1151 samples/s (three alternating pairs on one box),
    # enqueueing a launch's rasterisation takes 0.3 ms instead of 178 (profiles/r06_order_runs.log).
    plan_ahead = True
    sim_gate = None      # optional SimGate shared by the generators of a device (bench.py --serial-sim): ONE persistent kernel at a time

    def prime(self, edges_per_sample=9600):
        """One-time set-up of this generator's rasteriser (scratch growth -- 2.7 GB of polygon sides, 1.4 GB of row lists: hipMalloc + an
        implicit device synchronisation each --, its output pool and the render kernels' code objects) on a synthetic edge list of a launch's
        size: nothing is simulated. Callers that time launches (bench.py) prime every generator before the clock starts."""
        import torch
        n = self.batch
        g = torch.Generator(device=self.device).manual_seed(1)
        e = torch.rand((n * edges_per_sample, 7), device=self.device, dtype=torch.float64, generator=g)
        e[:, 3:6] = e[:, 0:3] + (e[:, 3:6] - 0.5) * 0.03               # short segments
        e[:, 6] = 0.002 + 0.004 * e[:, 6]
        off = np.arange(n + 1, dtype=np.int64) * edges_per_sample
        fake = type("Primer", (), {"d_edges": e, "edges": None, "edge_off": off, "n_art": np.full(n, edges_per_sample // 2, np.int64)})()
        if self.sim_gate is not None and self._gated_streams is None:
            # the two streams of a gated generator, each used once: their hardware queues exist before the first timed launch
            self._gated_streams = (torch.cuda.Stream(device=self.device, priority=-1), torch.cuda.Stream(device=self.device, priority=0))
            for st in self._gated_streams:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=self.device)
                st.synchronize()
        if self._gated_streams is not None:
            with torch.cuda.stream(self._gated_streams[1]):
                self._render(fake, True)
                self._gated_streams[1].synchronize()
            return
        self._render(fake, True)

    def generate(self, seeds, want_label=True):
        """Returns dict(result=SimulationResult, image=uint8 CUDA [B,H,W], label=uint8 CUDA {0,255} [B,1216,1216])."""
        import time
        import torch
        t0 = t_req = t_rel = time.time()
        gate = self.sim_gate
        if gate is not None:
            # Several generators in flight, ONE persistent kernel at a time: the next launch starts when this one has left the GPU, and this
            # launch's rasterisation then shares the GPU with it (it fills the tail of that launch). The ORDER of the two matters -- a
            # rasterisation dispatched while the next launch's workgroups are being placed takes CUs it keeps, and the launch lasts two
            # samples -- and it is kept on the device (round 6; round 5 polled a launch counter and slept):
            #   * a successor is waiting at the gate: this rasterisation's stream waits, in a one-wave gate kernel, until the successor's
            #     launch (ticket + 1) is resident (csrc/order.hip); its render workgroups then get what finished samples leave;
            #   * nobody is waiting: the GPU is free, the rasterisation goes at once -- with the gate held while it is enqueued, and the NEXT
            #     launch, whoever makes it, is ordered behind this rasterisation's completion event on its own stream.
            # The launch and the rasterisation run on two streams this generator owns, of DIFFERENT queue priority: the runtime keeps a pool
            # of hardware queues per priority, so the spinning gate kernel can never sit in front of the launch it waits for in one hardware
            # queue (two streams of equal priority may share one: measured as a 5 s gate time-out in the full test suite), and the
            # dispatcher prefers the simulator's workgroups when both kinds are pending.
            if self._gated_streams is None:
                self._gated_streams = (torch.cuda.Stream(device=self.device, priority=-1), torch.cuda.Stream(device=self.device, priority=0))
            sim_stream, render_stream = self._gated_streams
            caller = torch.cuda.current_stream()
            entered = torch.cuda.Event()
            entered.record(caller)
            with gate:
                t0 = time.time()
                with torch.cuda.stream(sim_stream):
                    sim_stream.wait_event(entered)                          # what the caller enqueued before this call
                    n_launch = gate.order_launch(sim_stream)                # ... and what must be off the GPU when this launch is placed
                    t_r0 = time.time()
                    res = self.sim.run(seeds)
                    t_r1 = time.time()
                    ready = torch.cuda.Event()
                    ready.record(sim_stream)                                # behind the edge export of this run
                ticket = _native.lib().octa_sim_launch_count()
                # the rasteriser's planning step (per-edge records, scans, its one host wait) INSIDE the gate: its kernels use LDS, and the
                # successor's launch must not be placed beside them (SimGate); the successor's host-side set-up (~5 ms) has not begun yet
                planned = None
                with torch.cuda.stream(render_stream):
                    render_stream.wait_event(ready)
                    if res.d_edges is not None:
                        res.d_edges.record_stream(render_stream)
                    plans = self._plan(res, want_label) if (self.plan_ahead and not self.time_render) else None
                    if plans is not None:
                        planned = torch.cuda.Event()
                        planned.record(render_stream)
                gate.after_launch(ready, planned)
                successor = gate.waiting() > 0
            t_rel = time.time()
            host_stamps = {"gate_to_run_ms": 1e3 * (t_r0 - t0), "run_ms": 1e3 * (t_r1 - t_r0), "plan_in_gate_ms": 1e3 * (t_rel - t_r1)}
            with torch.cuda.stream(render_stream):
                out = self._ordered_render(gate, res, want_label, successor, ticket, plans)
                done = torch.cuda.Event()
                done.record(render_stream)
            gate.note_render(n_launch, done)
            out["done_event"] = done                                       # for callers that keep several results of one generator in flight
            caller.wait_event(done)                                        # the caller's stream sees finished outputs, as with one stream
            for k in ("image", "label", "label_grey"):
                if out.get(k) is not None:
                    out[k].record_stream(caller)
            if res.d_edges is not None:
                res.d_edges.record_stream(caller)
            out["wall"].update({"t_start": t0, "sim_run_s": out["wall"]["t1"] - t0, "t_request": t_req, "t_released": t_rel, "host": host_stamps})
            return out
        res = self.sim.run(seeds)
        t_rel = time.time()
        plans = self._plan(res, want_label) if (self.plan_ahead and not self.time_render) else None
        t1 = time.time()
        out = self._render(res, want_label, plans)
        out["wall"] = {"t_start": t0, "sim_run_s": t1 - t0, "render_enqueue_s": time.time() - t1,      # host-side stamps (bench.py's slot accounting)
                       "t_request": t_req, "t_released": t_rel}                                                               # when the call asked for the gate
        return out

    _gated_streams = None    # (simulator stream, rasteriser stream) of a gated generator, made on first use (see generate)

    def _ordered_render(self, gate, res, want_label, successor, ticket, plans):
        """The rasterisation of a gated launch on the CURRENT stream, ordered against the launches (see generate)."""
        import time
        import torch
        if not successor:
            if gate.enter_for_render():
                try:
                    t1 = time.time()
                    out = self._render(res, want_label, plans)
                    ev = torch.cuda.Event()
                    ev.record()
                    gate.set_barrier(ev)
                finally:
                    gate.__exit__()
            else:
                successor = True                   # a launch took the gate (or stands at it) since this one released it: ticket + 1
        gate_out = None
        if successor:
            gate_out = torch.zeros(3, dtype=torch.int32, device=self.device)        # what the gate kernel saw: 1 resident / 2 timed out, ticks waited, workgroups signed in
            _native.check(_native.lib().octa_order_wait_launch(_native.ctx(self.device.index), ticket + 1, gate.timeout_us, gate.settle_us,
                                                               ctypes.c_void_p(gate_out.data_ptr()), _native.current_stream_ptr()), "octa_order_wait_launch")
            t1 = time.time()
            out = self._render(res, want_label, plans)
        out["wall"] = {"t1": t1, "render_enqueue_s": time.time() - t1, "ordered_behind_successor": bool(successor), "gate": gate_out, "ticket": int(ticket)}
        return out

    def _plan(self, res, want_label):
        """The host-synchronising part of the rasterisation of `res` (tree2img.rasterize_edges_device_plan): CSV read-back emulation,
        per-edge records and side-offset scans of the image and of the label rasterisation, each on its own context."""
        import torch
        B = self.batch
        off, n_art = res.edge_off, res.n_art
        # the simulator exported the edge list on the device (round 3): no host BFS, no PCIe round trip of ~93 MB per 128 samples
        d_edges = res.d_edges if res.d_edges is not None else torch.from_numpy(res.edges).to(self.device, non_blocking=True)
        d_rb = None
        with _native.use_ctx(self._ctx):
            if self.image_mode == "cli":
                # 2B graphs: arterial_k, venous_k interleaved -> max of the pairs (np.maximum(art_mat, ven_mat))
                split = np.empty(2 * B + 1, np.int64)
                split[0::2] = off
                split[1::2] = off[:-1] + n_art
                p_img = tree2img.rasterize_edges_device_plan(d_edges, split, self.image_res, self.proj_axis)
            else:
                d_rb = graph_io.edges_as_read_back_device(d_edges)
                p_img = tree2img.rasterize_edges_device_plan(d_rb, off, self.image_res, self.proj_axis, min_radius=self.image_min_radius)
        p_lab = None
        if want_label:
            with _native.use_ctx(self._ctx_label):
                if d_rb is None:
                    d_rb = graph_io.edges_as_read_back_device(d_edges)
                p_lab = tree2img.rasterize_edges_device_plan(d_rb, off, self.label_res, 2, min_radius=self.label_min_radius)
        return dict(image=p_img, label=p_lab, keep=(d_edges, d_rb))

    def _render(self, res, want_label, plans=None):
        import torch
        B = self.batch
        if plans is not None:
            # planned while the GPU was free (generate): nothing below waits for the device
            image = tree2img.rasterize_edges_device_draw(plans["image"])
            if self.image_mode == "cli":
                pair = image.view(B, 2, image.shape[1], image.shape[2])
                with _native.use_ctx(self._ctx):
                    image = tree2img.maximum_u8_device(pair[:, 0].contiguous(), pair[:, 1].contiguous())
            out = dict(result=res, image=image)
            if want_label:
                grey = tree2img.rasterize_edges_device_draw(plans["label"])
                out["label_grey"] = grey
                with _native.use_ctx(self._ctx_label):
                    out["label"] = tree2img.binarize_label_device(grey)
            return out
        with _native.use_ctx(self._ctx):
            return self._render_timed(res, want_label)

    def _render_timed(self, res, want_label):
        """The rasterisation stage by stage in one go (plan and draw back to back), with HIP events around the stages when
        `time_render` is set -- what bench.py's rasteriser figures are read from."""
        import torch
        B = self.batch
        off, n_art = res.edge_off, res.n_art
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if self.time_render else None
        mark = (lambda i: ev[i].record()) if ev else (lambda i: None)       # on the current stream = the stream the kernels go to
        # the simulator exported the edge list on the device (round 3): no host BFS, no PCIe round trip of ~93 MB per 128 samples
        d_edges = res.d_edges if res.d_edges is not None else torch.from_numpy(res.edges).to(self.device, non_blocking=True)
        mark(0)
        d_rb = None
        if self.image_mode == "cli":
            # 2B graphs: arterial_k, venous_k interleaved -> max of the pairs (np.maximum(art_mat, ven_mat))
            split = np.empty(2 * B + 1, np.int64)
            split[0::2] = off
            split[1::2] = off[:-1] + n_art
            pair = tree2img.rasterize_edges_device(d_edges, split, self.image_res, self.proj_axis)
            pair = pair.view(B, 2, pair.shape[1], pair.shape[2])
            image = tree2img.maximum_u8_device(pair[:, 0].contiguous(), pair[:, 1].contiguous())
        else:
            d_rb = graph_io.edges_as_read_back_device(d_edges)
            image = tree2img.rasterize_edges_device(d_rb, off, self.image_res, self.proj_axis, min_radius=self.image_min_radius)
        mark(1)
        out = dict(result=res, image=image)
        if want_label:
            if d_rb is None:
                d_rb = graph_io.edges_as_read_back_device(d_edges)
            mark(2)
            grey = tree2img.rasterize_edges_device(d_rb, off, self.label_res, 2, min_radius=self.label_min_radius)
            mark(3)
            out["label_grey"] = grey
            out["label"] = tree2img.binarize_label_device(grey)
            mark(4)
        if ev:
            # read with render_ms() after the stream has been synchronised
            out["render_events"] = ev if want_label else ev[:2]
        return out

    @staticmethod
    def render_ms(out):
        """GPU milliseconds of the render stages of a finished generate() (time_render=True): image rasterisation (2B graphs at
        the image resolution + max), CSV read-back emulation, label rasterisation, dither."""
        ev = out.get("render_events")
        if not ev:
            return None
        names = ["image_raster_ms", "read_back_ms", "label_raster_ms", "dither_ms"]
        return {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names) if i + 1 < len(ev)}
