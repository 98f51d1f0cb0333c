"""Trainer and generator taking turns on one GPU (train_synthetic.py, round 5).

The persistent simulator kernel holds every CU's LDS and registers: a training step that shares the GPU with a generator launch takes
112 instead of 17 ms while the launch stretches from 410 to 729 ms. So the PRODUCER asks for the GPU before a launch,
the CONSUMER answers at its next step boundary -- its queued kernels drained -- and waits until the producer hands the GPU back.

    producer thread                         consumer (training) thread
    ---------------                         --------------------------
    turns.ask(stop)          ---------->    turns.step_aside_if_asked(drain, alive)     # at every step boundary AND while waiting for data
    ... launch, wait ...                    (blocked)
    turns.hand_back()        ---------->    continues

The consumer must call step_aside_if_asked() also inside its wait for the producer's next batch: the producer may be asking for the GPU
to make the very batch the consumer is waiting for.
"""
import threading


class GpuTurns:
    def __init__(self, enabled=True, poll_s=0.05, aligned=False):
        """aligned: several ranks train in lock step (one gradient all-reduce per step), so a turn given at one rank's own moment stalls
        every rank; with aligned=True turns are only given where the caller says `at_boundary` -- a step every rank reaches together
        (the step that needs the next generator batch) -- so that the ranks' stalls coincide instead of adding up."""
        self.enabled = bool(enabled)
        self.aligned = bool(aligned)
        self.poll_s = float(poll_s)
        self._want, self._aside, self._back = threading.Event(), threading.Event(), threading.Event()

    # ---- producer side
    def ask(self, stop=None):
        """Request the GPU and wait until the consumer has stepped aside (or `stop` is set). Returns True when the turn was granted."""
        if not self.enabled:
            return True
        self._want.set()
        while not self._aside.wait(self.poll_s):
            if stop is not None and stop.is_set():
                return False
        return True

    def hand_back(self):
        """The producer's launch is over (call it in a `finally`): the consumer may continue."""
        self._want.clear()
        self._aside.clear()
        self._back.set()

    # ---- consumer side
    def step_aside_if_asked(self, drain, producer_alive=lambda: True, at_boundary=True):
        """If the producer has asked: `drain()` (wait for the consumer's own queued GPU work), tell the producer, and wait for hand_back()
        -- or for the producer to die. Returns True when a turn was given away. An aligned instance only gives turns `at_boundary`."""
        if not (self.enabled and self._want.is_set()):
            return False
        if self.aligned and not at_boundary:
            return False
        drain()
        self._back.clear()
        self._aside.set()
        while not self._back.wait(self.poll_s):
            if not producer_alive():
                break
        return True
