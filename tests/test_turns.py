"""CPU: the turn-taking handshake of the on-the-fly loops (utils/turns.py; train_synthetic.py runs it between the generator thread and the trainer)."""
import queue
import threading
import time

from octa_autosegmentation_amd.utils.turns import GpuTurns


def test_consumer_steps_aside_between_steps_and_while_it_waits_for_data():
    """The producer asks before every batch but the first; the consumer calls step_aside_if_asked() at its step boundaries and inside its wait
    for data -- the producer may be asking for the GPU to make the very batch the consumer waits for (a consumer that only looked at its step
    boundaries would deadlock here). While the producer has the turn the consumer runs no step."""
    turns = GpuTurns(poll_s=0.005)
    q = queue.Queue(maxsize=1)
    stop = threading.Event()
    log = []
    lock = threading.Lock()

    def note(what):
        with lock:
            log.append(what)

    def produce():
        for i in range(4):
            if i > 0:
                assert turns.ask(stop)
            try:
                note(("launch_begin", i))
                time.sleep(0.02)
                note(("launch_end", i))
            finally:
                turns.hand_back()
            q.put(i)

    th = threading.Thread(target=produce)
    th.start()
    drained = []
    got = []
    t_end = time.time() + 5.0
    while len(got) < 4 and time.time() < t_end:
        turns.step_aside_if_asked(lambda: drained.append(len(got)), th.is_alive)
        try:
            got.append(q.get(timeout=0.005))
        except queue.Empty:
            continue
        for _ in range(3):                                             # three "training steps" per batch
            turns.step_aside_if_asked(lambda: drained.append(len(got)), th.is_alive)
            note(("step", got[-1]))
            time.sleep(0.002)
    th.join(2.0)
    assert got == [0, 1, 2, 3] and not th.is_alive()
    assert len(drained) == 3                                           # one turn given away per asked launch
    # no step between a launch's begin and end, except for launch 0 (nobody asked)
    inside = False
    for what, i in log:
        if what == "launch_begin" and i > 0:
            inside = True
        elif what == "launch_end":
            inside = False
        elif what == "step":
            assert not inside, log


def test_disabled_turns_never_block():
    turns = GpuTurns(enabled=False)
    assert turns.ask() is True
    assert turns.step_aside_if_asked(lambda: (_ for _ in ()).throw(AssertionError("no drain when disabled"))) is False
    turns.hand_back()


def test_a_stopped_run_releases_the_producer_and_a_dead_producer_releases_the_consumer():
    turns = GpuTurns(poll_s=0.005)
    stop = threading.Event()
    res = []
    th = threading.Thread(target=lambda: res.append(turns.ask(stop)))
    th.start()
    time.sleep(0.02)
    stop.set()                                                         # the consumer never answers: the run is over
    th.join(1.0)
    assert res == [False] and not th.is_alive()
    turns.hand_back()
    # the consumer gives its turn away and the producer dies without handing the GPU back
    turns2 = GpuTurns(poll_s=0.005)
    dead = threading.Thread(target=lambda: None)
    dead.start(); dead.join()
    turns2._want.set()
    t0 = time.time()
    assert turns2.step_aside_if_asked(lambda: None, dead.is_alive) is True
    assert time.time() - t0 < 1.0


def test_aligned_turns_are_only_given_at_the_boundary_every_rank_shares():
    """Several ranks (train_synthetic.py with WORLD_SIZE > 1): a turn given mid-batch would stall every rank at the step's all-reduce, each
    rank at its own moment; an aligned instance keeps training through a mid-batch request and gives the turn at the batch boundary."""
    turns = GpuTurns(poll_s=0.005, aligned=True)
    granted = []
    th = threading.Thread(target=lambda: granted.append(turns.ask()))
    th.start()
    time.sleep(0.02)                                                   # the producer is asking now
    drains = []
    for step in range(5):                                              # mid-batch steps: asked, but not the agreed step
        assert turns.step_aside_if_asked(lambda: drains.append(step), th.is_alive, at_boundary=False) is False
    assert not drains and not granted
    releaser = threading.Timer(0.03, turns.hand_back)
    releaser.start()
    assert turns.step_aside_if_asked(lambda: drains.append("boundary"), lambda: True, at_boundary=True) is True
    th.join(1.0)
    releaser.join()
    assert drains == ["boundary"] and granted == [True]
