"""Sound-sink adaptation: the queue and rate-error loop of CSoundOut
(interface/soundout.cpp:155-468, both modes).  CPU: the oracle restatement's behaviour rules; GPU: the
product (csdr_soundsink_*, device resampler) against the oracle under the same simulated clocks -- resampled
counts exact, queue levels and rate corrections identical, samples within 1 LSB."""
import numpy as np
import pytest


def drive(sink, seconds, fs_audio, producer_ppm, block=1024, tick_hz=100, stereo=False, seed=3, volume=80):
    """A producer delivering `block` audio samples at fs_audio * (1 + ppm) and a sound card popping 48000 / tick_hz
    samples per tick; returns per-tick (level, correction) and everything popped."""
    rng = np.random.default_rng(seed)
    sink.ChangeUserDataRate(fs_audio)
    sink.SetVolume(volume)
    t_in = 0.0
    n_in = 0
    popped, trace, counts = [], [], []
    phase = 0.0
    for tick in range(int(seconds * tick_hz)):
        t_in += fs_audio * (1.0 + producer_ppm * 1e-6) / tick_hz
        while n_in + block <= t_in:
            k = np.arange(block) + n_in
            tone = 9000.0 * np.sin(2 * np.pi * 1000.0 * k / fs_audio)
            x = tone + 1j * 9000.0 * np.cos(2 * np.pi * 700.0 * k / fs_audio) if stereo else tone
            counts.append(sink.PutOutQueue(x))
            n_in += block
        popped.append(sink.GetOutQueue(48000 // tick_hz))
        trace.append((sink.level(), sink.rate_correction(), sink.ave_level(), sink.ppm_error()))
    return np.array(counts), np.concatenate(popped), np.array(trace)


def test_oracle_soundsink_rules(oracle):
    s = oracle.CSoundOut()
    s.ChangeUserDataRate(62500.0)
    assert s.ave_level() == 8192.0 and s.level() == 0
    out = s.GetOutQueue(480)
    assert not out.any()                                         # start-up: silence until half full
    x = 5000.0 * np.ones(1024)
    n = 0
    while s.level() <= 8192:
        n += s.PutOutQueue(x)
        assert not s.GetOutQueue(10).any() or s.level() > 8192 - 10
    assert s.level() > 8192 - 10
    # overflow: a quarter of the queue is dropped and the average snaps to the level
    while True:
        before = s.level()
        k = s.PutOutQueue(x)
        if s.level() < before + k:
            assert before + k - s.level() >= 4096 - k and s.ave_level() == pytest.approx(s.level(), rel=2e-3)
            break
    # underflow: the tail backs up a quarter
    lvl = s.level()
    s.GetOutQueue(lvl + 100)
    assert s.level() == 4096 - 100 + 0 or s.level() > 0
    # volume law: 0 mutes, 99 is unity (SetVolume :180-189)
    s2 = oracle.CSoundOut(); s2.ChangeUserDataRate(48000.0); s2.SetVolume(0)
    for _ in range(20): s2.PutOutQueue(x)
    assert not s2.GetOutQueue(9000).any()


def test_oracle_rate_loop_follows_the_first_order_law(oracle):
    """Producer 1000 ppm fast: the queue fills at 48 samples/s, the P-controller (CalcError :456-468, once per
    second of consumed samples) answers with correction = 2.38e-7 * (average level - 8192): a first-order loop
    with time constant 1 / (P_GAIN * 48000) = 87.5 s that settles where the correction equals the clock error
    (level 8192 + 4202).  40 simulated seconds: correction = 1e-3 * (1 - exp(-40 / 87.5)) within 10 %."""
    s = oracle.CSoundOut()
    counts, popped, trace = drive(s, 40.0, 62500.0, 1000.0)
    corr = trace[:, 1]
    assert corr[:500].max() == 0.0                               # first update delayed by 5 s of consumed samples
    assert np.all(np.diff(corr[600:]) >= 0.0)                    # rises monotonically towards the clock error
    assert corr[-1] == pytest.approx(1e-3 * (1.0 - np.exp(-40.0 / 87.5)), rel=0.10)
    assert trace[-1, 3] == int(corr[-1] * 1e6)
    assert abs(trace[-1, 0] - (8192 + corr[-1] / 2.38e-7)) < 400  # the level that correction stands for


@pytest.mark.gpu
@pytest.mark.parametrize("stereo", [False, True], ids=["mono", "stereo"])
def test_soundsink_matches_oracle(oracle, stereo):
    import cutesdr_amd as ca
    g, r = ca.CSoundOut(stereo), oracle.CSoundOut(stereo)
    cg, pg, tg = drive(g, 9.0, 62500.0, 800.0, stereo=stereo)
    cr, pr, tr = drive(r, 9.0, 62500.0, 800.0, stereo=stereo)
    assert np.array_equal(cg, cr)                                # resampled counts of every put call
    assert np.array_equal(tg, tr)                                # level, correction, average, ppm after every tick
    assert tr[-1, 1] != 0.0                                      # the loop has started correcting
    assert np.abs(pg.astype(np.int32) - pr.astype(np.int32)).max() <= 1
    assert np.abs(pr).max() > 1000


def drive_blocking(sink, calls, fs_audio, block=1024, stereo=False):
    """Blocking mode under a schedule on which a put never has to wait: one put, then the sound card takes what the
    queue holds above half (so start-up ends and the queue neither fills nor runs empty)."""
    sink.ChangeUserDataRate(fs_audio)
    sink.SetVolume(90)
    sink.SetBlocking(True)
    counts, popped, trace = [], [], []
    for c in range(calls):
        k = np.arange(block) + c * block
        tone = 9000.0 * np.sin(2 * np.pi * 1000.0 * k / fs_audio)
        x = tone + 1j * 9000.0 * np.cos(2 * np.pi * 700.0 * k / fs_audio) if stereo else tone
        counts.append(sink.PutOutQueue(x))
        take = sink.level() - 8192 - 100
        popped.append(sink.GetOutQueue(max(take, 64) if sink.level() > 8192 else 64))
        trace.append((sink.level(), sink.rate_correction(), sink.ave_level()))
    return np.array(counts), np.concatenate(popped), np.array(trace)


def test_oracle_blocking_mode_rules(oracle):
    """Blocking mode (interface/soundout.cpp:209-220, 354-358): nothing is dropped, nothing is averaged, the rate
    controller never runs; the single-threaded restatement reports the put that would have had to sleep."""
    s = oracle.CSoundOut()
    counts, popped, trace = drive_blocking(s, 60, 62500.0)
    assert (counts > 0).all()
    # no correction ever; the average is set once, when the start-up ends (:316-333), and never filtered afterwards
    assert trace[:, 1].max() == 0.0 and len(np.unique(trace[:, 2])) == 2 and trace[0, 2] == 8192.0
    body = popped[np.flatnonzero(popped)[0]:]
    r = oracle.CFractResampler(); r.Init(8192)
    k = np.arange(60 * 1024)
    ref = np.concatenate([r.Resample(9000.0 * np.sin(2 * np.pi * 1000.0 * k[i:i + 1024] / 62500.0), 62500.0 / 48000.0,
                                     gain=10 ** ((90 - 99.0) / 39.2)) for i in range(0, len(k), 1024)])
    i0 = int(np.flatnonzero(ref)[0])
    assert np.array_equal(body[:2000], ref[i0:i0 + 2000])                        # the resampled stream, in order, no gaps
    full = oracle.CSoundOut(); full.ChangeUserDataRate(48000.0); full.SetBlocking(True)
    x = 5000.0 * np.ones(4096)
    rc = [full.PutOutQueue(x) for _ in range(5)]
    assert rc[:3] == [rc[0]] * 3 and rc[0] > 4000 and rc[-1] < 0                 # the queue is full: the reference sleeps here
    assert full.level() == 16383


@pytest.mark.gpu
@pytest.mark.parametrize("stereo", [False, True], ids=["mono", "stereo"])
def test_soundsink_blocking_mode_matches_oracle(oracle, stereo):
    import cutesdr_amd as ca
    g, r = ca.CSoundOut(stereo), oracle.CSoundOut(stereo)
    cg, pg, tg = drive_blocking(g, 60, 62500.0, stereo=stereo)
    cr, pr, tr = drive_blocking(r, 60, 62500.0, stereo=stereo)
    assert np.array_equal(cg, cr) and np.array_equal(tg, tr)
    assert np.abs(pg.astype(np.int32) - pr.astype(np.int32)).max() <= 1 and np.abs(pr).max() > 1000


@pytest.mark.gpu
def test_soundsink_blocking_put_waits_for_the_sound_card():
    """Two threads as in the reference (IQ thread: PutOutQueue, audio thread: GetOutQueue): the producer offers 40 000
    resampled samples as fast as it can to a 16 384-entry queue; the consumer pops 480 at a time with a pause.  The
    producer must have waited (it cannot finish before the consumer has made room), and what the consumer got, after
    the start-up silence, is the resampled stream in order: nothing dropped, nothing repeated."""
    import threading, time
    import cutesdr_amd as ca
    fs, block, calls = 48000.0, 1000, 40
    s = ca.CSoundOut(False)
    s.ChangeUserDataRate(fs); s.SetVolume(99); s.SetBlocking(True)
    ref = ca.CFractResampler(); ref.Init(8192)
    x = [6000.0 * np.sin(2 * np.pi * 0.013 * (np.arange(block) + c * block)) + 2000.0 for c in range(calls)]
    want = np.concatenate([ref.Resample(xc, 1.0, gain=1.0) for xc in x])
    t_done = {}

    def producer():
        for xc in x:
            assert s.PutOutQueue(xc) > 0
        t_done["put"] = time.perf_counter()
    got, t0 = [], time.perf_counter()
    th = threading.Thread(target=producer)
    th.start()
    time.sleep(0.3)                                             # the producer runs into the full queue and waits
    assert th.is_alive() and s.level() == 16383
    while sum(len(g) for g in got) < len(want) + 9000:
        got.append(s.GetOutQueue(480))
        if th.is_alive():
            time.sleep(0.0005)
        else:
            break
    th.join(timeout=30)
    assert not th.is_alive()
    t_first_room = t0 + 0.3
    assert t_done["put"] > t_first_room                          # it finished only after the consumer had made room
    while s.level() > 0:
        got.append(s.GetOutQueue(min(480, s.level())))
    got = np.concatenate(got)
    body = got[np.flatnonzero(got)[0]:]
    i0 = int(np.flatnonzero(want)[0])
    n = min(len(body), len(want) - i0)
    assert n > 30000 and np.array_equal(body[:n], want[i0:i0 + n])
    assert s.rate_correction() == 0.0
