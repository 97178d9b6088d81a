"""Sound-sink adaptation: the queue and rate-error loop of CSoundOut
(interface/soundout.cpp:155-468, non-blocking mode).  CPU: the oracle restatement's behaviour rules; GPU: the
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
