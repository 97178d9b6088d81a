"""Seeded synthetic IQ modelled on the reference test bench generator
(gui/testbench.cpp:396-445): full scale 32767, carriers in dBFS, additive Gaussian noise.
Own PRNG (numpy PCG64), seed = 0xC0DE0000 + channel."""
import numpy as np

FULL_SCALE = 32767.0


def channel_rng(channel):
    return np.random.Generator(np.random.PCG64(0xC0DE0000 + int(channel)))


def tones_plus_noise(channel, n, fs, tones_hz, tone_dbfs=-20.0, noise_dbfs=-70.0, start=0):
    rng = channel_rng(channel)
    t = np.arange(start, start + n, dtype=np.float64)
    x = np.zeros(n, dtype=np.complex128)
    amp = FULL_SCALE * 10 ** (tone_dbfs / 20.0)
    for f in tones_hz:
        x += amp * np.exp(2j * np.pi * f * t / fs)
    sig = FULL_SCALE * 10 ** (noise_dbfs / 20.0)
    x += sig * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x


def fm_carrier(n, fs, fc, fmod=1000.0, dev=3000.0, dbfs=-20.0, noise_dbfs=-70.0, channel=0):
    rng = channel_rng(channel)
    t = np.arange(n, dtype=np.float64) / fs
    ph = 2 * np.pi * fc * t + (dev / fmod) * np.sin(2 * np.pi * fmod * t)
    x = FULL_SCALE * 10 ** (dbfs / 20.0) * np.exp(1j * ph)
    sig = FULL_SCALE * 10 ** (noise_dbfs / 20.0)
    return x + sig * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


def am_carrier(n, fs, fc, fmod=1000.0, depth=0.5, dbfs=-20.0, noise_dbfs=-70.0, channel=0):
    rng = channel_rng(channel)
    t = np.arange(n, dtype=np.float64) / fs
    env = 1.0 + depth * np.sin(2 * np.pi * fmod * t)
    x = FULL_SCALE * 10 ** (dbfs / 20.0) * env * np.exp(2j * np.pi * fc * t)
    sig = FULL_SCALE * 10 ** (noise_dbfs / 20.0)
    return x + sig * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
