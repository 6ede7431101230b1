"""CPU restatement of the waveform augmentation chain (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/utils/reader.py (cited per function).  Pinned by
tests/test_oracle_golden.py against outputs of the reference's own functions run
in the build container (tests/golden/make_golden.py), except ``resample`` which
is PARITY UNPINNED (librosa/soxr are not installed anywhere in this pipeline).

Arithmetic notes that matter for parity:
  * change_speed interpolates in float64 and rounds once to float32
    (np.interp promotes; utils/reader.py:291);
  * volume / noise gains are Python floats applied with an in-place ``*=`` on a
    float32 array, i.e. the scale is rounded to float32 first (NumPy 2 weak
    scalars) and the product is a single float32 multiply;
  * rms_db is np.mean of the float32 squares (pairwise float32 summation) and
    returns a float32 scalar under NumPy 2.
"""
from __future__ import annotations

import numpy as np


def change_speed(sample: np.ndarray, speed_rate: float) -> np.ndarray:
    """utils/reader.py:281-292.  new_len = int(old_len / rate); positions
    linspace(0, old_len, new_len) -- the stop is old_len (one past the last
    sample), so the tail clamps to the last sample."""
    if speed_rate == 1.0:
        return sample
    if speed_rate <= 0:
        raise ValueError("speed rate must be > 0")
    old_len = sample.shape[0]
    new_len = int(old_len / speed_rate)
    pos = np.linspace(0, old_len, new_len)
    return np.interp(pos, np.arange(old_len), sample).astype(np.float32)


def shift_samples(shift_ms: int, sample_rate: int) -> int:
    """utils/reader.py:300 -- int() truncates toward zero."""
    return int(shift_ms * sample_rate / 1000)


def shift(sample: np.ndarray, sample_rate: int, shift_ms: int) -> np.ndarray:
    """utils/reader.py:295-307 (in place): k>0 advances the signal and zeroes
    the last k samples, k<0 delays it and zeroes the first |k|."""
    if abs(shift_ms) / 1000.0 > sample.shape[0] / sample_rate:
        raise ValueError("|shift_ms| exceeds the clip duration")
    k = shift_samples(shift_ms, sample_rate)
    if k > 0:
        sample[:-k] = sample[k:]
        sample[-k:] = 0
    elif k < 0:
        sample[-k:] = sample[:k]
        sample[:-k] = 0
    return sample


def volume(sample: np.ndarray, gain: int) -> np.ndarray:
    """utils/reader.py:310-313 (in place)."""
    sample *= 10. ** (gain / 20.)
    return sample


def rms_db(sample: np.ndarray):
    """utils/reader.py:350-353."""
    return 10 * np.log10(np.mean(sample ** 2))


def noise_gains(sample: np.ndarray, noise: np.ndarray, snr_db: int, max_gain_db: float = 300.0):
    """The two scalar gains of add_noise (utils/reader.py:325-331) as the Python
    floats that multiply the arrays.  Does not modify its inputs."""
    gain = min(max_gain_db, -20 - rms_db(sample))
    g1 = 10. ** (gain / 20.)
    scaled = sample * g1
    noise_gain_db = min(rms_db(scaled) - rms_db(noise) - snr_db, max_gain_db)
    g2 = 10. ** (noise_gain_db / 20.)
    return g1, g2


def add_noise(sample: np.ndarray, noise: np.ndarray, snr_db: int, randint,
              max_gain_db: float = 300.0) -> np.ndarray:
    """utils/reader.py:322-340 with the noise clip already decoded.  ``randint``
    is random.randint (drawn only when the noise is longer than the sample)."""
    noise = noise.copy()
    gain = min(max_gain_db, -20 - rms_db(sample))
    sample *= 10. ** (gain / 20.)
    s_db, n_db = rms_db(sample), rms_db(noise)
    noise_gain_db = min(s_db - n_db - snr_db, max_gain_db)
    noise *= 10. ** (noise_gain_db / 20.)
    ls, ln = sample.shape[0], noise.shape[0]
    if ln < ls:
        noise = np.pad(noise, (0, ls - ln), "wrap")
    elif ln > ls:
        start = randint(0, ln - ls)
        noise = noise[start:ls + start]
    sample += noise
    return sample


def add_reverb(sample: np.ndarray, rir: np.ndarray) -> np.ndarray:
    """utils/reader.py:342-348 with the impulse response already decoded (and at the
    utterance's rate): scipy.signal.fftconvolve(mode='same') -- the reference's own
    third-party call, installed here (scipy 1.15.3) -- then the peak normalisation.
    Pinned against scipy by construction; the three lines around it are restated."""
    from scipy.signal import fftconvolve
    out = fftconvolve(sample, rir, mode="same")
    max_rate = np.max(np.abs(sample)) / np.max(np.abs(out))
    return out * max_rate


def augment(sample: np.ndarray, sample_rate: int, configs: list, rnd, noise_bank: list,
            speed_rates=None, np_rnd=None, noise_rates=None, rir_bank=None, rir_rates=None):
    """utils/reader.py:246-278: config entries in file order; a ``random()`` gate
    is drawn for every entry whose type is handled (even at prob 0), then the
    type-specific draws.  ``rnd`` is a ``random.Random`` (or the module);
    ``noise_bank`` is the list of decoded noise clips standing in for
    ``self.noise_files`` (decode itself is outside the pinned path).
    The resample entry (:263-267) draws its rate with ``np_rnd.choice`` (the
    reference uses numpy's global RNG) and resamples with oracle/resample.py (PARITY
    UNPINNED, see there); ``noise_rates`` are the native rates of the noise clips, which
    ``librosa.load(noise_path, sr=sample_rate)`` (:323) brings to the utterance's
    current rate.  ``rir_bank`` / ``rir_rates``: the decoded impulse responses standing
    in for ``self.rir_files`` (:274-277)."""
    from . import resample as _rs
    trace = []
    for cfg in configs:
        kind, prob, prm = cfg["type"], cfg.get("prob"), cfg.get("params", {})
        if kind == "speed" and rnd.random() < prob:
            if speed_rates is None:
                speed_rates = np.linspace(prm["min_speed_rate"], prm["max_speed_rate"],
                                          prm["num_rates"], endpoint=True)
            rate = rnd.choice(speed_rates)
            sample = change_speed(sample, rate)
            trace.append(("speed", float(rate)))
        if kind == "shift" and rnd.random() < prob:
            ms = rnd.randint(prm["min_shift_ms"], prm["max_shift_ms"])
            sample = shift(sample, sample_rate, ms)
            trace.append(("shift", ms))
        if kind == "volume" and rnd.random() < prob:
            g = rnd.randint(prm["min_gain_dBFS"], prm["max_gain_dBFS"])
            sample = volume(sample, g)
            trace.append(("volume", g))
        if kind == "resample" and rnd.random() < prob:
            new_rate = int((np_rnd or np.random).choice(prm["new_sample_rates"]))
            sample = _rs.resample(sample, sample_rate, new_rate)
            sample_rate = new_rate
            trace.append(("resample", new_rate))
        if kind == "noise" and rnd.random() < prob:
            if noise_bank:
                j = rnd.choice(range(len(noise_bank)))
                snr = rnd.randint(prm["min_snr_dB"], prm["max_snr_dB"])
                noise = noise_bank[j]
                if noise_rates is not None and int(noise_rates[j]) != int(sample_rate):
                    noise = _rs.resample(noise, int(noise_rates[j]), int(sample_rate))
                sample = add_noise(sample, noise, snr, rnd.randint)
                trace.append(("noise", j, snr))
        if kind == "reverb" and rnd.random() < prob:
            if rir_bank:
                j = rnd.choice(range(len(rir_bank)))
                rir = rir_bank[j]
                if rir_rates is not None and int(rir_rates[j]) != int(sample_rate):
                    rir = _rs.resample(rir, int(rir_rates[j]), int(sample_rate))
                sample = add_reverb(sample, rir)
                trace.append(("reverb", j))
    return sample, sample_rate, trace
