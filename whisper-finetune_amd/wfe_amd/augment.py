"""Host side of the augmentation chain: draws the random decisions exactly as
``CustomDataset.augment`` does (/root/reference/utils/reader.py:246-278 -- same
``random`` calls in the same order, so a seeded run replays the reference's
choices) and turns them into the per-utterance op list the HIP kernels apply
while staging the waveform (include/wfe.h ``wfe_aug_op``).  No sample is touched
on the host.

Draw order restated from the reference:
  * config entries in file order; an entry whose ``type`` is one of speed / shift /
    volume / resample / noise / reverb draws ``random.random()`` once (also at
    prob 0.0); any other type (e.g. specaug) draws nothing here;
  * speed:  ``random.choice(np.linspace(min, max, num))``; rate 1.0 is a no-op; the rates are those of the
            FIRST speed entry that fires (``self.speed_rates`` is built once, reader.py:249-252); every
            further speed change of one utterance starts a new STAGE (below);
  * shift:  ``random.randint(min_ms, max_ms)``; samples = int(ms * sr / 1000);
  * volume: ``random.randint(min_dB, max_dB)``; multiplier float32(10 ** (g / 20));
  * noise:  only when noise clips exist: ``random.choice(files)``,
            ``random.randint(min_snr, max_snr)`` and, when the clip is longer than
            the sample, ``random.randint(0, len_noise - len_sample)`` (reader.py:337);
  * resample: ``np.random.choice(new_sample_rates)`` (numpy's global RNG, reader.py:265); the
            sample rate changes for every later entry (shift samples, noise clips loaded at the
            new rate).  A host-only marker op (HOST_RESAMPLE, rate) splits the chain into STAGES
            (``split_stages``): each stage is a fused chain at one rate, stages are joined by the
            GPU resampler (include/wfe.h wfe_resample), the last one ends at the target rate;
  * reverb: only when impulse responses exist: ``random.choice(rir_files)``; a host-only marker op
            (HOST_REVERB, index) ends a stage like a resample does: the stage's waveform goes
            through the GPU convolution (include/wfe.h wfe_reverb) at the current rate.
"""
from __future__ import annotations

import random as _random
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import AUG_NOISE, AUG_SHIFT, AUG_SPEED, AUG_VOLUME

Op = Tuple[int, int, int, int, float]      # (kind, i0, i1, i2, f0) -- see wfe_aug_op
HOST_RESAMPLE = 100                        # (HOST_RESAMPLE, new_rate, 0, 0, 0.0): never sent to the device
HOST_REVERB = 101                          # (HOST_REVERB, rir index, 0, 0, 0.0): likewise


def resampled_length(n: int, orig_sr: int, target_sr: int) -> int:
    """librosa.resample / librosa.load: int(ceil(n * (float(target_sr) / orig_sr)))."""
    return int(n) if int(orig_sr) == int(target_sr) else int(np.ceil(n * (float(target_sr) / orig_sr)))


def split_stages(ops: Sequence[Op], length: int, sample_rate: int, target_rate: int):
    """-> [{"sr", "ops", "len_in", "len_out", "post"}]: ``ops`` (device ops only) run on
    ``len_in`` samples at ``sr`` and leave ``len_out``; ``post`` is what joins the stage
    to the next one: ("resample", new_sr), ("reverb", rir index), ("pass", 0) in front of a
    second speed change, or None after the last stage, which is always at ``target_rate``
    (reader.py:157-158)."""
    stages, cur, sr, n = [], [], int(sample_rate), int(length)
    n_in = n
    for op in list(ops) + [(HOST_RESAMPLE, int(target_rate), 0, 0, 0.0)]:
        if op[0] == HOST_RESAMPLE:
            new_sr = int(op[1])
            if new_sr != sr:
                stages.append({"sr": sr, "ops": cur, "len_in": n_in, "len_out": n, "post": ("resample", new_sr)})
                n = resampled_length(n, sr, new_sr)
                cur, sr, n_in = [], new_sr, n
            continue
        if op[0] == HOST_REVERB:
            stages.append({"sr": sr, "ops": cur, "len_in": n_in, "len_out": n, "post": ("reverb", int(op[1]))})
            cur, n_in = [], n
            continue
        if op[0] == AUG_SPEED:
            if any(o[0] == AUG_SPEED for o in cur):
                # a second speed change (a config with several 'speed' entries, reader.py:248-254): the device
                # chain interpolates once, so the chain so far is materialised and a new stage starts here
                stages.append({"sr": sr, "ops": cur, "len_in": n_in, "len_out": n, "post": ("pass", 0)})
                cur, n_in = [], n
            n = int(op[1])
        cur.append(op)
    stages.append({"sr": sr, "ops": cur, "len_in": n_in, "len_out": n, "post": None})
    return stages


def is_host_op(op: Op) -> bool:
    return op[0] in (HOST_RESAMPLE, HOST_REVERB)


class AugmentPlanner:
    def __init__(self, configs: Sequence[dict], noise_lengths: Optional[Sequence[int]] = None,
                 n_rir_files: int = 0, noise_rates: Optional[Sequence[int]] = None):
        """``noise_lengths`` / ``noise_rates``: length and native sampling rate of every
        noise clip (rates default to "the utterance's own rate": no resampling of noise)."""
        self.configs = list(configs)
        self.noise_lengths = list(noise_lengths) if noise_lengths else []
        self.noise_rates = list(noise_rates) if noise_rates else None
        self.n_rir_files = int(n_rir_files)
        self.speed_rates = None            # lazily built, like CustomDataset.speed_rates (reader.py:249-252)

    def draw(self, length: int, sample_rate: int, rnd=_random, np_rnd=None) -> Tuple[List[Op], int]:
        """Decisions for one utterance of ``length`` samples -> (ops, final length at the
        final rate).  ``np_rnd``: the numpy generator the resample entry draws from
        (default: numpy's global one, as the reference)."""
        ops: List[Op] = []
        n = int(length)
        sample_rate = int(sample_rate)
        for cfg in self.configs:
            kind, prob, prm = cfg["type"], cfg.get("prob"), cfg.get("params", {})
            if kind == "speed" and rnd.random() < prob:
                if self.speed_rates is None:
                    self.speed_rates = np.linspace(prm["min_speed_rate"], prm["max_speed_rate"],
                                                   prm["num_rates"], endpoint=True)
                rate = rnd.choice(self.speed_rates)
                if rate != 1.0:
                    if rate <= 0:
                        raise ValueError("speed rate must be > 0")
                    new_len = int(n / rate)
                    ops.append((AUG_SPEED, new_len, 0, 0, 0.0))
                    n = new_len
            if kind == "shift" and rnd.random() < prob:
                ms = rnd.randint(prm["min_shift_ms"], prm["max_shift_ms"])
                if abs(ms) / 1000.0 > n / sample_rate:
                    raise ValueError("|shift_ms| exceeds the clip duration")
                k = int(ms * sample_rate / 1000)
                if k != 0:
                    ops.append((AUG_SHIFT, k, 0, 0, 0.0))
            if kind == "volume" and rnd.random() < prob:
                gain = rnd.randint(prm["min_gain_dBFS"], prm["max_gain_dBFS"])
                ops.append((AUG_VOLUME, 0, 0, gain, float(np.float32(10. ** (gain / 20.)))))
            if kind == "resample" and rnd.random() < prob:
                new_rate = int((np_rnd or np.random).choice(prm["new_sample_rates"]))
                if new_rate != sample_rate:
                    ops.append((HOST_RESAMPLE, new_rate, 0, 0, 0.0))
                    n = resampled_length(n, sample_rate, new_rate)
                    sample_rate = new_rate
            if kind == "noise" and rnd.random() < prob:
                if self.noise_lengths:
                    j = rnd.choice(range(len(self.noise_lengths)))
                    snr = rnd.randint(prm["min_snr_dB"], prm["max_snr_dB"])
                    ln = self.noise_lengths[j]
                    if self.noise_rates is not None:                # librosa.load(noise_path, sr=sample_rate)
                        ln = resampled_length(ln, self.noise_rates[j], sample_rate)
                    start = rnd.randint(0, ln - n) if ln > n else 0
                    ops.append((AUG_NOISE, j, start, snr, 0.0))
            if kind == "reverb" and rnd.random() < prob:
                if self.n_rir_files > 0:
                    ops.append((HOST_REVERB, rnd.choice(range(self.n_rir_files)), 0, 0, 0.0))
        return ops, n
