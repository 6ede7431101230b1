"""Drop-in ``WhisperFeatureExtractor`` whose arithmetic runs on the MI355X.

``MI355XWhisperFeatureExtractor`` IS a ``transformers.WhisperFeatureExtractor``
(subclass): attributes, ``pad``, ``save_pretrained`` / ``from_pretrained`` and the
``WhisperProcessor`` type check are inherited, so ``finetune.py``,
``finetune_all.py``, ``evaluation.py`` and ``infer_tfs.py`` keep working on a
processor whose ``feature_extractor`` was swapped for this class
(``install(processor)``).  Only ``__call__`` is replaced: instead of padding on
the host and running torch.stft on CPU threads (transformers/models/whisper/
feature_extraction_whisper.py:193-346, "HF" below) the raw utterances go to the
GPU ragged, and truncation, zero/reflect padding, STFT, mel, log and clamp happen
in the HIP kernels (include/wfe.h ``wfe_logmel``).

There is no CPU fallback: without the HIP library or a gfx950 device ``__call__``
raises.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from transformers import WhisperFeatureExtractor
from transformers.feature_extraction_utils import BatchFeature

from . import _lib
from .frontend import HOP, HipFrontEnd, pack_ragged

_FRONT_ENDS: dict = {}      # (device index, n_mel) -> HipFrontEnd; kept out of the instance so that
                            # to_dict / save_pretrained / pickling never see a device handle


def front_end(n_mel: int, device=None) -> HipFrontEnd:
    if not torch.cuda.is_available():
        raise _lib.WfeError("MI355XWhisperFeatureExtractor needs an MI355X: there is no CPU fallback")
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    key = (idx, int(n_mel))
    if key not in _FRONT_ENDS:
        _FRONT_ENDS[key] = HipFrontEnd(n_mel=n_mel, device=idx)
    return _FRONT_ENDS[key]


def plan_lengths(lengths, n_samples: int, truncation=True, pad_to_multiple_of=None, padding="max_length",
                 max_length=None):
    """Restates the length arithmetic of SequenceFeatureExtractor.pad/_truncate/_pad
    (feature_extraction_sequence_utils.py:51-337) as called from HF:300-307.
    Returns (target length, kept length per utterance)."""
    ml = max_length if max_length else n_samples
    if pad_to_multiple_of is not None and ml % pad_to_multiple_of != 0:
        ml = (ml // pad_to_multiple_of + 1) * pad_to_multiple_of
    kept = [min(int(n), ml) if truncation else int(n) for n in lengths]
    if padding is True or padding == "longest":
        target = max(kept) if kept else 0
        if pad_to_multiple_of is not None and target % pad_to_multiple_of != 0:
            target = (target // pad_to_multiple_of + 1) * pad_to_multiple_of
    elif padding == "max_length":
        target = ml
    elif padding is False or padding == "do_not_pad":
        target = max(kept) if kept else 0
    else:
        raise ValueError(f"unknown padding strategy {padding!r}")
    if any(k > target for k in kept) or ((padding is False or padding == "do_not_pad") and len(set(kept)) > 1):
        # the reference fails here too (np.stack of ragged rows, HF:316-319)
        raise ValueError("utterances do not share one length after padding/truncation; "
                         "use padding='max_length' or 'longest' with truncation=True")
    return target, kept


class MI355XWhisperFeatureExtractor(WhisperFeatureExtractor):
    """Same constructor and call signature as WhisperFeatureExtractor (HF:69-103,
    193-206).  Extra, optional: ``device`` (GPU ordinal, default current) and
    ``device_output`` -- when True, ``return_tensors="pt"`` yields tensors that
    stay on the GPU (the reference always returns CPU data)."""

    def __init__(self, *args, device=None, device_output: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        if self.n_fft != 400 or self.hop_length != 160 or self.sampling_rate != 16000:
            raise ValueError("the HIP front end implements n_fft=400, hop_length=160, sampling_rate=16000 only")
        if not 1 <= self.feature_size <= 128:
            raise ValueError("feature_size must be within 1..128")
        self._device = device
        self._device_output = bool(device_output)

    # ---- persistence: checkpoints stay loadable by stock transformers ----------
    def to_dict(self):
        d = super().to_dict()
        d.pop("_device", None)
        d.pop("_device_output", None)
        d["feature_extractor_type"] = "WhisperFeatureExtractor"
        return d

    @classmethod
    def from_extractor(cls, fe: WhisperFeatureExtractor, device=None, device_output=False):
        return cls(feature_size=fe.feature_size, sampling_rate=fe.sampling_rate, hop_length=fe.hop_length,
                   chunk_length=fe.chunk_length, n_fft=fe.n_fft, padding_value=fe.padding_value,
                   dither=getattr(fe, "dither", 0.0), return_attention_mask=fe.return_attention_mask,
                   device=device, device_output=device_output)

    # ---- the hot path ---------------------------------------------------------
    def __call__(self, raw_speech, truncation: bool = True, pad_to_multiple_of: Optional[int] = None,
                 return_tensors=None, return_attention_mask: Optional[bool] = None, padding="max_length",
                 max_length: Optional[int] = None, sampling_rate: Optional[int] = None,
                 do_normalize: Optional[bool] = None, device=None, **kwargs) -> BatchFeature:
        if sampling_rate is not None and sampling_rate != self.sampling_rate:       # HF:265-271
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a"
                f" sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input"
                f" was sampled with {self.sampling_rate} and not {sampling_rate}.")
        # HF:278-294 input normalisation: always a batch, float32, mono
        if isinstance(raw_speech, torch.Tensor):
            raw_speech = raw_speech.detach().cpu().numpy()
        is_batched_numpy = isinstance(raw_speech, np.ndarray) and raw_speech.ndim > 1
        if is_batched_numpy and raw_speech.ndim > 2:
            raise ValueError(f"Only mono-channel audio is supported for input to {self}")
        is_batched = is_batched_numpy or (
            isinstance(raw_speech, (list, tuple)) and len(raw_speech) > 0
            and isinstance(raw_speech[0], (np.ndarray, tuple, list)))
        waves = list(raw_speech) if is_batched else [raw_speech]
        waves = [np.asarray(w, dtype=np.float32).reshape(-1) for w in waves]

        target, kept = plan_lengths([len(w) for w in waves], self.n_samples, truncation, pad_to_multiple_of,
                                    padding, max_length)
        fe = front_end(self.feature_size, self._device if device in (None, "cpu") else device)
        flat, off, lens = pack_ragged([w[:k] for w, k in zip(waves, kept)], fe.device)
        dither = float(getattr(self, "dither", 0.0) or 0.0)
        if dither != 0.0:
            # HF:146-147: `waveform += dither * torch.randn(waveform.shape)` on the PADDED batch (the zero padding
            # is dithered too), drawn from torch's global CPU generator.  The draw stays on the host -- the same
            # generator, shape and order, so a seeded run sees the reference's noise -- the multiply and the add
            # are the same two float32 roundings on the GPU, and the dithered batch goes through the kernels at
            # full length (no reference caller sets dither: not a hot path, one extra H2D copy).
            if do_normalize:                                                       # HF:310-316, before the dither
                plans = [[(_lib.AUG_AFFINE, 0, 0, 0, 0.0)] for _ in waves]
                dense, _ = fe.augment_waveform(flat, lens, plans, None, n_samples=target, offsets=off)
            else:
                host = np.zeros((len(waves), target), dtype=np.float32)
                for i, (w, k) in enumerate(zip(waves, kept)):
                    host[i, :k] = w[:k]
                dense = torch.from_numpy(host).to(fe.device)
            noise = torch.randn((len(waves), target), dtype=torch.float32)
            dense.add_(noise.to(fe.device).mul_(dither))
            full = torch.full((len(waves),), target, dtype=torch.int32, device=fe.device)
            feats = fe.logmel(dense, full, n_samples=target)
        elif do_normalize:                                                         # HF:310-316
            plans = [[(_lib.AUG_AFFINE, 0, 0, 0, 0.0)] for _ in waves]
            feats = fe.augment_logmel(flat, lens, plans, None, n_samples=target, offsets=off)
        else:
            feats = fe.logmel(flat, lens, n_samples=target, offsets=off)

        data = {}
        # HF:300-307,332-341: pad() is asked for the sample-level mask when
        # return_attention_mask OR do_normalize is set (None otherwise falls back to the
        # extractor default); it is rescaled to frames only when return_attention_mask is
        # truthy -- so do_normalize alone leaves the sample-level mask in the output.
        asked = return_attention_mask or do_normalize
        has_mask = asked if asked is not None else self.return_attention_mask
        if has_mask:
            mask = np.zeros((len(waves), target), dtype=np.int32)
            for i, k in enumerate(kept):
                mask[i, :k] = 1
            if return_attention_mask:
                mask = mask[:, ::HOP]
                if target % HOP != 0:
                    mask = mask[:, :-1]
            data["attention_mask"] = mask
        if return_tensors in ("pt", "torch") and self._device_output:
            data["input_features"] = feats
            if "attention_mask" in data:
                data["attention_mask"] = torch.from_numpy(data["attention_mask"]).to(feats.device)
            return BatchFeature(data)
        data["input_features"] = feats.cpu().numpy()
        out = BatchFeature(data)
        if return_tensors is not None:
            out = out.convert_to_tensors(return_tensors)
        return out


def install(processor, device=None, device_output: bool = False):
    """Swap ``processor.feature_extractor`` for the MI355X one (same config) and
    return the processor: the one line a maintainer adds after
    ``WhisperProcessor.from_pretrained`` (finetune.py:47-52, evaluation.py:41-46)."""
    processor.feature_extractor = MI355XWhisperFeatureExtractor.from_extractor(
        processor.feature_extractor, device=device, device_output=device_output)
    return processor
