"""Device-side driver of the HIP front end: one ``HipFrontEnd`` per (GPU, n_mel).

torch is used for device memory and streams only; every transform below is a
call through the C ABI (include/wfe.h) into hand-written gfx950 kernels.  There
is no CPU path: constructing a ``HipFrontEnd`` without a usable MI355X raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

N_FFT = 400
HOP = 160
N_FREQ = 201
N_SAMPLES = 480000


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class NoiseBank:
    """Noise clips resident on the device, back to back (the decoded contents of
    ``CustomDataset.noise_files``, utils/reader.py:75-78,323)."""

    _serial = 0

    @classmethod
    def _next_serial(cls) -> int:
        NoiseBank._serial += 1
        return NoiseBank._serial

    def __init__(self, clips: Sequence[np.ndarray], device: torch.device):
        self.serial = self._next_serial()       # identity of the CONTENTS (a freed bank's address may be reused)
        self.lengths = [int(len(c)) for c in clips]
        off = np.zeros(len(clips) + 1, dtype=np.int64)
        off[1:] = np.cumsum(self.lengths)
        flat = np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1) for c in clips]) if clips else \
            np.zeros(0, dtype=np.float32)
        self.data = torch.from_numpy(flat).to(device)
        self.offsets = torch.from_numpy(off).to(device)
        self.n = len(clips)

    @classmethod
    def from_tensors(cls, clips: Sequence[torch.Tensor], device: torch.device) -> "NoiseBank":
        """Clips that already live on the device (e.g. resampled there)."""
        self = cls.__new__(cls)
        self.serial = cls._next_serial()
        self.lengths = [int(c.numel()) for c in clips]
        off = np.zeros(len(clips) + 1, dtype=np.int64)
        off[1:] = np.cumsum(self.lengths)
        self.data = torch.cat([c.reshape(-1).to(device=device, dtype=torch.float32) for c in clips]) if clips else \
            torch.zeros(0, dtype=torch.float32, device=device)
        self.offsets = torch.from_numpy(off).to(device)
        self.n = len(clips)
        return self


class HipFrontEnd:
    def __init__(self, n_mel: int = 80, device: int | torch.device | str | None = None,
                 sampling_rate: int = 16000, n_fft: int = N_FFT, hop_length: int = HOP):
        self._h = None
        if not torch.cuda.is_available():
            raise _lib.WfeError("no MI355X visible to this process: the HIP front end has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if dev.type != "cuda":
            raise _lib.WfeError(f"device {dev} is not a GPU")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.n_mel = int(n_mel)
        cfg = _lib.WfeConfig(self.n_mel, int(n_fft), int(hop_length), int(sampling_rate), self.device.index)
        h = C.c_void_p()
        _lib.check(_lib.lib().wfe_create(C.byref(cfg), C.byref(h)), None)
        self._h = h
        self.out_dtype = torch.float32
        self._bank_serial = None          # the NoiseBank whose mean squares the handle has cached
        self._rir_serial = None           # the impulse-response bank whose spectra it has cached (reverb)

    _OUT_FORMATS = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}     # wfe_out_format

    def set_output_dtype(self, dtype: torch.dtype) -> None:
        """Element type of the log-mel output of logmel / augment_logmel: float32 (the
        reference's), or float16 / bfloat16 = the float32 result rounded to nearest even,
        written directly (the Trainer casts input_features to the training
        dtype anyway, finetune.py:122)."""
        if dtype not in self._OUT_FORMATS:
            raise ValueError(f"output dtype {dtype} is not float32, float16 or bfloat16")
        _lib.check(_lib.lib().wfe_set_output_format(self._h, self._OUT_FORMATS[dtype]), self._h)
        self.out_dtype = dtype

    def close(self):
        if self._h is not None:
            _lib.lib().wfe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def mel_filters(self) -> np.ndarray:
        out = np.zeros((N_FREQ, self.n_mel), dtype=np.float64)
        _lib.check(_lib.lib().wfe_mel_filters(self._h, out.ctypes.data), self._h)
        return out

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_inputs(self, wav, lengths, offsets, batch):
        if wav.dtype not in (torch.float32, torch.int16) or wav.device != self.device or not wav.is_contiguous():
            raise ValueError(f"wav must be a contiguous float32 (or int16 PCM) tensor on {self.device}")
        if lengths.device != self.device or lengths.dtype != torch.int32 or not lengths.is_contiguous():
            raise ValueError(f"lengths must be a contiguous int32 tensor on {self.device}")
        if lengths.numel() != batch:
            raise ValueError("lengths must have one entry per utterance")
        if offsets is not None and (offsets.device != self.device or offsets.dtype != torch.int64
                                    or offsets.numel() != batch):
            raise ValueError("offsets must be int64[batch] on the device")

    def _geometry(self, wav, lengths, offsets):
        if offsets is None:
            if wav.dim() != 2:
                raise ValueError("wav must be (batch, stride) when offsets is None")
            return wav.shape[0], wav.shape[1]
        return lengths.numel(), 0

    def kernel_timing(self, enable: bool) -> None:
        _lib.check(_lib.lib().wfe_kernel_timing(self._h, 1 if enable else 0), self._h)

    def kernel_times(self):
        """(frontend_ms, clamp_ms, launches) summed since the last query; synchronises."""
        f, c, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(_lib.lib().wfe_kernel_times(self._h, C.byref(f), C.byref(c), C.byref(n)), self._h)
        return f.value, c.value, n.value

    def _out(self, out, shape):
        if out is None:
            return torch.empty(shape, dtype=self.out_dtype, device=self.device)
        if tuple(out.shape) != shape or out.dtype != self.out_dtype or not out.is_contiguous() \
                or out.device != self.device:
            raise ValueError(f"out must be a contiguous {self.out_dtype} {shape} tensor on {self.device}")
        return out

    # ---------------------------------------------------------------- hot path
    def logmel(self, wav: torch.Tensor, lengths: torch.Tensor, n_samples: int = N_SAMPLES,
               offsets: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
               channels: int = 1) -> torch.Tensor:
        """(B, n_mel, n_samples // 160) float32 (or set_output_dtype's) on the device.  ``wav`` is (B, stride)
        or, with ``offsets``, a flat buffer of ragged utterances; float32 samples, or
        int16 PCM with ``channels`` interleaved channels (lengths in frames, offsets and
        stride in int16 elements), converted and mixed down on the device."""
        batch, stride = self._geometry(wav, lengths, offsets)
        self._check_inputs(wav, lengths, offsets, batch)
        shape = (batch, self.n_mel, n_samples // HOP)
        out = self._out(out, shape)
        if wav.dtype == torch.int16:
            _lib.check(_lib.lib().wfe_logmel_pcm16(
                self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride, int(channels), batch, int(n_samples),
                None, None, 0, None, None, 0, _ptr(out), self._stream()), self._h)
            return out
        _lib.check(_lib.lib().wfe_logmel(self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride, batch,
                                         int(n_samples), _ptr(out), self._stream()), self._h)
        return out

    def power_spectrum(self, wav: torch.Tensor, lengths: torch.Tensor, n_samples: int = N_SAMPLES,
                       offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
        batch, stride = self._geometry(wav, lengths, offsets)
        self._check_inputs(wav, lengths, offsets, batch)
        out = torch.empty((batch, N_FREQ, n_samples // HOP), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().wfe_power_spectrum(self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride, batch,
                                                 int(n_samples), _ptr(out), self._stream()), self._h)
        return out

    # -------------------------------------------------------------- resampling
    @staticmethod
    def resampled_length(n: int, orig_sr: int, target_sr: int) -> int:
        """librosa.resample's output length: int(ceil(n * (float(target_sr) / orig_sr)))."""
        return int(np.ceil(n * (float(target_sr) / orig_sr)))

    def resample(self, wav: torch.Tensor, lengths: torch.Tensor, orig_sr: int, target_sr: int,
                 offsets: Optional[torch.Tensor] = None, channels: int = 1, out_stride: Optional[int] = None,
                 host_lengths: Optional[Sequence[int]] = None):
        """CustomDataset.resample (reader.py:316-319) for a batch -> (out (B, out_stride)
        float32, out_lengths int32), both on the device; rows are zero behind their
        length.  Input as for ``logmel`` (float32 or int16 PCM).  ``host_lengths``
        saves the device read-back of ``lengths``."""
        if int(orig_sr) < 1 or int(target_sr) < 1:
            raise ValueError(f"bad sampling rates {orig_sr} -> {target_sr}")
        batch, stride = self._geometry(wav, lengths, offsets)
        self._check_inputs(wav, lengths, offsets, batch)
        if host_lengths is None:
            host_lengths = lengths.cpu().tolist()
        out_len = [self.resampled_length(max(int(n), 0), orig_sr, target_sr) for n in host_lengths]
        if out_stride is None:
            out_stride = max(max(out_len, default=1), 1)
            out_stride = (out_stride + 3) // 4 * 4                    # rows stay 16-byte aligned
        out = torch.empty((batch, int(out_stride)), dtype=torch.float32, device=self.device)
        out_lengths = torch.tensor([min(n, int(out_stride)) for n in out_len], dtype=torch.int32).to(self.device)
        _lib.check(_lib.lib().wfe_resample(
            self._h, _ptr(wav), int(channels) if wav.dtype == torch.int16 else 0, _ptr(offsets), _ptr(lengths), stride,
            batch, int(orig_sr), int(target_sr), _ptr(out), int(out_stride), _ptr(out_lengths), self._stream()), self._h)
        return out, out_lengths

    # ------------------------------------------------------------ reverberation
    def reverb(self, wav: torch.Tensor, lengths: torch.Tensor, rirs: "NoiseBank", rir_index: Sequence[int],
               offsets: Optional[torch.Tensor] = None, out_stride: Optional[int] = None,
               host_lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        """CustomDataset.add_reverb (reader.py:342-348) for a batch of float32 utterances:
        (B, out_stride) float32, rows zero behind their length (lengths do not change).
        ``rirs``: impulse responses on the device (a NoiseBank); ``rir_index[b] < 0``
        copies utterance b through."""
        batch, stride = self._geometry(wav, lengths, offsets)
        self._check_inputs(wav, lengths, offsets, batch)
        if wav.dtype != torch.float32:
            raise ValueError("reverb takes float32 utterances")
        if len(rir_index) != batch:
            raise ValueError("one impulse-response index per utterance")
        if any(int(i) >= rirs.n for i in rir_index):
            raise ValueError("impulse-response index out of range")
        if out_stride is None:
            if host_lengths is None:
                host_lengths = lengths.cpu().tolist()
            out_stride = (max(max(host_lengths, default=1), 1) + 3) // 4 * 4
        out = torch.empty((batch, int(out_stride)), dtype=torch.float32, device=self.device)
        idx = torch.tensor([int(i) for i in rir_index], dtype=torch.int32).to(self.device)
        if rirs.serial != self._rir_serial:           # (the library keeps the bank's spectra by pointer, as _use_bank)
            _lib.check(_lib.lib().wfe_noise_bank_changed(self._h), self._h)
            self._rir_serial = rirs.serial
        _lib.check(_lib.lib().wfe_reverb(self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride, batch,
                                         _ptr(rirs.data), _ptr(rirs.offsets), rirs.n, _ptr(idx), _ptr(out),
                                         int(out_stride), self._stream()), self._h)
        return out

    # ------------------------------------------------------------ augmentation
    _OP_DTYPE = np.dtype([("kind", "<i4"), ("i0", "<i4"), ("i1", "<i4"), ("i2", "<i4"),
                          ("f0", "<f4"), ("f1", "<f4"), ("d0", "<f8")])     # == wfe_aug_op, 32 bytes

    def pack_plans(self, plans: Sequence[Sequence[tuple]], noise: Optional["NoiseBank"] = None):
        """plans[b] = list of (kind, i0, i1, i2, f0) -> (device op array, device op_begin,
        total): the arguments of wfe_augment_logmel.  The result may be passed to
        augment_logmel / augment_waveform in place of ``plans``; give the bank those calls
        will use as ``noise`` to have the plans validated against it here."""
        if noise is not None:
            self.validate_plans(plans, noise)
        counts = np.fromiter((len(p) for p in plans), dtype=np.int64, count=len(plans))
        total = int(counts.sum())
        arr = np.zeros(max(total, 1), dtype=self._OP_DTYPE)
        if total:
            flat = [op for plan in plans for op in plan]
            arr["kind"][:total] = [o[0] for o in flat]
            arr["i0"][:total] = [o[1] for o in flat]
            arr["i1"][:total] = [o[2] for o in flat]
            arr["i2"][:total] = [o[3] for o in flat]
            arr["f0"][:total] = [o[4] for o in flat]
        begin = np.zeros(len(plans) + 1, dtype=np.int32)
        np.cumsum(counts, out=begin[1:])
        return (torch.from_numpy(arr.view(np.uint8)).to(self.device), torch.from_numpy(begin).to(self.device), total)

    @staticmethod
    def validate_plans(plans, noise: Optional["NoiseBank"]) -> None:
        """Host-side check of op lists before they go to the device, where nothing can be
        checked any more (a NOISE op indexes the noise bank through device pointers: a
        missing bank or an index outside it would be a GPU memory fault, not an error)."""
        for b, plan in enumerate(plans):
            n_speed = 0
            for op in plan:
                kind = int(op[0])
                if kind not in (_lib.AUG_SPEED, _lib.AUG_SHIFT, _lib.AUG_VOLUME, _lib.AUG_NOISE, _lib.AUG_AFFINE):
                    raise ValueError(f"utterance {b}: op kind {kind} is not a device op (wfe_aug_kind)")
                if kind == _lib.AUG_SPEED:
                    n_speed += 1
                    if int(op[1]) < 0 or n_speed > 1:
                        raise ValueError(f"utterance {b}: a chain holds at most one SPEED op, with a length >= 0")
                elif kind == _lib.AUG_NOISE:
                    if noise is None or noise.n == 0:
                        raise ValueError(f"utterance {b}: NOISE op without a noise bank")
                    j, start = int(op[1]), int(op[2])
                    if not 0 <= j < noise.n:
                        raise ValueError(f"utterance {b}: noise clip index {j} outside the bank of {noise.n}")
                    if not 0 <= start <= max(noise.lengths[j] - 1, 0):
                        raise ValueError(f"utterance {b}: noise start {start} outside clip {j} "
                                         f"({noise.lengths[j]} samples)")

    def _use_bank(self, noise: Optional["NoiseBank"]) -> None:
        """The library keeps the mean squares of the last noise bank it saw, keyed by its device
        pointers (include/wfe.h wfe_noise_bank_changed).  torch's allocator hands a freed bank's
        address to the next one, so the key here is the NoiseBank OBJECT: a different one -- or one
        whose ``data`` was rewritten in place, after ``noise.serial`` was bumped -- drops the cache."""
        if noise is None:
            return
        if noise.serial != self._bank_serial:
            _lib.check(_lib.lib().wfe_noise_bank_changed(self._h), self._h)
            self._bank_serial = noise.serial

    def _pack_ops(self, plans, noise=None):
        """Host op lists are validated and packed; a tuple from ``pack_plans`` is taken as it is
        (it was validated when it was packed IF a bank was given there)."""
        if isinstance(plans, tuple) and len(plans) == 3 and isinstance(plans[0], torch.Tensor):
            return plans
        self.validate_plans(plans, noise)
        return self.pack_plans(plans)

    def augment_logmel(self, wav, lengths, plans, noise: Optional[NoiseBank] = None, n_samples: int = N_SAMPLES,
                       offsets=None, out=None, channels: int = 1) -> torch.Tensor:
        batch, stride = self._geometry(wav, lengths, offsets)
        self._check_inputs(wav, lengths, offsets, batch)
        ops, begin, total = self._pack_ops(plans, noise)
        if begin.numel() != batch + 1:
            raise ValueError("one op list per utterance")
        shape = (batch, self.n_mel, n_samples // HOP)
        out = self._out(out, shape)
        self._use_bank(noise)
        tail = (_ptr(ops), _ptr(begin), total, _ptr(noise.data) if noise else None,
                _ptr(noise.offsets) if noise else None, noise.n if noise else 0, _ptr(out), self._stream())
        if wav.dtype == torch.int16:
            _lib.check(_lib.lib().wfe_logmel_pcm16(self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride,
                                                   int(channels), batch, int(n_samples), *tail), self._h)
        else:
            _lib.check(_lib.lib().wfe_augment_logmel(self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride, batch,
                                                     int(n_samples), *tail), self._h)
        return out

    def augment_waveform(self, wav, lengths, plans, noise: Optional[NoiseBank] = None, n_samples: int = N_SAMPLES,
                         offsets=None):
        batch, stride = self._geometry(wav, lengths, offsets)
        self._check_inputs(wav, lengths, offsets, batch)
        ops, begin, total = self._pack_ops(plans, noise)
        out = torch.empty((batch, n_samples), dtype=torch.float32, device=self.device)
        out_len = torch.empty((batch,), dtype=torch.int32, device=self.device)
        self._use_bank(noise)
        _lib.check(_lib.lib().wfe_augment_waveform(
            self._h, _ptr(wav), _ptr(offsets), _ptr(lengths), stride, batch, int(n_samples), _ptr(ops), _ptr(begin),
            total, _ptr(noise.data) if noise else None, _ptr(noise.offsets) if noise else None,
            noise.n if noise else 0, _ptr(out), _ptr(out_len), self._stream()), self._h)
        return out, out_len


def pack_ragged(waves: Sequence[np.ndarray], device: torch.device, pinned: bool = False):
    """Host list of float32 utterances -> (flat device buffer, int64 offsets,
    int32 lengths).  Each utterance starts on a 16-byte boundary so that the
    kernel's float4 staging path applies."""
    lens = np.array([len(w) for w in waves], dtype=np.int64)
    padded = (lens + 3) // 4 * 4
    off = np.zeros(len(waves), dtype=np.int64)
    if len(waves) > 1:
        off[1:] = np.cumsum(padded)[:-1]
    total = int(padded.sum()) if len(waves) else 0
    flat = torch.zeros(max(total, 4), dtype=torch.float32, pin_memory=pinned)
    fnp = flat.numpy()
    for o, w in zip(off, waves):
        fnp[o:o + len(w)] = np.asarray(w, dtype=np.float32).reshape(-1)
    return (flat.to(device, non_blocking=pinned), torch.from_numpy(off).to(device),
            torch.from_numpy(lens.astype(np.int32)).to(device))


class RaggedPacker:
    """Host list of utterances -> one pinned staging buffer -> ONE host-to-device
    copy per batch.  Utterances are float32 (N,) or 16-bit PCM int16 (frames,
    channels) as ``read_pcm16`` maps them; a batch that is all int16 with one channel
    count travels as int16 (half the bytes, no host conversion) and is converted on
    the GPU, any other batch as float32 mono.  The int64 offsets and int32 lengths
    ride in the same buffer.  Two pinned buffers alternate so that batch i+1 can be
    packed while the copy of batch i is still in flight; each is reused only after
    its copy's event has completed."""

    def __init__(self, device: torch.device, threads: int = 8):
        self.device = device
        self._bufs = [None, None]
        self._events = [None, None]
        self._turn = 0
        self._threads = max(1, int(threads))
        self._pool = None

    @staticmethod
    def _as_float_mono(w: np.ndarray) -> np.ndarray:
        if w.dtype == np.int16:             # soundfile float32 decode + librosa.to_mono, on the host
            f = np.asarray(w).astype(np.float32) / np.float32(32768.0)
            return f.reshape(-1) if f.ndim == 1 or f.shape[1] == 1 else np.mean(f.T, axis=0)
        return np.asarray(w, dtype=np.float32).reshape(-1)

    def pack(self, waves: Sequence[np.ndarray], force_float: bool = False):
        """-> (flat device tensor, int64 offsets, int32 lengths, channels); offsets
        count elements of ``flat``, lengths count frames."""
        n = len(waves)
        pcm = n > 0 and not force_float and all(getattr(w, "dtype", None) == np.int16 for w in waves)
        ch = 1
        if pcm:
            chans = {1 if w.ndim == 1 else w.shape[1] for w in waves}
            pcm = len(chans) == 1
            ch = chans.pop() if pcm else 1
        if not pcm:
            waves = [self._as_float_mono(w) for w in waves]
        item = 2 if pcm else 4
        dtype = np.int16 if pcm else np.float32
        frames = np.array([w.shape[0] for w in waves], dtype=np.int64)
        elems = frames * ch
        per16 = 16 // item                                       # utterances start on 16-byte boundaries
        padded = (elems + per16 - 1) // per16 * per16
        off = np.zeros(n, dtype=np.int64)
        if n > 1:
            off[1:] = np.cumsum(padded)[:-1]
        total = int(padded.sum()) if n else 0
        head = (12 * n + 15) // 16 * 16                          # offsets (8n) + lengths (4n), padded
        nbytes = head + max(total, per16) * item
        k = self._turn
        self._turn ^= 1
        if self._events[k] is not None:
            self._events[k].synchronize()
        if self._bufs[k] is None or self._bufs[k].numel() < nbytes:
            self._bufs[k] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        host = self._bufs[k].numpy()
        host[:8 * n].view(np.int64)[:] = off
        host[8 * n:12 * n].view(np.int32)[:] = frames.astype(np.int32)
        body = host[head:head + total * item].view(dtype)

        def put(i):
            w = waves[i]
            if hasattr(w, "read_into"):                          # data.PcmRef: file -> pinned buffer, one pread
                w.read_into(body[off[i]:off[i] + elems[i]])
            else:
                body[off[i]:off[i] + elems[i]] = w.reshape(-1)
            body[off[i] + elems[i]:off[i] + padded[i]] = 0

        if n >= 4 and total * item >= (8 << 20) and self._threads > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self._threads)
            list(self._pool.map(put, range(n)))
        else:
            for i in range(n):
                put(i)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        dev.copy_(self._bufs[k][:nbytes], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._events[k] = ev
        flat = dev[head:head + max(total, per16) * item].view(torch.int16 if pcm else torch.float32)
        return flat, dev[:8 * n].view(torch.int64), dev[8 * n:12 * n].view(torch.int32), ch
