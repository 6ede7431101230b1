"""CPU restatement of the batch collation and label arithmetic (TEST
INFRASTRUCTURE ONLY).  PARITY UNPINNED for the collator: the reference ships no
source for DataCollatorSpeechSeq2SeqWithPadding (its utils/data_utils.py is a
copy of evaluation.py), so this follows the call-site contract of SURVEY 8a
C1/C2 (finetune.py:70,131-132; evaluation.py:63,94-111,131;
utils/reader.py:209-216).  The timestamp label arithmetic follows
utils/reader.py:164-179 and is pinned by hand-computed cases in the tests."""
from __future__ import annotations

import numpy as np


def collate(items, n_mel, frames, strip_tokens=(), with_ids=False):
    """items: dicts with 'input_features' ((1, M, T) array or None), 'labels', 'id'."""
    kept = [it for it in items if it.get("input_features") is not None]
    feats = np.zeros((len(kept), n_mel, frames), dtype=np.float32)
    for i, it in enumerate(kept):
        feats[i] = np.asarray(it["input_features"], dtype=np.float32).reshape(n_mel, frames)
    width = max((len(it["labels"]) for it in kept), default=0)
    labels = np.full((len(kept), width), -100, dtype=np.int64)
    for i, it in enumerate(kept):
        labels[i, :len(it["labels"])] = it["labels"]
    # training variant: a start token shared by every row is dropped (the model prepends its
    # decoder start token when it shifts the labels right); the eval variant keeps the prompt
    if len(kept) and width and any((labels[:, 0] == t).all() for t in strip_tokens):
        labels = labels[:, 1:]
    out = {"input_features": feats, "labels": labels}
    if with_ids:
        out["id"] = [it["id"] for it in kept]
    return out


def timestamp_labels(segments, prefix3, timestamp_begin, endoftext, encode):
    """utils/reader.py:164-179.  ``encode(text)`` returns the full tokenizer ids of
    the text (4 prompt tokens + text + eot); the reference keeps ids[4:-1]."""
    labels = list(prefix3)
    for t in segments:
        start = t["start"] if round(t["start"] * 100) % 2 == 0 else t["start"] + 0.01
        start = timestamp_begin + round(start * 100) // 2
        end = t["end"] if round(t["end"] * 100) % 2 == 0 else t["end"] - 0.01
        end = timestamp_begin + round(end * 100) // 2
        labels += [start] + list(encode(t["text"])[4:-1]) + [end]
    return labels + [endoftext]
