"""Host-side checks that stand between a bad plan and a GPU memory fault, and the error
contract of the collator (no GPU needed: everything here raises before any device call)."""
import json
import types

import numpy as np
import pytest
import torch

import stubs
from wfe_amd import _lib
from wfe_amd import data as wd
from wfe_amd.frontend import HipFrontEnd


def _bank(lengths):
    return types.SimpleNamespace(n=len(lengths), lengths=list(lengths))


def test_validate_plans_rejects_what_the_device_cannot_check():
    ok = [[(_lib.AUG_NOISE, 1, 5, 20, 0.0), (_lib.AUG_SPEED, 100, 0, 0, 0.0)], [], [(_lib.AUG_AFFINE, 0, 0, 0, 0.0)]]
    HipFrontEnd.validate_plans(ok, _bank([10, 20]))
    HipFrontEnd.validate_plans([[(_lib.AUG_AFFINE, 0, 0, 0, 0.0)]], None)        # do_normalize: no bank needed
    with pytest.raises(ValueError, match="without a noise bank"):
        HipFrontEnd.validate_plans([[(_lib.AUG_NOISE, 0, 0, 20, 0.0)]], None)
    with pytest.raises(ValueError, match="without a noise bank"):
        HipFrontEnd.validate_plans([[(_lib.AUG_NOISE, 0, 0, 20, 0.0)]], _bank([]))
    with pytest.raises(ValueError, match="clip index 2 outside"):
        HipFrontEnd.validate_plans([[(_lib.AUG_NOISE, 2, 0, 20, 0.0)]], _bank([10, 20]))
    with pytest.raises(ValueError, match="clip index -1 outside"):
        HipFrontEnd.validate_plans([[(_lib.AUG_NOISE, -1, 0, 20, 0.0)]], _bank([10, 20]))
    with pytest.raises(ValueError, match="noise start 20 outside"):
        HipFrontEnd.validate_plans([[(_lib.AUG_NOISE, 1, 20, 20, 0.0)]], _bank([10, 20]))
    with pytest.raises(ValueError, match="at most one SPEED"):
        HipFrontEnd.validate_plans([[(_lib.AUG_SPEED, 5, 0, 0, 0.0), (_lib.AUG_SPEED, 4, 0, 0, 0.0)]], None)
    with pytest.raises(ValueError, match="not a device op"):
        HipFrontEnd.validate_plans([[(100, 8000, 0, 0, 0.0)]], None)              # HOST_RESAMPLE never reaches the device


def test_collator_refuses_noise_plan_without_clips():
    col = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=stubs.StubProcessor())
    plans = [[(_lib.AUG_NOISE, 0, 0, 20, 0.0)]]
    with pytest.raises(ValueError, match="refers to noise clips, but the collator has none"):
        col._bank_for(None, 16000, plans, None)
    assert col._bank_for(None, 16000, [[(_lib.AUG_VOLUME, 0, 0, 3, 1.4)]], None) is None
    # two datasets' clip sets in one batch cannot be told apart by index
    a, b = wd.ClipSet([np.zeros(4, np.float32)]), wd.ClipSet([np.zeros(4, np.float32)])
    with pytest.raises(ValueError, match="different noise / reverb clip sets"):
        col._clip_set_of([{"aug_bank": a}, {"aug_bank": b}])
    assert col._clip_set_of([{"aug_bank": a}, {"aug_bank": a}, {}]) is a


def test_collator_in_a_dataloader_worker_raises_with_the_reason(monkeypatch):
    """finetune.py / evaluation.py default to --num_workers 8 (finetune.py:26,129): the collator
    then runs inside a forked worker, which cannot use the GPU.  The failure must be loud and
    name the remedy (wfe_amd.launch injects --num_workers 0)."""
    col = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=stubs.StubProcessor())
    monkeypatch.setattr(torch.utils.data, "get_worker_info", lambda: object())
    item = {"waveform": np.zeros(1600, np.float32), "aug_ops": [], "labels": [1, 2], "id": "x"}
    with pytest.raises(RuntimeError, match="--num_workers 0"):
        col([item])


def test_config_with_two_speed_entries_is_accepted(tmp_path):
    """utils/reader.py:248-254 applies every speed entry; the device chain interpolates once, so the planner
    emits both SPEED ops and split_stages cuts the chain in front of the second (round 3; refused before)."""
    from wfe_amd import _lib
    from wfe_amd.augment import split_stages
    mf = tmp_path / "list.json"
    mf.write_text("")
    cfg = tmp_path / "aug.json"
    speed = {"type": "speed", "prob": 1.0, "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 2}}
    cfg.write_text(json.dumps([speed, speed]))
    ds = wd.CustomDataset(str(mf), stubs.StubProcessor(), augment_config_path=str(cfg))
    assert isinstance(ds.clip_set, wd.ClipSet)
    ops, n = ds.planner.draw(16000, 16000, __import__("random").Random(3))
    assert [o[0] for o in ops] == [_lib.AUG_SPEED, _lib.AUG_SPEED] and n == ops[-1][1]
    st = split_stages(ops, 16000, 16000, 16000)
    assert len(st) == 2 and st[0]["post"] == ("pass", 0) and st[1]["len_in"] == st[0]["len_out"] == ops[0][1]
    with pytest.raises(ValueError, match="at most one SPEED op"):         # a DEVICE chain still holds one
        from wfe_amd.frontend import HipFrontEnd
        HipFrontEnd.validate_plans([ops], None)
