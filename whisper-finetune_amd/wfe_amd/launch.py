"""Runs a Whisper-Finetune script UNCHANGED on the MI355X front end:

    python -m wfe_amd.launch finetune.py --base_model openai/whisper-large-v3 ...
    torchrun --nproc_per_node=8 -m wfe_amd.launch finetune.py ...

What it does, in this process and before anything touches a GPU:

  * ``utils.reader`` and ``utils.data_utils`` -- exactly those two module names -- resolve to
    this package's CustomDataset / collators / text helpers (the reference imports them at
    finetune.py:11,13, finetune_all.py:11,13, evaluation.py:13,15; its own utils/data_utils.py
    defines none of the names it is asked for).  Every other ``utils.*`` module
    (utils.callback, utils.model_utils, utils.utils, utils.binary ...) and the ``utils``
    package itself keep coming from the script's own tree: the script's directory is put at
    ``sys.path[0]`` exactly as ``python script.py`` does.
  * ``--num_workers 0`` is injected (or an explicit other value replaced, loudly) for scripts
    that import the dataset / collator: the collator runs the fused HIP launch, and a forked
    DataLoader worker cannot use the GPU (finetune.py:26,129 and evaluation.py:24,64-65
    default to 8).  Host decode is parallelised inside CustomDataset.__getitems__ instead.
  * with ``WFE_INSTALL_EXTRACTOR=1``, ``WhisperProcessor.from_pretrained`` returns a
    processor whose feature extractor is the MI355X one (infer_tfs.py:27-37, evaluation.py:41-46).

  * for scripts that use the front end: ``WFE_DEVICE_OUTPUT`` defaults to 1 and
    ``Seq2SeqTrainingArguments`` / ``TrainingArguments`` default to ``dataloader_pin_memory=False`` (the
    reference never sets it, finetune.py:111-133), so ``input_features`` stay on the GPU instead of going
    GPU -> host -> pinned -> GPU; the DataLoader of the GPU collator is iterated one batch ahead on a thread
    of the main process (``wfe_amd.data.install_prefetch``; WFE_PREFETCH=0 turns it off), which is what the
    reference's 8 workers were for (finetune.py:26,129).  Keyword names that the installed transformers has
    renamed since the reference was written (``evaluation_strategy`` -> ``eval_strategy``, the Trainer's
    ``tokenizer=`` -> ``processing_class=``; finetune.py:120,149) are passed on under their new names.

The script then runs in-process (``runpy.run_path``, ``__name__ == "__main__"``): no re-exec,
so a torchrun-launched rank stays the process torchrun started.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import runpy
import sys

_REDIRECTED = {
    "utils.reader": ("CustomDataset",),
    "utils.data_utils": ("DataCollatorSpeechSeq2SeqWithPadding", "DataCollatorSpeechSeq2SeqWithPaddingforEval",
                         "remove_punctuation", "to_simple", "to_lower"),
}


class _Redirect(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Meta-path finder + loader for the two redirected names; nothing else is touched."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname in _REDIRECTED:
            return importlib.machinery.ModuleSpec(fullname, self, origin="wfe_amd.data")
        return None

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        from . import data
        for name in _REDIRECTED[module.__name__]:
            setattr(module, name, getattr(data, name))
        module.__file__ = data.__file__
        module.__wfe_amd__ = True


def install_redirect() -> None:
    if not any(isinstance(f, _Redirect) for f in sys.meta_path):
        sys.meta_path.insert(0, _Redirect())


def _uses_front_end(script_path: str) -> bool:
    try:
        with open(script_path, "r", encoding="utf-8", errors="replace") as f:
            text = f.read()
    except OSError:
        return False
    return ("utils.reader" in text or "utils.data_utils" in text) and "num_workers" in text


def force_num_workers_zero(argv: list, note=sys.stderr) -> list:
    """argv of the script with ``--num_workers 0``; says why whenever it changes anything."""
    reason = ("[wfe_amd.launch] --num_workers {} -> 0: the GPU collator runs in the rank's main process (a forked "
              "DataLoader worker cannot use the GPU); host decode is threaded inside CustomDataset (WFE_DECODE_THREADS)")
    def is_flag(name: str) -> bool:
        # argparse (allow_abbrev) also takes the unambiguous prefixes: the reference's scripts define
        # --num_workers and --num_train_epochs, so everything from "--num_w" on means the former
        return len(name) >= len("--num_w") and "--num_workers".startswith(name)

    out, i, seen = [], 0, False
    while i < len(argv):
        a = argv[i]
        if is_flag(a) and i + 1 < len(argv):
            seen = True
            if argv[i + 1] != "0":
                print(reason.format(argv[i + 1]), file=note)
            out += ["--num_workers", "0"]
            i += 2
            continue
        if "=" in a and is_flag(a.split("=", 1)[0]):
            seen = True
            if a.split("=", 1)[1] != "0":
                print(reason.format(a.split("=", 1)[1]), file=note)
            out.append("--num_workers=0")
            i += 1
            continue
        out.append(a)
        i += 1
    if not seen:
        print(reason.format("(default 8)"), file=note)
        out += ["--num_workers", "0"]
    return out


def _install_extractor_hook() -> None:
    from transformers import WhisperProcessor
    from .feature_extractor import install
    original = WhisperProcessor.from_pretrained.__func__

    def from_pretrained(cls, *args, **kwargs):
        return install(original(cls, *args, **kwargs))
    WhisperProcessor.from_pretrained = classmethod(from_pretrained)


def _install_training_defaults(note=sys.stderr) -> None:
    """dataloader_pin_memory=False by default (CUDA tensors cannot be pinned) and the two keyword renames of
    newer transformers releases.  Best effort: without transformers there is nothing to patch."""
    try:
        import inspect
        import transformers
    except Exception:  # noqa: BLE001
        return
    for cls_name in ("TrainingArguments", "Seq2SeqTrainingArguments"):
        cls = getattr(transformers, cls_name, None)
        if cls is None or getattr(cls.__init__, "__wfe_defaults__", False):
            continue
        original = cls.__init__
        params = inspect.signature(original).parameters

        def __init__(self, *args, __orig=original, __params=params, **kw):
            kw.setdefault("dataloader_pin_memory", False)
            if "evaluation_strategy" in kw and "evaluation_strategy" not in __params and "eval_strategy" in __params:
                kw["eval_strategy"] = kw.pop("evaluation_strategy")
            __orig(self, *args, **kw)
        __init__.__wfe_defaults__ = True
        cls.__init__ = __init__
    for cls_name in ("Trainer", "Seq2SeqTrainer"):
        cls = getattr(transformers, cls_name, None)
        if cls is None or getattr(cls.__init__, "__wfe_defaults__", False):
            continue
        original = cls.__init__
        params = inspect.signature(original).parameters

        def __init__(self, *args, __orig=original, __params=params, **kw):
            if "tokenizer" in kw and "tokenizer" not in __params and "processing_class" in __params:
                kw["processing_class"] = kw.pop("tokenizer")
            __orig(self, *args, **kw)
        __init__.__wfe_defaults__ = True
        cls.__init__ = __init__


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if argv else 2)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"wfe_amd.launch: no such script: {argv[0]}")
    args = argv[1:]
    if _uses_front_end(script):
        args = force_num_workers_zero(args)
        os.environ.setdefault("WFE_DEVICE_OUTPUT", "1")      # features stay on the GPU ...
        _install_training_defaults()                         # ... so the Trainer must not try to pin them
        from .data import install_prefetch
        install_prefetch()                                   # batch i + 1 is collated while step i runs
    install_redirect()
    if os.environ.get("WFE_INSTALL_EXTRACTOR", "0") == "1":
        _install_extractor_hook()
    # as `python script.py`: the script's directory leads sys.path, argv[0] is the script
    script_dir = os.path.dirname(script)
    sys.path.insert(0, script_dir)               # (-m put the cwd first; the script's directory goes ahead of it)
    sys.argv = [script] + args
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
