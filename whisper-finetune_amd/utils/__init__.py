"""Drop-in module names of the reference (``utils.reader``, ``utils.data_utils``):
put ``whisper-finetune_amd/`` ahead of the reference checkout on ``PYTHONPATH`` and
``finetune.py`` / ``finetune_all.py`` / ``evaluation.py`` import these instead."""
