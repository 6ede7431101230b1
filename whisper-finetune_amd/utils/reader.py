"""``from utils.reader import CustomDataset`` (finetune.py:13, evaluation.py:15)."""
from wfe_amd.data import CustomDataset  # noqa: F401
