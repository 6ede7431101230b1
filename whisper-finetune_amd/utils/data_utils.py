"""``from utils.data_utils import DataCollatorSpeechSeq2SeqWithPadding, ...``
(finetune.py:11, evaluation.py:13).  The reference's own file of this name is a byte
copy of evaluation.py and defines none of these names."""
from wfe_amd.data import (DataCollatorSpeechSeq2SeqWithPadding,  # noqa: F401
                          DataCollatorSpeechSeq2SeqWithPaddingforEval, remove_punctuation, to_lower, to_simple)
