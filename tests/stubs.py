"""A stand-in WhisperProcessor for tests: the real tokenizer needs vocabulary files
that are not available offline.  Token ids follow the multilingual
Whisper layout; text is tokenised per character."""
from transformers import WhisperFeatureExtractor

SOT, EOT, EN, ZH, TRANSCRIBE, NOSPEECH, NOTIMESTAMPS = 50258, 50257, 50259, 50260, 50359, 50362, 50363


class _Enc:
    def __init__(self, ids):
        self.input_ids = ids


class StubTokenizer:
    bos_token_id = EOT          # HF's WhisperTokenizer: bos_token is <|endoftext|>
    pad_token_id = EOT

    def __init__(self):
        self.language = EN

    def get_vocab(self):
        return {"<|startoftranscript|>": SOT, "<|endoftext|>": EOT, "<|notimestamps|>": NOTIMESTAMPS,
                "<|nospeech|>": NOSPEECH, "<|en|>": EN, "<|zh|>": ZH, "<|transcribe|>": TRANSCRIBE}

    @property
    def prefix_tokens(self):
        return [SOT, self.language, TRANSCRIBE, NOTIMESTAMPS]

    def set_prefix_tokens(self, language=None, task=None, predict_timestamps=None):
        self.language = {"zh": ZH, "chinese": ZH}.get((language or "en").lower(), EN)

    def __call__(self, text):
        return _Enc(self.prefix_tokens + [1000 + (ord(c) % 5000) for c in text] + [EOT])


class StubProcessor:
    def __init__(self, n_mel=80, feature_extractor=None):
        self.tokenizer = StubTokenizer()
        self.feature_extractor = feature_extractor or WhisperFeatureExtractor(feature_size=n_mel)

    def __call__(self, audio=None, sampling_rate=None, text=None):
        """The routing of WhisperProcessor.__call__ (transformers processing_whisper.py:31-54):
        audio -> feature extractor, text -> tokenizer, both -> features with ``labels``.  Only the
        golden generator calls this (the reference's CustomDataset runs the processor per item,
        reader.py:174,192,195,198); the GPU dataset never does."""
        from transformers.feature_extraction_utils import BatchFeature
        if audio is None and text is None:
            raise ValueError("You need to specify either an `audio` or `text` input to process.")
        inputs = self.feature_extractor(audio, sampling_rate=sampling_rate) if audio is not None else None
        enc = BatchFeature({"input_ids": self.tokenizer(text).input_ids}) if text is not None else None
        if text is None:
            return inputs
        if audio is None:
            return enc
        inputs["labels"] = enc["input_ids"]
        return inputs
