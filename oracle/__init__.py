"""CPU oracle for the audio front-end hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The shipped path
(``whisper-finetune_amd/``) never imports this package and fails loudly when
its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * logmel.py   -- pinned against golden vectors produced in the build
                   container by the installed third-party
                   ``transformers.WhisperFeatureExtractor`` (5.15.0), the class
                   the reference calls at utils/reader.py:192,195,198, and by the
                   reference's own data file convert-ggml/mel_filters.npz.
  * augment.py  -- speed/shift/volume/rms_db/noise mixing/draw order pinned
                   against outputs of the reference's utils/reader.py executed
                   in the build container (tests/golden/make_golden.py);
                   resample is PARITY UNPINNED (librosa/soxr absent).
  * collate.py  -- PARITY UNPINNED: the reference ships no collator source
                   (utils/data_utils.py is a copy of evaluation.py); restated
                   from the call-site contract only.
"""
