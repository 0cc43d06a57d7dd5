"""MI355X-native x-vector engine behind the tf-kaldi-speaker Trainer API.

Sub-packages mirror the reference's import names so that
``PYTHONPATH=<repo>/tf_kaldi_speaker_amd`` lets ``from model.trainer import Trainer``
resolve to this implementation (drop-in for egs/*/nnet/lib/*.py).
"""
__version__ = "0.1.0"
