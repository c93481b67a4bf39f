"""MI355X-native hot path of the EEG imagined-speech decoder (isd_amd).

Feature extraction (Butterworth filterbank -> STFT -> log band power) and the
small-CNN classifier forward/backward as hand-written HIP kernels for gfx950
behind a C ABI (include/isd_hip.h), with the reference's Python call surface
on top: ``extract_features(trials)``, ``clf.fit(X, y)`` / ``clf.predict(X)``.
Importing this package does not touch the GPU and does not load the library.
"""
from .constants import BANDS_5, BANDS_9, BANDS_40, CLASSES, ELECTRODES, ZONES, zone_index_lists  # noqa: F401
from .features import FeatureExtractor, Filterbank, Stft, band_bins, extract_features  # noqa: F401
from .bandpass import FirFilter, filter_data  # noqa: F401
from .filter_design import butter_bandpass_resonators, butter_bandpass_sos, fir_design  # noqa: F401
from . import data, experiment  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .classifier import (EEGNetPath, FASTHeadClassifier, FilterbankCNNClassifier,  # noqa: F401
                         FilterbankEEGNetClassifier, GradientBucket, HotPath, NotFittedError, Trainer,
                         cosine_scheduler, lr_multiplier, smoke_classifier)

__all__ = ["FusedAdamW", "FilterbankCNNClassifier", "FilterbankEEGNetClassifier", "FASTHeadClassifier", "EEGNetPath", "NotFittedError", "Trainer", "HotPath", "GradientBucket",
           "cosine_scheduler", "lr_multiplier", "extract_features", "FeatureExtractor", "Filterbank", "Stft", "band_bins", "butter_bandpass_sos",
           "butter_bandpass_resonators", "BANDS_5", "BANDS_9", "BANDS_40", "CLASSES", "ELECTRODES", "ZONES",
           "zone_index_lists", "data", "experiment", "FirFilter", "filter_data", "fir_design"]
