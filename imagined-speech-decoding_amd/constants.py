"""Dataset constants the hot path needs as data (label order, montage, zones).

Values restate src/fast/data/preprocess.py:20 (CLASSES), :24-30 (Electrodes),
:33-42 (Zones) of the reference; dict order is zone order (fast.py:205-207).
"""
CLASSES = ["hello", "help-me", "stop", "thank-you", "yes"]

ELECTRODES = [
    "Fp1", "Fp2", "F7", "F3", "Fz", "F4", "F8", "FC5", "FC1", "FC2", "FC6", "T7", "C3", "Cz", "C4",
    "T8", "TP9", "CP5", "CP1", "CP2", "CP6", "TP10", "P7", "P3", "Pz", "P4", "P8", "PO9", "O1", "Oz",
    "O2", "PO10", "AF7", "AF3", "AF4", "AF8", "F5", "F1", "F2", "F6", "FT9", "FT7", "FC3", "FC4", "FT8",
    "FT10", "C5", "C1", "C2", "C6", "TP7", "CP3", "CPz", "CP4", "TP8", "P5", "P1", "P2", "P6", "PO7",
    "PO3", "POz", "PO4", "PO8",
]

ZONES = {
    "Pre-frontal": ["AF7", "Fp1", "Fp2", "AF8", "AF3", "AF4"],
    "Frontal": ["F7", "F5", "F3", "F1", "Fz", "F2", "F4", "F6", "F8"],
    "Pre-central": ["FC1", "FC2", "FC3", "FC4", "FC5", "FC6"],
    "Central": ["C1", "C2", "C3", "Cz", "C4", "C5", "C6"],
    "Post-central": ["CP1", "CP2", "CP3", "CPz", "CP4", "CP5", "CP6"],
    "Temporal": ["T7", "T8", "FT7", "FT8", "TP7", "TP8", "TP9", "TP10", "FT9", "FT10"],
    "Parietal": ["P1", "P2", "P3", "P4", "Pz", "P5", "P6", "P7", "P8", "PO3", "PO4", "PO7", "PO8",
                 "PO9", "PO10"],
    "Occipital": ["O1", "O2", "Oz", "POz"],
}

# scripts/global_shap_analysis.py:138-144
BANDS_5 = (("Delta", 0.5, 4.0), ("Theta", 4.0, 8.0), ("Alpha", 8.0, 13.0), ("Beta", 13.0, 30.0),
           ("Gamma", 30.0, 100.0))
# 4 Hz tiles of the notebook's 4-40 Hz pass band (notebooks/svm_baseline.ipynb:238)
BANDS_9 = tuple((f"B{i}", 4.0 + 4.0 * i, 8.0 + 4.0 * i) for i in range(9))
# stress configuration: 2 Hz bands, 4-84 Hz
BANDS_40 = tuple((f"N{i}", 4.0 + 2.0 * i, 6.0 + 2.0 * i) for i in range(40))


def zone_index_lists(electrodes=None, zones=None):
    """Per-zone channel indices, as ``Head.__init__`` builds them (fast.py:206)."""
    electrodes = ELECTRODES if electrodes is None else list(electrodes)
    zones = ZONES if zones is None else zones
    return [[electrodes.index(ch) for ch in names] for names in zones.values()]


def band_edges(bands):
    return [(float(b[-2]), float(b[-1])) for b in bands]
