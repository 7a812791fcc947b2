"""Paired clean/RIR test set with the preprocessing of reference ``datasets/vctk.py:148-226`` (``VCTKTestPaired``):
RIR trimmed to its absolute maximum (direct path) and peak-normalised.  Reads wavs with scipy (soundfile is absent)."""
from __future__ import annotations

import glob
import os

import numpy as np
from scipy.io import wavfile


def _read(path):
    fs, d = wavfile.read(path)
    if d.dtype.kind == "i":
        d = d.astype(np.float64) / np.iinfo(d.dtype).max
    return d.astype(np.float64), fs


class VCTKTestPaired:
    def __init__(self, fs=16000, segment_length=65536, path="", speakers_discard=(), speakers_test=(), normalize=False, seed=0,
                 num_examples=8, shuffle=True):
        if normalize:
            raise NotImplementedError("normalization not implemented yet")
        self.test_samples, self.rir_samples = [], []
        for s in os.listdir(os.path.join(path, "clean")):
            if s in speakers_discard or (len(speakers_test) and s not in speakers_test):
                continue
            new = glob.glob(os.path.join(path, "clean", s, "*.wav"))
            self.test_samples.extend(new)
            for f in new:
                self.rir_samples.append(os.path.join(path, "rir", s, os.path.splitext(os.path.basename(f))[0] + ".wav"))
        assert len(self.test_samples) >= num_examples, "error in dataloading: not enough examples"
        if num_examples > 0:
            self.test_samples, self.rir_samples = self.test_samples[:num_examples], self.rir_samples[:num_examples]
        self.fs = fs
        self.test_audio, self.test_rir, self.filenames = [], [], []
        for f, fr in zip(self.test_samples, self.rir_samples):
            data, sr = _read(f)
            rir, sr2 = _read(fr)
            assert sr == fs and sr2 == fs, "wrong sampling rate"
            assert data.ndim == 1 and rir.ndim == 1, "wrong number of channels"
            rir = rir[np.argmax(np.abs(rir)):]
            rir = rir / np.abs(rir).max()
            self.test_audio.append(data); self.test_rir.append(rir); self.filenames.append(os.path.basename(f))

    def __getitem__(self, idx):
        return self.test_audio[idx], self.test_rir[idx], self.filenames[idx]

    def __len__(self):
        return len(self.test_samples)
