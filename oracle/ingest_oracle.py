"""numpy restatements of the ingest / LS-baseline helpers (TEST INFRASTRUCTURE, see aft_oracle.c).

Pinned by tests/golden/I_ingest.npz, produced by running the reference's own MatDataset,
extract_values and get_ls_mse_per_folder on synthetic .mat files (tests/golden/make_golden.py)."""
import re

import numpy as np


def extract_values(file_name):
    """reference src/utils.py:68-110"""
    m = re.match(r"(\d+)_SNR-(\d+)_DS-(\d+)_DOP-(\d+)_N-(\d+)_([A-Z\-]+)\.mat", file_name)
    if not m:
        raise ValueError("Cannot extract file information.")
    return tuple(float(int(m.group(i))) for i in range(1, 6)) + (m.group(6),)


def process_channel_data(H, pilot_size):
    """reference src/data/dataset.py:95-144: (pilots [Ps,Pt], h_ideal [S,T]) from H [S,T,>=2]."""
    h_ideal = H[:, :, 0].astype(np.complex64)
    hzero = H[:, :, 1].astype(np.complex64)
    hp = hzero[hzero != 0]                      # boolean mask = row-major order
    if hp.size != pilot_size[0] * pilot_size[1]:
        raise ValueError(f"Expected {pilot_size[0] * pilot_size[1]} pilot values, got {hp.size}")
    return hp.reshape(pilot_size), h_ideal


def mse_db(x, y):
    """reference src/utils.py:248-261"""
    return 10 * np.log10(np.mean(np.square(np.abs(x - y))))
