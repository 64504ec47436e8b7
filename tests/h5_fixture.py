"""Synthetic LOFAR-extract-shaped SAP (the H5 layout read by src/lofar_tools.py:76-109) shared by the golden
generator (fed to the REFERENCE loader through a stand-in h5py module) and the GPU parity test."""
import numpy as np


def make_sap(nbase=6, ntime=200, nfreq=260, seed_phase=0.3):
    i = np.arange(nbase * ntime * nfreq * 4 * 2, dtype=np.float64)
    vis = np.round(90.0 * np.sin(0.0173 * i + seed_phase) + 30.0 * np.cos(1.31 * i)).astype(np.int8)
    vis = vis.reshape(nbase, ntime, nfreq, 4, 2)
    j = np.arange(nbase * nfreq * 4, dtype=np.float64)
    scale = (0.02 + 0.015 * np.sin(0.37 * j + 1.0) ** 2).astype(np.float32).reshape(nbase, nfreq, 4)
    scale[1, 5, 0] = 40.0  # makes the +-1e3 clamp bite on one channel/frequency
    freqs = (120e6 + 195312.5 * np.arange(nfreq)).astype(np.float64)
    nst = 5
    xyz = np.stack([3826000.0 + 700.0 * np.sin(1.3 * np.arange(nst)), 461000.0 + 900.0 * np.cos(0.7 * np.arange(nst)),
                    5064000.0 + 10.0 * np.arange(nst)], axis=1)
    baselines = np.array([(a, b) for a in range(nst) for b in range(a + 1, nst)][:nbase], dtype=np.int64)
    sap = {"visibilities": vis, "visibility_scale_factors": scale, "central_frequencies": freqs,
           "baselines": baselines, "antenna_locations": {"XYZ": xyz}}
    info = {"start_time": [b"2019-03-14 17:42:31.5"]}
    return sap, info


def drawn_baselines(seed, nfiles, nbase, batch_size):
    """The two np.random.randint draws of get_data_minibatch (:71, :88) for a given seed."""
    np.random.seed(seed)
    np.random.randint(0, nfiles)
    return np.random.randint(0, nbase, batch_size)
