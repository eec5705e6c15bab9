"""Decoding-matrix cache in the reference's on-disk format (src/utils/caching.py): same key recipe, same ``.npz`` keys, so
caches written by either side are interchangeable."""
import hashlib
import os

import numpy as np

_KEYS = ("HdecZ", "HdecX", "channel_probsZ", "channel_probsX", "HZ_full", "HX_full")
_SCALARS = ("first_logical_rowZ", "first_logical_rowX", "num_cycles", "k")


def compute_cache_key(Hx, Hz, Lx, Lz, num_cycles, error_rate):
    """sha256 over the raw bytes of the four arrays (dtype matters: the reference hashes int64 H and uint8 L as stored in
    codes/*.npz), str(num_cycles) and the rate printed with six decimals; first 16 hex digits (caching.py:6-11)."""
    h = hashlib.sha256()
    for arr in (Hx, Hz, Lx, Lz):
        h.update(np.asarray(arr).tobytes())
    h.update(str(num_cycles).encode())
    h.update(f"{error_rate:.6f}".encode())
    return h.hexdigest()[:16]


def _dense(M):
    return M.toarray() if hasattr(M, "toarray") else np.asarray(M)


def save_matrices(cache_dir, cache_key, matrices):
    """np.savez_compressed with the reference's key set (caching.py:13-26); scalars are stored as 1-element arrays."""
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"matrices_{cache_key}.npz")
    payload = {k: _dense(matrices[k]) for k in _KEYS}
    payload.update({k: np.array([matrices[k]]) for k in _SCALARS})
    np.savez_compressed(path, **payload)
    return path


def load_matrices(cache_dir, cache_key):
    """Inverse of save_matrices; returns None when the file is missing or unreadable, like the reference (caching.py:28-42)."""
    path = os.path.join(cache_dir, f"matrices_{cache_key}.npz")
    if not os.path.exists(path):
        return None
    try:
        with np.load(path) as data:          # allow_pickle stays False: plain arrays only
            out = {k: data[k] for k in _KEYS}
            out.update({k: int(data[k][0]) for k in _SCALARS})
        return out
    except Exception:
        return None
