"""Test-mode frame/feature index logic of the reference's datasets:
`_get_start_indices` (feature_dataset.py:243-259) and the frame loop of `_get` (:359-367), vectorised
(one numpy gather instead of 512*S Python-level tensor indexings).  Bit-exact index tables are pinned by
tests/golden/tables.npz, which were produced by the reference's own code."""
from __future__ import annotations

import math

import numpy as np


def test_start_indices(num_frames: int, num_segments: int, frames_per_segment: int, stride: int = 1):
    """-> (start_indices int64 [num_segments*S], segment_size S)."""
    unit = num_segments * frames_per_segment * stride
    end_frame = math.ceil(num_frames / unit) * unit                       # round_to_nearest, feature_dataset.py:17-26
    starts = np.arange(end_frame / (frames_per_segment * stride)) * (frames_per_segment * stride)
    return starts.astype(np.int64), len(starts) // num_segments


def frame_index_table(start_indices: np.ndarray, frames_per_segment: int, stride: int, num_frames: int) -> np.ndarray:
    """flat list of source frame indices, wrapped modulo the video length (feature_dataset.py:362)."""
    off = np.arange(frames_per_segment, dtype=np.int64) * stride
    return ((start_indices[:, None].astype(np.int64) + off[None, :]) % num_frames).reshape(-1)


def gather_test_features(features: np.ndarray, num_segments: int, frames_per_segment: int, stride: int, ncrops: int = 1):
    """features [T*ncrops, D] (as stored in the .npy files) -> ([ncrops, N*S*L, D], segment_size), the tensor the
    reference's test loader yields (feature_dataset.py:347-376)."""
    D = features.shape[-1]
    f = features.reshape(-1, ncrops, D)                                   # (t, ncrops, D)
    T = f.shape[0]
    starts, S = test_start_indices(T, num_segments, frames_per_segment, stride)
    idx = frame_index_table(starts, frames_per_segment, stride, T)
    return np.ascontiguousarray(f[idx].transpose(1, 0, 2)), S
