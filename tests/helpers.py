"""Shared helpers for the parity tests: oracle mesh -> C-ABI mesh, error metrics."""
import numpy as np
import oracle_lib as O
import math_audio_amd as ma

C_SOUND = 343.0
RADIUS = 0.1


def to_ma_mesh(om):
    """oracle_lib.Mesh -> math_audio_amd.MeshArrays (same arrays the Rust shim would pass)."""
    return ma.MeshArrays(om.nodes, om.conn, om.center, om.normal, om.area, dof=om.dof, bc_type=om.bc_type,
                         bc_values=om.bc_values, bc_len=om.bc_len, is_eval=om.is_eval)


def k_from_ka(ka, radius=RADIUS, c=C_SOUND):
    """qa_suite.rs:210-212: k = ka / radius; freq = k c / 2pi; PhysicsParams::new recomputes k = 2 pi f / c."""
    k = ka / radius
    freq = k * c / (2.0 * np.pi)
    return O.wave_number(freq, c)


def rowscaled_maxerr(A, B):
    """max_ij |A_ij - B_ij| / max_j |B_ij| (per-row infinity-norm scaling)."""
    scale = np.abs(B).max(axis=1, keepdims=True)
    return float((np.abs(A - B) / scale).max())


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))
