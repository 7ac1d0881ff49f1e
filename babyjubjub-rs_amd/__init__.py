"""babyjubjub-rs_amd: MI355X-native batched BabyJubJub scalar multiplication,
Poseidon(t=6) and EdDSA-Poseidon verification -- the hot path of the Rust crate
arnaucube/babyjubjub-rs, behind a C ABI (include/bjj_hip.h).

This Python package is only the thin host-side mirror used by the tests and the
benchmark (import name: babyjubjub_rs_amd, see ../babyjubjub_rs_amd.py).  The
product is csrc/libbjj_hip.so.
"""
from .api import (  # noqa: F401
    Q, B8, SUBORDER, BjjError, Context, MultiContext, Point, PointProjective, Signature, default_context,
    mul_scalar_batch, mul_fixed_base_batch, poseidon5_batch, verify_batch, verify, point_add_batch,
    decompress_point, decompress_signature, PrivateKey, verify_schnorr, new_key,
)
from ._lib import LIB_PATH, EXPORTED_SYMBOLS, BJJ_WINDOW_AUTO as WINDOW_AUTO  # noqa: F401
