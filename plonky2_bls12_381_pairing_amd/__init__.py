"""MI355X-native batched BLS12-381 pairing engine: host-side mirror of the reference's `native`
module (plonky2-bls12-381-pairing) over hand-written gfx950 kernels behind a C ABI."""
from ._lib import Bls381Error, LIB_PATH  # noqa: F401
from .native import Context, MultiContext, witness_kinds, witness_rows  # noqa: F401
