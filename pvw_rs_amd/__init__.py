"""pvw_rs_amd -- MI355X-native (gfx950) PVW multi-receiver encrypt/decrypt hot path.

`csrc/` holds the hand-written HIP kernels and the C ABI (include/pvw_hip.h);
`api.py` is the host-side mirror of the reference's `pvw::{params,crs,keys,crypto}`
interface over that ABI; `host/pvw.hpp` is the same mirror in C++.
There is no CPU fallback: device entry points fail loudly without the library / a GPU.
"""
from ._ffi import (DOM_CRS, DOM_E1, DOM_E2, DOM_EKEY, DOM_GAUSS, DOM_PK, DOM_R, DOM_SK, PREPARE_MFMA, PREPARE_PACKED,
                   REPR_NTT, REPR_POWER)
from .api import (DeviceSecretKey, GlobalPublicKey, Party, PvwCiphertext, PvwCrs, PvwError, PvwParameters,
                  PvwParametersBuilder, SecretKey, decode_scalar_pvw, decode_scalar_pvw_host, decrypt_party_shares,
                  decrypt_party_value, device_available, encrypt, encrypt_all_party_shares,
                  encrypt_broadcast, encrypt_many, encrypt_party_shares)

__all__ = [
    "PvwParametersBuilder", "PvwParameters", "PvwCrs", "SecretKey", "DeviceSecretKey", "Party", "GlobalPublicKey",
    "PvwCiphertext", "PvwError", "encrypt", "encrypt_party_shares", "encrypt_all_party_shares",
    "encrypt_broadcast", "encrypt_many", "decrypt_party_value", "decrypt_party_shares", "decode_scalar_pvw",
    "device_available", "REPR_POWER", "REPR_NTT",
]
