"""mental-poker_amd: MI355X-native shuffle-proof engine behind the `BarnettSmartProtocol` hot path of
geometryxyz/mental-poker (shuffle_and_remask / verify_shuffle).  The package directory name is not a Python
identifier: import it with `importlib.import_module("mental-poker_amd")`.

Everything is computed by libmpshuffle.so (hand-written HIP for gfx950); importing works without a GPU,
creating an engine does not (no CPU fallback)."""
from . import _native, canonical
from ._native import Engine, NativeError, NoDeviceError, Serializer, build, load
from .protocol import (CardProtocolError, ChaCha20Rng, CryptoError, DLCards, Parameters, Permutation, fr_rand)

__all__ = ["Engine", "Serializer", "NativeError", "NoDeviceError", "build", "load", "DLCards", "Parameters", "Permutation",
           "CryptoError", "CardProtocolError", "ChaCha20Rng", "fr_rand", "_native", "canonical"]
