"""Oracle: dense little-endian code packing (TEST INFRASTRUCTURE -- see oracle/__init__.py).
Build-defined wire format for SURVEY 8(f4); the reference only states the bit budget
(``networks/utils.py:137-147``: bits per frame = num_quantizers * log2(codebook_size))."""
import numpy as np


def pack(codes: np.ndarray, bits: int) -> np.ndarray:
    codes = np.asarray(codes, dtype=np.uint64).reshape(-1)
    shifts = np.arange(bits, dtype=np.uint64)
    bitmat = ((codes[:, None] >> shifts[None, :]) & np.uint64(1)).astype(np.uint8)   # LSB first
    return np.packbits(bitmat.reshape(-1), bitorder="little")


def unpack(stream: np.ndarray, n_codes: int, bits: int) -> np.ndarray:
    flat = np.unpackbits(np.asarray(stream, dtype=np.uint8), bitorder="little")[: n_codes * bits]
    weights = (np.uint64(1) << np.arange(bits, dtype=np.uint64))
    return (flat.reshape(n_codes, bits).astype(np.uint64) * weights).sum(axis=1).astype(np.int64)
