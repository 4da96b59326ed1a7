"""TEST INFRASTRUCTURE (oracle side) -- deterministic counter-based synthetic data.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product generates the same values on the device (csrc/aux_kernels.hpp::synth_fill_kernel); a GPU
test checks that both generators agree bit for bit, so host and device can each
regenerate the BASELINE.json workloads without moving the
index over PCIe.

value(seed, row, col) is a pure function, so any sub-block can be produced
independently (row-sharded ranks generate only their own rows).

Two kinds:
  LATTICE  integers in [-127, 127] / 64 : exact in bf16, every product and every
           768-term sum exact in fp32 -> scores are order independent, ties occur,
           GPU and CPU must agree bit-exactly on scores AND indices.
  GAUSS    approximately N(0,1), rounded to bf16 (RNE).  Built from a sum of eight
           16-bit uniforms (Irwin-Hall) with integer arithmetic only, NOT
           Box-Muller: logf/cosf differ between host libm and the GPU, which would
           make the two generators disagree in the last bf16 bit.
"""
from __future__ import annotations

import numpy as np

KIND_LATTICE = 0
KIND_GAUSS = 1
KIND_LATTICE_FP8 = 2  # integers in [-8, 8] / 8 : exact in fp8 e4m3 as well

SEED_DOCS = 0xD0C5
SEED_QUERIES = 0x0E21

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ROWMUL = np.uint64(0xD6E8FEB86659FD93)

# 1 / (65536 * sqrt(8/12)): scales the centred sum of eight u16 uniforms to unit variance
GAUSS_SCALE = np.float32(1.0 / (65536.0 * 0.816496580927726))


def _mix(z: np.ndarray) -> np.ndarray:
    """SplitMix64 step (increment + finaliser) on uint64 arrays, wrapping arithmetic."""
    with np.errstate(over="ignore"):
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _row_keys(seed: int, rows: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return _mix(np.uint64(seed) ^ (rows.astype(np.uint64) * _ROWMUL))


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 -> nearest-even bf16, returned as float32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 values that are exactly bf16-representable -> uint16 bit patterns."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def generate(seed: int, row0: int, nrows: int, d: int, kind: int) -> np.ndarray:
    """float32 [nrows, d]; every value exactly representable in bf16."""
    rows = np.arange(row0, row0 + nrows, dtype=np.uint64)
    key = _row_keys(seed, rows)[:, None]                      # [n,1]
    cols = np.arange(d, dtype=np.uint64)[None, :]             # [1,d]
    with np.errstate(over="ignore"):
        if kind == KIND_LATTICE:
            h = _mix(key + cols)
            v = ((h >> np.uint64(40)) % np.uint64(255)).astype(np.int64) - 127
            return (v.astype(np.float32) / np.float32(64.0)).astype(np.float32)
        if kind == KIND_LATTICE_FP8:
            h = _mix(key + cols)
            v = ((h >> np.uint64(40)) % np.uint64(17)).astype(np.int64) - 8
            return (v.astype(np.float32) / np.float32(8.0)).astype(np.float32)
        if kind == KIND_GAUSS:
            h1 = _mix(key + np.uint64(2) * cols)
            h2 = _mix(key + np.uint64(2) * cols + np.uint64(1))
            s = np.zeros(h1.shape, dtype=np.int64)
            for h in (h1, h2):
                for sh in (0, 16, 32, 48):
                    s += ((h >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
            s -= 4 * 65535
            x = s.astype(np.float32) * GAUSS_SCALE
            return round_to_bf16(x)
    raise ValueError(f"unknown kind {kind}")


def generate_blocked(seed: int, row0: int, nrows: int, d: int, kind: int, block: int = 65536):
    """Yield (start_row, float32 block) pairs; bounds peak memory for large N."""
    r = row0
    end = row0 + nrows
    while r < end:
        n = min(block, end - r)
        yield r, generate(seed, r, n, d, kind)
        r += n


# --------------------------------------------------------------------------
# OCP fp8 e4m3 ("e4m3fn": bias 7, 3 mantissa bits, max 448, 0x7f = NaN, no inf) -- the same integer
# algorithm as csrc/aux_kernels.hpp::f32_to_e4m3 (round to nearest even, saturate, NaN stays NaN).
# --------------------------------------------------------------------------
def e4m3_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> uint8 e4m3 codes."""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    sign = ((u >> np.uint32(24)) & np.uint32(0x80)).astype(np.uint8)
    a = u & np.uint32(0x7FFFFFFF)
    e = (a >> np.uint32(23)).astype(np.int64) - 127
    m3 = ((a >> np.uint32(20)) & np.uint32(7)).astype(np.int64)
    rem = (a & np.uint32(0xFFFFF)).astype(np.int64)
    up = (rem > 0x80000) | ((rem == 0x80000) & ((m3 & 1) == 1))
    m3 = m3 + up
    ee = e + 7 + (m3 == 8)
    m3 = np.where(m3 == 8, 0, m3)
    normal = ((ee << 3) | m3).astype(np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        sub = np.rint(np.abs(f).astype(np.float32) * np.float32(512.0))
    sub = np.where(np.isfinite(sub), sub, 0).astype(np.int64)
    code = np.where(e < -6, sub, normal)
    code = np.where(a >= np.uint32(0x43E00000), 0x7E, code)
    code = np.where(a > np.uint32(0x7F800000), 0x7F, code)
    return (sign | code.astype(np.uint8)).astype(np.uint8)


def e4m3_bits_to_f32(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, dtype=np.uint8).astype(np.uint32)
    e = (b >> np.uint32(3)) & np.uint32(15)
    m = b & np.uint32(7)
    sub = m.astype(np.float32) * np.float32(1.0 / 512.0)
    norm = (((e + np.uint32(120)) << np.uint32(23)) | (m << np.uint32(20))).astype(np.uint32).view(np.float32)
    v = np.where(e == 0, sub, norm).astype(np.float32)
    v = np.where((e == 15) & (m == 7), np.float32(np.nan), v)
    return np.where((b & np.uint32(0x80)) != 0, -v, v).astype(np.float32)


def round_to_e4m3(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest e4m3 value (saturating), returned as float32."""
    return e4m3_bits_to_f32(e4m3_bits(x))
