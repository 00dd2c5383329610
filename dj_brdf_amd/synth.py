"""Synthetic, bit-reproducible workloads for the dj_brdf hot path.

Nothing here is reference behaviour: the reference ships no data, no RNG and no
benchmark inputs.  These generators define the inputs that
tests/ and bench.py feed to BOTH the HIP path and the CPU oracle:

* ``directions`` / ``uniforms``: counter-based hash -> float32, using only exact
  IEEE operations, so the numpy version here and the on-device generator
  (csrc/djb_gen.hip, ``djb_gen_directions`` / ``djb_gen_uniforms``) produce the
  same bits.
* ``merl_table``: a MERL-format table (3 x 90 x 90 x 180 doubles, file units,
  negative below the horizon) filled from an analytic GGX + diffuse BRDF at the
  bin centres; ``write_merl_binary`` stores it in the on-disk layout that
  ``djb::merl`` reads (reference dj_brdf.h:963-983).
"""
from __future__ import annotations

import math
import numpy as np

SEED_I = 0xD1B00001
SEED_O = 0xD1B00002
SEED_U1 = 0xD1B00003
SEED_U2 = 0xD1B00004

MERL_N = 90 * 90 * 180
MERL_SCALE = (1.00 / 1500.0, 1.15 / 1500.0, 1.66 / 1500.0)  # reference dj_brdf.h:897-899
MERL_FILE_BYTES = 12 + 8 * 3 * MERL_N


def _pcg(x: np.ndarray) -> np.ndarray:
    """PCG-RXS-M-XS 32-bit output hash (uint32 wraparound arithmetic)."""
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        state = x * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    return (word >> np.uint32(22)) ^ word


def hash_u32(seed: int, k: np.ndarray, c: int) -> np.ndarray:
    """h(seed, c, k): three PCG rounds over (seed + c*golden, lo32(k), hi32(k))."""
    k = np.asarray(k, dtype=np.uint64)
    lo = (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (k >> np.uint64(32)).astype(np.uint32)
    s = np.uint32((seed + c * 0x9E3779B9) & 0xFFFFFFFF)
    h = _pcg(np.full(k.shape, s, dtype=np.uint32))
    h = _pcg(h ^ lo)
    with np.errstate(over="ignore"):
        h = _pcg(h + hi)
    return h


def uniforms(n: int, seed: int, start: int = 0) -> np.ndarray:
    """n float32 uniforms in [0, 1): (h >> 8) * 2^-24."""
    k = np.arange(start, start + n, dtype=np.uint64)
    h = hash_u32(seed, k, 0)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def rng_uniforms(n: int, seed: int, start: int = 0) -> np.ndarray:
    """The uniforms the library generates ON CHIP (sample_rng, evalp_is_rng, djb_gen_uniforms): gen_uniform of
    csrc/djb_device_units.inc restated -- one lowbias32 finaliser over lo32(k) * 0x9E3779B9 + hi32(k) * 0x85EBCA6B + pcg(seed), top 24 bits.
    (uniforms() above, three chained PCG rounds, stays the generator of test inputs and golden vectors.)"""
    k = np.arange(start, start + n, dtype=np.uint64)
    lo = (k & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (k >> np.uint64(32)).astype(np.uint32)
    key = _pcg(np.full(1, seed & 0xFFFFFFFF, dtype=np.uint32))[0]
    with np.errstate(over="ignore"):
        x = lo * np.uint32(0x9E3779B9) + hi * np.uint32(0x85EBCA6B) + key
        x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def directions(n: int, seed: int, start: int = 0):
    """n unit vectors in the upper hemisphere as three float32 arrays (SoA).

    x, y = 2*(h>>8)*2^-24 - 1; if x^2+y^2 >= 0.998 both are halved (exact);
    z = sqrtf(1 - x^2 - y^2) with every product/sum rounded to float (no FMA).
    """
    k = np.arange(start, start + n, dtype=np.uint64)
    f = np.float32
    x = (hash_u32(seed, k, 0) >> np.uint32(8)).astype(f) * f(2.0 ** -23) - f(1.0)
    y = (hash_u32(seed, k, 1) >> np.uint32(8)).astype(f) * f(2.0 ** -23) - f(1.0)
    r2 = x * x + y * y
    big = r2 >= f(0.998)
    x = np.where(big, x * f(0.5), x)
    y = np.where(big, y * f(0.5), y)
    z = np.sqrt((f(1.0) - x * x) - y * y, dtype=f)
    return x, y, z


def directions_aos(n: int, seed: int, start: int = 0) -> np.ndarray:
    return np.stack(directions(n, seed, start), axis=1).astype(np.float32)


# --------------------------------------------------------------------------- MERL tables
def _trig_deg(deg):
    """(cos, sin) of angles in degrees via libm's scalar double routines (math.*), NOT numpy's
    SIMD loops, so the table is bit-reproducible wherever the same glibc is installed."""
    rad = [math.radians(float(d)) for d in deg]
    return (np.array([math.cos(r) for r in rad], dtype=np.float64),
            np.array([math.sin(r) for r in rad], dtype=np.float64))


def _bin_centre_io():
    """(i, o, h) unit vectors (float64, shape [90,90,180,3]) at the MERL bin centres."""
    ih = np.arange(90, dtype=np.float64) + 0.5
    ct, st = _trig_deg(ih * ih / 90.0)
    ctd, std = _trig_deg(np.arange(90, dtype=np.float64) + 0.5)
    cpd, spd = _trig_deg(np.arange(180, dtype=np.float64) + 0.5)
    ct, st = ct[:, None, None], st[:, None, None]
    ctd, std = ctd[None, :, None], std[None, :, None]
    cpd, spd = cpd[None, None, :], spd[None, None, :]
    # d in the half-vector frame, then rotate by theta_h about y (phi_h = 0)
    dx, dy, dz = std * cpd, std * spd, ctd + 0 * cpd
    ix, iy, iz = ct * dx + st * dz, dy + 0 * ct, -st * dx + ct * dz
    hx, hy, hz = st + 0 * dx, 0 * dx + 0 * ct, ct + 0 * dx
    idh = ix * hx + iy * hy + iz * hz
    ox, oy, oz = 2 * idh * hx - ix, 2 * idh * hy - iy, 2 * idh * hz - iz
    return np.stack([ix, iy, iz], -1), np.stack([ox, oy, oz], -1), np.stack([hx, hy, hz], -1)


def merl_table(alpha: float = 0.3, diffuse=(0.10, 0.08, 0.05), f0=(0.9, 0.7, 0.4)) -> np.ndarray:
    """Synthetic MERL table, shape [3, 90, 90, 180] float64, in FILE units.

    GGX isotropic(alpha) with a separable-style Smith G, Schlick Fresnel(f0) plus diffuse/pi,
    evaluated at bin centres and divided by the per-channel MERL scale; entries whose i or o is
    below the horizon are -1 (real MERL files carry negative values there and djb::merl::eval
    returns 0 for them).  Only + - * / sqrt on doubles after the bin-centre trig, so the bits do
    not depend on numpy's vector-math dispatch.
    """
    i, o, h = _bin_centre_io()
    iz, oz, hz = i[..., 2], o[..., 2], h[..., 2]
    valid = (iz > 1e-6) & (oz > 1e-6)
    izs, ozs = np.where(valid, iz, 1.0), np.where(valid, oz, 1.0)
    a2 = alpha * alpha
    den = hz * hz * (a2 - 1.0) + 1.0
    D = a2 / (math.pi * den * den)
    lam = lambda c: 0.5 * (-1.0 + np.sqrt(1.0 + a2 * (1.0 - c * c) / (c * c)))
    G = 1.0 / (1.0 + lam(izs) + lam(ozs))
    cd = np.minimum(np.maximum((o * h).sum(-1), 0.0), 1.0)
    omc = 1.0 - cd
    omc5 = omc * omc * omc * omc * omc
    out = np.empty((3,) + iz.shape, dtype=np.float64)
    for c in range(3):
        F = f0[c] + (1.0 - f0[c]) * omc5
        fr = F * D * G / (4.0 * izs * ozs) + diffuse[c] / math.pi
        out[c] = np.where(valid, fr / MERL_SCALE[c], -1.0)
    return out


def merl_table_hashed(seed: int = 7, negative_every: int = 97) -> np.ndarray:
    """Index-revealing table built from integer hashing only (exact on any machine).

    value[c, idx] = (hash(seed, idx, c) >> 8) / 2^12  (a multiple of 2^-12 below 4096),
    and every ``negative_every``-th red entry is negative (below-horizon marker).
    """
    idx = np.arange(MERL_N, dtype=np.uint64)
    out = np.empty((3, MERL_N), dtype=np.float64)
    for c in range(3):
        out[c] = (hash_u32(seed, idx, c) >> np.uint32(8)).astype(np.float64) / 4096.0
    out[0, ::negative_every] = -out[0, ::negative_every] - 1.0
    return out.reshape(3, 90, 90, 180)


def merl_table_grazing(power: int = 30, lo: float = 1e-3, hi: float = 50.0) -> np.ndarray:
    """A MERL-format table whose value depends on the theta_h bin only and rises steeply towards grazing
    half-angles: lo + hi * (ih / 89)^power (exact integer powers of doubles, no libm).  Fitted as a
    tabular_anisotropic it puts so much slope-pdf mass at the horizon that the conditional CDFs cannot be
    inverted for the upper quantiles -- the case in which the reference's m_qf2 comes up short
    (dj_brdf.h:3005-3034)."""
    x = np.arange(90, dtype=np.float64) / 89.0
    g = np.ones(90, dtype=np.float64)
    for _ in range(int(power)):
        g = g * x
    g = lo + g * hi
    t = np.empty((3, 90, 90, 180), dtype=np.float64)
    t[:] = g[None, :, None, None]
    return t


UTIA_N = 3 * 288 * 288


def utia_table_smooth() -> np.ndarray:
    """A UTIA-format payload (3 x 288 x 288 doubles, sRGB-coded * 140; dj_brdf.h:1039-1059) of a smooth,
    azimuthally ANISOTROPIC glossy material: value(theta_i, phi_i, theta_v, phi_v) from rational functions of
    the grid indices only (+ - * / on doubles: the same bits on any machine, unlike numpy's SIMD trig).
    idx = isp*288*288 + 288*(48*iti + ipi) + 48*itv + ipv."""
    iti = np.arange(6, dtype=np.float64)[:, None, None, None]
    ipi = np.arange(48, dtype=np.float64)[None, :, None, None]
    itv = np.arange(6, dtype=np.float64)[None, None, :, None]
    ipv = np.arange(48, dtype=np.float64)[None, None, None, :]
    k = np.abs(ipi - ipv)
    d = np.minimum(k, 48.0 - k) / 24.0                      # azimuth difference in [0, 1]; 0 = same azimuth
    brush = ((ipi % 24.0) - 12.0) * ((ipi % 24.0) - 12.0) / 144.0       # period 180 degrees in phi_i
    dt = (iti - itv) / 6.0
    # lobe around the BACK-scattering configuration (what compute_p22_smith probes), wider across the brush direction
    spec = 1.0 / (1.0 + 30.0 * dt * dt + 25.0 * d * d * (0.25 + brush))
    tilt = 1.0 / (1.0 + 0.35 * (iti + itv))
    out = np.empty((3, 6, 48, 6, 48), dtype=np.float64)
    for c, (kd, ks) in enumerate(((0.30, 0.55), (0.22, 0.60), (0.12, 0.65))):
        out[c] = 140.0 * (kd * 0.5 + ks * spec * tilt)
    out[0, 0, 0, 0, 0] = -3.0                               # one negative sample: clamped by utia::normalize
    return out.reshape(-1)


def write_merl_binary(path: str, table: np.ndarray) -> None:
    """int32 dims[3] = (90, 90, 180) + 3*n doubles, plane order R, G, B (dj_brdf.h:963-983)."""
    tab = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
    assert tab.size == 3 * MERL_N
    with open(path, "wb") as f:
        np.array([90, 90, 180], dtype=np.int32).tofile(f)
        tab.tofile(f)


# The 100 MERL material names (the public MERL-100 list; also the keys of the
# reference's sgd/abc parameter tables, dj_brdf.h:3313-3412).
MERL_NAMES = [
    "alum-bronze", "alumina-oxide", "aluminium", "aventurnine", "beige-fabric",
    "black-fabric", "black-obsidian", "black-oxidized-steel", "black-phenolic",
    "black-soft-plastic", "blue-acrylic", "blue-fabric", "blue-metallic-paint",
    "blue-metallic-paint2", "blue-rubber", "brass", "cherry-235", "chrome",
    "chrome-steel", "colonial-maple-223", "color-changing-paint1",
    "color-changing-paint2", "color-changing-paint3", "dark-blue-paint",
    "dark-red-paint", "dark-specular-fabric", "delrin", "fruitwood-241",
    "gold-metallic-paint", "gold-metallic-paint2", "gold-metallic-paint3",
    "gold-paint", "gray-plastic", "grease-covered-steel", "green-acrylic",
    "green-fabric", "green-latex", "green-metallic-paint", "green-metallic-paint2",
    "green-plastic", "hematite", "ipswich-pine-221", "light-brown-fabric",
    "light-red-paint", "maroon-plastic", "natural-209", "neoprene-rubber", "nickel",
    "nylon", "orange-paint", "pearl-paint", "pickled-oak-260", "pink-fabric",
    "pink-fabric2", "pink-felt", "pink-jasper", "pink-plastic", "polyethylene",
    "polyurethane-foam", "pure-rubber", "purple-paint", "pvc", "red-fabric",
    "red-fabric2", "red-metallic-paint", "red-phenolic", "red-plastic",
    "red-specular-plastic", "silicon-nitrade", "silver-metallic-paint",
    "silver-metallic-paint2", "silver-paint", "special-walnut-224",
    "specular-black-phenolic", "specular-blue-phenolic", "specular-green-phenolic",
    "specular-maroon-phenolic", "specular-orange-phenolic", "specular-red-phenolic",
    "specular-violet-phenolic", "specular-white-phenolic", "specular-yellow-phenolic",
    "ss440", "steel", "teflon", "tungsten-carbide", "two-layer-gold",
    "two-layer-silver", "violet-acrylic", "violet-rubber", "white-acrylic",
    "white-diffuse-bball", "white-fabric", "white-fabric2", "white-marble",
    "white-paint", "yellow-matte-plastic", "yellow-paint", "yellow-phenolic",
    "yellow-plastic",
]


def material_recipe(index: int):
    """Deterministic (alpha, diffuse, f0) for the index-th synthetic MERL material."""
    h = [int(hash_u32(0xABCD0000 + index, np.array([j], dtype=np.uint64), 0)[0]) for j in range(7)]
    u = [(v >> 8) / float(1 << 24) for v in h]
    alpha = 0.03 + 0.55 * u[0]
    diffuse = tuple(0.02 + 0.5 * u[1 + c] * (1.0 - 0.5 * u[0]) for c in range(3))
    f0 = tuple(0.04 + 0.9 * u[4 + c] for c in range(3))
    return alpha, diffuse, f0
