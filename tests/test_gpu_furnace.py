"""White-furnace check on UTIA data: the batch form of the reference's tests/nrm_utia.cpp:20-51
(albedo(o) = sum_i evalp(i, o) sin(theta_i) dtheta dphi <= 1 for every o), on a synthetic
UTIA-format file (the real dataset is not available offline), plus the same quadrature from the CPU
oracle's evalp values as the parity check."""
import numpy as np
import pytest

from dj_brdf_amd import djb

pytestmark = pytest.mark.gpu

NTHETA, NPHI = 64, 256


def _dir(theta, phi):
    st = np.sin(theta)
    return np.stack([st * np.cos(phi), st * np.sin(phi), np.cos(theta)], -1).astype(np.float32)


def test_utia_white_furnace(gpu_ctx, oracle, tmp_path):
    rng = np.random.default_rng(5)
    # stored samples are sRGB-encoded reflectance-like values scaled by 140 (utia::normalize, dj_brdf.h:1162-1177);
    # keep them small enough that the decoded BRDF has albedo < 1
    tab = rng.uniform(0.0, 0.45, size=3 * 288 * 288)
    p = str(tmp_path / "furnace.bin"); tab.tofile(p)
    u, ou = djb.utia(p, ctx=gpu_ctx), oracle.utia(p)
    dtheta, dphi = (np.pi / 2) / NTHETA, (2 * np.pi) / NPHI
    # the reference's inner grid: theta_j = j/ntheta * pi/2, phi_j = j2/nphi * 2pi for j2 < ntheta (sic, nrm_utia.cpp:40)
    tj = (np.arange(NTHETA) / NTHETA * np.pi / 2).astype(np.float32)
    pj = (np.arange(NTHETA) / NPHI * 2 * np.pi).astype(np.float32)
    TI, PI = np.meshgrid(tj, pj, indexing="ij")
    i_dirs = _dir(TI.ravel().astype(np.float64), PI.ravel().astype(np.float64))
    w = np.sin(TI.ravel().astype(np.float64))
    # outer grid: every 8th theta, every 32nd phi (64 outgoing directions)
    to = (np.arange(0, NTHETA, 8) / NTHETA * np.pi / 2)
    po = (np.arange(0, NPHI, 32) / NPHI * 2 * np.pi)
    TO, PO = np.meshgrid(to, po, indexing="ij")
    o_dirs = _dir(TO.ravel(), PO.ravel())
    n_o, n_i = o_dirs.shape[0], i_dirs.shape[0]
    I = np.tile(i_dirs, (n_o, 1)); O = np.repeat(o_dirs, n_i, axis=0)
    got = u.evalp(I, O).astype(np.float64).reshape(n_o, n_i, 3)
    want = oracle.eval(ou, I, O, None, "evalp").astype(np.float64).reshape(n_o, n_i, 3)
    alb_got = (got * w[None, :, None]).sum(1) * dtheta * dphi
    alb_want = (want * w[None, :, None]).sum(1) * dtheta * dphi
    assert (alb_got <= 1.0).all() and (alb_got > 0.0).any(), "white furnace violated"
    np.testing.assert_allclose(alb_got, alb_want, rtol=1e-5, atol=1e-9)   # fp32 evalp values, fp64 quadrature
