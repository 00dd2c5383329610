"""The reference's extension points on the GPU path: a BRDF defined by the caller (dj_brdf.h:74-109) is fitted by sampling its
eval() on the host at the fit's query directions and running k_fit / launch_fit_aniso on the samples (a per-slot source in HBM).
Tables, Fresnel splines and both moment fits are compared bit for bit with tests/golden/custom.npz (the REAL reference running
the same user classes) and with the live oracle.  The C++ side of the same feature -- classes derived from djb::brdf and
djb::fresnel::impl compiled against include/dj_brdf.h -- is in tests/test_gpu_facade_conformance.py and in
test_gpu_golden.py::test_reference_programs_unchanged (examples/custom_brdf.cpp)."""
import numpy as np
import pytest

import user_defined_cases
from dj_brdf_amd import djb
from golden_cases import CUSTOM_LOBES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(CUSTOM_LOBES))
def test_user_defined_brdf_is_fitted_from_host_samples(gpu_ctx, oracle, name):
    user_defined_cases.check_user_defined_fits(gpu_ctx, oracle, name)


def test_sample_count_errors(gpu_ctx):
    user_defined_cases.check_sample_count_errors(gpu_ctx)


def test_lambert_source(gpu_ctx, oracle):
    user_defined_cases.check_lambert_source(gpu_ctx, oracle)


def test_sampled_fit_equals_resident_fit(gpu_ctx):
    """a resident BRDF sampled at the query slots and fitted from the samples gives the tables of the resident fit: the slot
    numbering of djb_fit_query_dirs is the one k_fit reads (every slot, both resolutions' skipped pairs included)"""
    g = djb.ggx(djb.fresnel.schlick((0.9, 0.6, 0.3)), True, ctx=gpu_ctx)
    for res in (90, 11):
        qi, qo = djb.fit_query_dirs(res)
        ok = ~np.isnan(qo[:, 0])
        rgb = np.zeros((qi.shape[0], 3), np.float32)
        rgb[ok] = g.eval(qi[ok], qo[ok])
        a, b = djb.tabular.from_samples(res, rgb, True, ctx=gpu_ctx), djb.tabular(g, res, True, ctx=gpu_ctx)
        for fa, fb in ((a.get_p22v(), b.get_p22v()), (a.get_sigmav(), b.get_sigmav()), (a.get_cdfv(), b.get_cdfv()),
                       (a.get_qfv(), b.get_qfv()), (a.get_fresnel().get_points(), b.get_fresnel().get_points())):
            assert user_defined_cases.same(fa, fb)
    qi, qo = djb.fit_aniso_query_dirs(10, 12)
    ok = ~np.isnan(qo[:, 0])
    rgb = np.zeros((qi.shape[0], 3), np.float32)
    rgb[ok] = g.eval(qi[ok], qo[ok])
    a = djb.tabular_anisotropic.from_samples(10, 12, rgb, True, ctx=gpu_ctx)
    b = djb.tabular_anisotropic(g, 10, 12, True, ctx=gpu_ctx)
    assert user_defined_cases.same(a.get_p22v()[0], b.get_p22v()[0]) and user_defined_cases.same(a.get_sigmav()[0], b.get_sigmav()[0])
    assert user_defined_cases.same(a.get_fresnel().get_points(), b.get_fresnel().get_points())
