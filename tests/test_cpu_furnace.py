"""Known answers that come from physics, not from any implementation: a convex body inside a constant
environment of radiance 1 (host scene family "furnace").  A Lambertian surface must show its albedo, a mirror
its reflectance, a dielectric that absorbs nothing exactly 1 - whatever the integrator, the sampling
strategy or the MIS weights.  The oracle is held to these here (the reference ships no fixtures: this is the
one check of the restatement that does not go through the restatement); tests/test_gpu_furnace.py holds the
HIP path to the same numbers."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

W, H, SEED = 48, 36, 5
# variant -> (expected radiance on the body, integrators it holds for, relative tolerance of the mean)
CASES = {
    "matte_sphere": (0, 0.5, ("direct", "path", "pathsampled", "volpath"), 0.01),
    "matte_box": (4, 0.5, ("direct", "path", "volpath"), 0.01),
    "mirror_sphere": (1, 0.9, ("path", "pathsampled", "volpath"), 0.01),
    # specular glass: Fresnel reflection + transmission add up to 1 at every interface, nothing is absorbed
    "glass_sphere": (2, 1.0, ("path", "pathsampled", "volpath"), 0.02),
    "matte_mesh": (6, 0.5, ("direct", "path", "pathsampled", "volpath"), 0.01),
    # a ball of medium that only scatters (sigma_a = 0, g = 0.3): every photon that goes in comes out
    "scattering_ball": (5, 1.0, ("volpath",), 0.02),
    # Disney diffuse + retro-reflection at roughness 0.5 is built to keep the Lambertian albedo (Burley 2012, section 5.3):
    # the mean over the sphere is the base colour's 0.5 to within 2 % (tests/pins.py holds the lobes to the published formulas)
    "disney_sphere": (8, 0.5, ("direct", "path", "pathsampled", "volpath"), 0.025),
    # a floor under a sky that is 1 above the horizon and 0 below: irradiance pi, through the Distribution2D sampling
    "floor_half_sky": (7, 0.5, ("direct", "path", "pathsampled", "volpath"), 0.01),
}
INTEGRATORS = {"direct": pa.LI_DIRECT, "path": pa.LI_PATH, "pathsampled": pa.LI_PATHSAMPLED, "volpath": pa.LI_VOLPATH}


def body_mask(variant):
    """pixels whose centre ray hits the body with margin (the silhouette row mixes body and background)"""
    sc = pa.BuiltScene("furnace", W, H, 0, variant)
    o = orc.OracleScene(sc.flat, sc.rand_calls)
    oo, od, _ = o.generate_rays(sc.camera, W, H, SEED, 0)
    hit = o.trace_closest(oo, od, np.full(W * H, np.finfo(np.float32).max, np.float32))["prim"].reshape(H, W) >= 0
    inner = hit.copy()
    inner[1:] &= hit[:-1]; inner[:-1] &= hit[1:]; inner[:, 1:] &= hit[:, :-1]; inner[:, :-1] &= hit[:, 1:]
    o.close()
    return sc, inner


def check(img, inner, want, tol, what):
    assert inner.sum() > 100
    body = img[inner]
    assert np.isfinite(img).all(), what
    mean = body.mean(axis=0)
    assert np.all(np.abs(mean - want) <= tol * want), (what, mean, want)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_white_furnace(name):
    variant, want, integrators, tol = CASES[name]
    sc, inner = body_mask(variant)
    o = orc.OracleScene(sc.flat, sc.rand_calls)
    for integ in integrators:
        we, acc, _ = o.render(sc.camera, W, H, SEED, 0, 48, INTEGRATORS[integ], 8)
        img = orc.image_of(we, acc).reshape(H, W, 3)
        check(img, inner, want, tol, (name, integ))
        if not inner[0, 0]:
            assert np.allclose(img[0, 0], 1.0, atol=1e-5), (name, integ, img[0, 0])  # the environment itself
    o.close()


def test_oracle_rough_glass_loses_but_never_gains_energy():
    """The microfacet pair has no multiple-scattering term: it loses energy and must never gain any."""
    sc, inner = body_mask(3)
    o = orc.OracleScene(sc.flat, sc.rand_calls)
    we, acc, _ = o.render(sc.camera, W, H, SEED, 0, 48, pa.LI_VOLPATH, 8)
    img = orc.image_of(we, acc).reshape(H, W, 3)
    m = img[inner].mean()
    assert 0.5 < m < 1.005, m  # alpha = RoughnessToAlpha(0.1) = 0.46: two rough interfaces without a multiple-scattering term lose a third
    o.close()
