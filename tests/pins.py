"""Known answers that no implementation had a hand in: normalisation integrals, closed forms and sample / density
consistency of the shade stage's building blocks (reference: src/core/reflection.cpp:36-589, microfacet.cpp:3-120,
material.cpp:5-86, medium.cpp:4-64, sampling.h:19-77, bssrdf.cpp:383-445).  The reference ships no tests and cannot be
built here, so the oracle can only be pinned by physics and mathematics; every check here takes a *backend* - the
CPU oracle (tests/test_cpu_pins.py) or the HIP path through the C-ABI (tests/test_gpu_pins.py) - and holds it to the
same numbers.

A backend provides: ev(kind, rows, prim=-1) -> [n, out]; bsdf_eval(prim, wo, wi) -> (f[n,3], pdf[n]);
bsdf_sample(prim, wo, u2) -> (wi, f, pdf, flags)."""
import numpy as np

BSDF_REFLECTION, BSDF_TRANSMISSION, BSDF_DIFFUSE, BSDF_GLOSSY, BSDF_SPECULAR = 1, 2, 4, 8, 16

# zoo scene (host/pt_scenes.cpp ZooScene): primitive index -> material
ZOO = {"matte": 0, "oren_nayar": 1, "mirror": 2, "glass": 3, "rough_glass": 4, "metal": 5, "plastic": 6, "uber": 7, "translucent": 8}
ZOO_KD = {"matte": (.7, .3, .2)}
# a scene of one sphere per material for the integral checks (material_scene below): primitive index -> material.  Translucent is
# split into its diffuse pair (Lambertian reflection + transmission) and its glossy pair (microfacet reflection + transmission).
MATS = {"matte": 0, "oren_nayar": 1, "mirror": 2, "glass": 3, "rough_glass": 4, "metal": 5, "plastic": 6, "translucent_diffuse": 7,
        "translucent_glossy": 8, "disney_diffuse": 9, "disney_full": 10, "disney_coat": 11}
# the Disney material (PTR_MAT_DISNEY): diffuse + retro only; all five lobes; clearcoat alone (black base colour)
DISNEY = {"disney_diffuse": dict(color=(.6, .5, .4), roughness=.7),
          "disney_full": dict(color=(.6, .5, .4), roughness=.7, sheen=1.0, sheen_tint=.4, clearcoat=.8, clearcoat_gloss=.3, flatness=.35),
          "disney_coat": dict(color=(0, 0, 0), roughness=.5, clearcoat=1.0, clearcoat_gloss=.3)}
UNREACHABLE_T = ("rough_glass", "translucent_glossy")  # carry a MicrofacetTransmission lobe (eta 1.5)


def material_scene():
    from pathtracer_amd import scene as S

    b = S.FlatSceneBuilder()
    b.sphere(1, S.material("matte", kd=ZOO_KD["matte"]))
    b.sphere(1, S.material("matte", kd=(.3, .6, .3), sigma=20.0))
    b.sphere(1, S.material("mirror", kr=(.9, .9, .9)))
    b.sphere(1, S.material("glass", kr=(1, 1, 1), kt=(1, 1, 1), index=1.5))
    b.sphere(1, S.material("glass", kr=(.9, .9, .9), kt=(.95, .95, .95), index=1.5, uroughness=.1, vroughness=.2))
    b.sphere(1, S.material("metal", eta=(.2, .92, 1.1), k=(3.9, 2.45, 2.14), uroughness=.05, vroughness=.05))
    b.sphere(1, S.material("plastic", kd=(.2, .2, .7), ks=(.6, .6, .6), roughness=.1))
    b.sphere(1, S.material("translucent", kd=(.5, .4, .3), ks=(0, 0, 0), reflect=(.6, .6, .6), transmit=(.4, .4, .4), roughness=.2))
    b.sphere(1, S.material("translucent", kd=(0, 0, 0), ks=(.3, .3, .3), reflect=(.6, .6, .6), transmit=(.4, .4, .4), roughness=.2))
    for name in ("disney_diffuse", "disney_full", "disney_coat"):
        b.sphere(1, S.material("disney", **DISNEY[name]))
    return b.finish()


def sphere_grid(nt, npz, hemisphere=False):
    """midpoint quadrature over the sphere (or the +z hemisphere): directions [n, 3] and solid-angle weights [n]"""
    tmax = np.pi / 2 if hemisphere else np.pi
    th = (np.arange(nt) + 0.5) / nt * tmax
    ph = (np.arange(npz) + 0.5) / npz * 2 * np.pi
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1).reshape(-1, 3)
    w = (np.sin(T) * (tmax / nt) * (2 * np.pi / npz)).reshape(-1)
    return d.astype(np.float32), w


def stratified(n_side, seed=0):
    rng = np.random.default_rng(seed)
    i, j = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    u = np.stack([(i + rng.random(i.shape)) / n_side, (j + rng.random(j.shape)) / n_side], -1).reshape(-1, 2)
    return np.clip(u, 1e-6, 1 - 1e-6).astype(np.float32)


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


WOS = [unit((0, 0, 1)), unit((0.5, 0.2, 0.8)), unit((0.8, -0.3, 0.35))]


# ---------------------------------------------------------------------------------------------- Fresnel
def check_fresnel(B):
    for eta in (1.33, 1.5, 2.4):
        F0 = B.ev("fr_dielectric", [[1.0, 1.0, eta]])[0, 0]
        assert abs(F0 - ((eta - 1) / (eta + 1)) ** 2) < 1e-6  # normal incidence
        assert B.ev("fr_dielectric", [[1e-6, 1.0, eta]])[0, 0] > 0.999  # grazing
        # from inside, beyond the critical angle: total internal reflection (cosThetaI < 0: the interface is seen from the etaT side)
        cos_c = np.sqrt(1 - 1 / eta ** 2)
        assert B.ev("fr_dielectric", [[-0.9 * cos_c, 1.0, eta]])[0, 0] == 1.0
        # the same interface from either side: F(theta_i; n1 -> n2) = F(theta_t; n2 -> n1)
        ci = np.linspace(0.05, 1, 40)
        ct = np.sqrt(1 - (1 - ci ** 2) / eta ** 2)
        a = B.ev("fr_dielectric", np.stack([ci, np.ones_like(ci), np.full_like(ci, eta)], 1))[:, 0]
        b = B.ev("fr_dielectric", np.stack([ct, np.full_like(ci, eta), np.ones_like(ci)], 1))[:, 0]
        assert np.allclose(a, b, atol=2e-6)
        # Brewster's angle: the p-polarised part vanishes, F = Rs / 2
        tb = np.arctan(eta)
        rs = ((np.cos(tb) - eta * np.cos(np.arcsin(np.sin(tb) / eta))) / (np.cos(tb) + eta * np.cos(np.arcsin(np.sin(tb) / eta)))) ** 2
        assert abs(B.ev("fr_dielectric", [[np.cos(tb), 1.0, eta]])[0, 0] - rs / 2) < 1e-6
    eta, k = np.array([.2, .92, 1.1]), np.array([3.9, 2.45, 2.14])
    F = B.ev("fr_conductor", [[1.0, 1, 1, 1, *eta, *k]])[0]
    assert np.allclose(F, ((eta - 1) ** 2 + k ** 2) / ((eta + 1) ** 2 + k ** 2), atol=2e-6)
    assert (B.ev("fr_conductor", [[1e-6, 1, 1, 1, *eta, *k]])[0] > 0.999).all()
    # k = 0: a conductor without absorption is the dielectric
    c = np.linspace(0.05, 1, 20)
    rows = np.concatenate([c[:, None], np.ones((20, 3)), np.full((20, 3), 1.5), np.zeros((20, 3))], 1)
    d = B.ev("fr_dielectric", np.stack([c, np.ones_like(c), np.full_like(c, 1.5)], 1))[:, 0]
    assert np.allclose(B.ev("fr_conductor", rows)[:, 0], d, atol=2e-6)


# ---------------------------------------------------------------------------------------------- GGX
def check_ggx(B):
    wh, w = sphere_grid(1024, 512, hemisphere=True)
    for ax, ay in ((0.3, 0.3), (0.08, 0.08), (0.1, 0.45)):
        a2 = np.full((wh.shape[0], 2), [ax, ay], np.float32)
        D = B.ev("ggx_d", np.concatenate([a2, wh], 1))[:, 0].astype(np.float64)
        assert abs((D * wh[:, 2] * w).sum() - 1) < 2e-3  # the microfacets' projected area is the macro surface's
        for wo in WOS:
            wo_r = np.tile(wo, (wh.shape[0], 1))
            G1 = B.ev("ggx_g1", np.concatenate([a2[:1], wo[None]], 1))[0, 0]
            # Smith: what a direction sees of the microsurface is the macro surface's projected area
            vis = (G1 * np.maximum(0, (wh * wo).sum(1)) * D * w).sum()
            assert abs(vis - wo[2]) < 3e-3 * max(wo[2], 0.3), (ax, ay, wo, vis)
            pdf = B.ev("ggx_pdf", np.concatenate([a2, wo_r, wh], 1))[:, 0].astype(np.float64)
            # the visible-normal density lives on the facets that face wo.  (Pdf is written with |wo . wh| (microfacet.cpp),
            # so it is also positive on back-facing normals, which the sampler never returns: over the whole hemisphere it
            # integrates to more than one at grazing wo - the reference's behaviour, kept; the facing part is the density.)
            pdf = np.where((wh * wo).sum(1) > 0, pdf, 0.0)
            assert abs((pdf * w).sum() - 1) < 3e-3
            # the sampler draws from that density: moments of the samples against the quadrature of the density
            u = stratified(300, 1)
            s = B.ev("ggx_sample_wh", np.concatenate([np.full((u.shape[0], 2), [ax, ay], np.float32), np.tile(wo, (u.shape[0], 1)), u], 1)).astype(np.float64)
            assert np.allclose(np.linalg.norm(s, axis=1), 1, atol=1e-4)
            for g in (lambda v: v[:, 2], lambda v: v[:, 0], lambda v: v[:, 1] ** 2, lambda v: v[:, 0] * v[:, 2]):
                assert abs(g(s).mean() - (g(wh.astype(np.float64)) * pdf * w).sum()) < 4e-3, (ax, ay, wo)
    # RoughnessToAlpha is its own definition (microfacet.h:38-43): the polynomial in log(roughness)
    r = np.array([0.001, 0.01, 0.1, 0.5, 1.0])
    x = np.log(np.maximum(r, 1e-3))
    want = 1.62142 + 0.819955 * x + 0.1734 * x ** 2 + 0.0171201 * x ** 3 + 0.000640711 * x ** 4
    assert np.allclose(B.ev("rough_to_alpha", r[:, None])[:, 0], want, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------- Henyey-Greenstein
def check_hg(B):
    c = np.linspace(-1, 1, 400001)
    for g in (-0.7, 0.0, 0.3, 0.85):
        p = B.ev("hg_p", np.stack([c, np.full_like(c, g)], 1))[:, 0].astype(np.float64)
        assert abs(np.trapezoid(p, c) * 2 * np.pi - 1) < 1e-4
        # PBRT's convention: the argument is cos(wo, wi) with wo pointing back along the ray, so forward scattering is -1
        assert abs(np.trapezoid(p * c, c) * 2 * np.pi + g) < 1e-4
        u = stratified(300, 2)
        for wo in WOS[:2]:
            s = B.ev("hg_sample", np.concatenate([np.full((u.shape[0], 1), g, np.float32), np.tile(wo, (u.shape[0], 1)), u], 1)).astype(np.float64)
            wi, ps = s[:, :3], s[:, 3]
            assert np.allclose(np.linalg.norm(wi, axis=1), 1, atol=1e-4)
            cosw = wi @ wo.astype(np.float64)
            assert abs(cosw.mean() + g) < 3e-3  # mean cosine
            assert abs((cosw ** 2).mean() - np.trapezoid(p * c * c, c) * 2 * np.pi) < 3e-3
            again = B.ev("hg_p", np.stack([cosw, np.full_like(cosw, g)], 1))[:, 0]
            assert np.allclose(ps, again, rtol=2e-3, atol=1e-6)  # the value returned with the sample is the density at the sample


# ---------------------------------------------------------------------------------------------- Distribution2D
def check_env(B, nu, nv):
    """the light map's piecewise-constant density over [0,1]^2 (sampling.h:49-77)"""
    cu, cv = (np.arange(nu) + 0.5) / nu, (np.arange(nv) + 0.5) / nv
    U, V = np.meshgrid(cu, cv, indexing="xy")
    uv = np.stack([U.ravel(), V.ravel()], 1).astype(np.float32)
    pdf = B.ev("env_pdf", uv)[:, 0].astype(np.float64)
    assert (pdf >= 0).all() and abs(pdf.mean() - 1) < 1e-4  # integrates to one (constant per cell: the mean over cell centres)
    u = stratified(400, 3)
    s = B.ev("env_sample", u).astype(np.float64)
    assert ((s[:, :2] >= 0) & (s[:, :2] <= 1)).all()
    again = B.ev("env_pdf", s[:, :2].astype(np.float32))[:, 0]
    # (a sample that rounds onto a cell border reads the neighbouring cell's density)
    assert np.isclose(s[:, 2], again, rtol=1e-4, atol=1e-6).mean() > 0.995
    # the samples follow the density: moments against the quadrature over cells
    for g in (lambda a, b: a, lambda a, b: b, lambda a, b: a * b, lambda a, b: (b - 0.5) ** 2):
        assert abs(g(s[:, 0], s[:, 1]).mean() - (g(uv[:, 0].astype(np.float64), uv[:, 1].astype(np.float64)) * pdf).mean()) < 3e-3
    # ... and cell by cell on a coarse histogram
    hist, _, _ = np.histogram2d(s[:, 0], s[:, 1], bins=(8, 4), range=((0, 1), (0, 1)))
    want = pdf.reshape(nv, nu).reshape(4, nv // 4, 8, nu // 8).mean(axis=(1, 3)).T / 32
    assert np.abs(hist / len(s) - want).max() < 4e-3


# ---------------------------------------------------------------------------------------------- BSDFs of the materials
def check_material_integrals(B, name, prim, fine=False):
    """density integrates to (at most) one, sampling agrees with evaluation, energy is conserved"""
    wi, w = sphere_grid(1536 if fine else 512, 1536 if fine else 512)
    u = stratified(400, 4)
    for wo in WOS:
        f, pdf = B.bsdf_eval(prim, np.tile(wo, (wi.shape[0], 1)), wi)
        f, pdf = f.astype(np.float64), pdf.astype(np.float64)
        assert (f >= 0).all() and (pdf >= 0).all() and np.isfinite(f).all() and np.isfinite(pdf).all()
        # The reference's MicrofacetTransmission::f / Pdf (reflection.cpp:44-52, 180-200; PBRT-v3 before its later fix) do not
        # reject pairs whose generalised half vector has wo and wi on the SAME side (wo.wh * wi.wh > 0): no refraction maps wo
        # to such a wi, the sampler never returns one, yet f and Pdf are positive there.  Kept, like every other quirk; the
        # sampler is compared with the quadrature over the directions it can reach.
        reach = np.ones(wi.shape[0], bool)
        if name in UNREACHABLE_T:
            below = wi[:, 2] < 0
            wh = wo[None].astype(np.float64) + wi.astype(np.float64) * 1.5
            wh /= np.linalg.norm(wh, axis=1, keepdims=True)
            wh[wh[:, 2] < 0] *= -1  # the microfacet normal, on the upper side like the ones the sampler draws (visible from wo)
            reach = ~below | (((wh @ wo.astype(np.float64)) > 0) & ((wh * wi).sum(1) < 0))
        total = (pdf * w)[reach].sum()
        # a visible-normal microfacet lobe loses the half vectors whose reflection dips below the surface: <= 1, and not by much
        # (the Disney clearcoat samples GTR1 half vectors from the whole distribution, not the visible ones, and GTR1 has long
        # tails: at grazing wo a quarter of the reflections dip below the surface)
        assert (0.6 if name == "disney_coat" else 0.80) < total < 1.004, (name, wo, total)
        if name in ("matte", "oren_nayar"):
            assert abs(total - 1) < 2e-3
        albedo = (f * np.abs(wi[:, 2:3]) * w[:, None]).sum(0)
        # energy conservation of a lobe; plastic and translucent ADD a diffuse and a glossy lobe with no coupling (plastic.cpp,
        # translucent.cpp), so their sum may pass one by the glossy lobe's share at grazing angles - bounded by Kd + Ks
        assert (albedo < (1.01 if name in ("matte", "oren_nayar", "metal", "rough_glass", "disney_diffuse", "disney_coat") else 1.31)).all(), (name, wo, albedo)
        if name == "matte":
            assert np.allclose(albedo, ZOO_KD["matte"], atol=2e-3)  # Lambert: f = R / pi
            assert np.allclose(f[wi[:, 2] > 0], np.array(ZOO_KD["matte"]) / np.pi, rtol=1e-5)
        # sampling: E[f cos / pdf] over the sampler == the quadrature of f cos (only where the density is positive)
        swi, sf, spdf, fl = B.bsdf_sample(prim, np.tile(wo, (u.shape[0], 1)), u)
        ok = spdf > 0
        assert not (fl[ok] & BSDF_SPECULAR).any()  # (the materials passed here have no specular lobe)
        est = np.zeros(3)
        est_w = (sf[ok].astype(np.float64) * np.abs(swi[ok, 2:3].astype(np.float64)) / spdf[ok, None].astype(np.float64))
        est = est_w.sum(0) / u.shape[0]
        integrand = f * np.abs(wi[:, 2:3]) * w[:, None]
        ref = integrand[(pdf > 0) & reach].sum(0)
        assert np.allclose(est, ref, rtol=0.03, atol=4e-3), (name, wo, est, ref)
        assert (integrand[(pdf > 0) & ~reach].sum(0) <= 0.5 * np.maximum(ref, 1e-3)).all()  # what the quirk adds (most at grazing wo) stays a fraction
        # the (f, pdf) returned with a sample are f(wo, wi), pdf(wo, wi) at the sampled direction
        idx = np.nonzero(ok)[0][:: max(1, ok.sum() // 4000)]
        f2, p2 = B.bsdf_eval(prim, np.tile(wo, (idx.size, 1)), swi[idx])
        assert np.allclose(f2, sf[idx], rtol=2e-3, atol=1e-5), name
        assert np.allclose(p2, spdf[idx], rtol=2e-3, atol=1e-5), name


def disney_f(p, wo, wi):
    """The five lobes from the published formulas (Burley 2012 "Physically based shading at Disney", eqs. of section 5, and
    Burley 2015 "Extending the Disney BRDF to a BSDF with integrated subsurface scattering", eq. 4), in float64 numpy,
    written from the papers - not from the reference, the oracle or the device code."""
    wo, wi = wo.astype(np.float64), wi.astype(np.float64)
    c = np.array(p.get("color", (.5, .5, .5)), np.float64)
    rough, sheen, tint_w = p.get("roughness", .5), p.get("sheen", 0.), p.get("sheen_tint", .5)
    cc, gloss, flat = p.get("clearcoat", 0.), p.get("clearcoat_gloss", 1.), p.get("flatness", 0.)
    co, ci = np.abs(wo[:, 2]), np.abs(wi[:, 2])
    h = wo + wi
    h /= np.linalg.norm(h, axis=1, keepdims=True)
    cd = (wi * h).sum(1)  # cos of the difference angle
    sw = lambda x: np.clip(1 - x, 0, 1) ** 5
    Fo, Fi = sw(co), sw(ci)
    f = np.zeros((wo.shape[0], 3))
    if c.any():
        diffuse = (1 - Fo / 2) * (1 - Fi / 2) / np.pi
        f += (1 - flat) * c * diffuse[:, None]
        if flat > 0:  # Hanrahan-Krueger-like fake subsurface term
            Fss90 = cd * cd * rough
            Fss = (1 + (Fss90 - 1) * Fo) * (1 + (Fss90 - 1) * Fi)
            f += flat * c * (1.25 * (Fss * (1 / (co + ci) - .5) + .5) / np.pi)[:, None]
        Rr = 2 * rough * cd * cd
        f += c * (Rr * (Fo + Fi + Fo * Fi * (Rr - 1)) / np.pi)[:, None]
        if sheen > 0:
            lum = .3 * c[0] + .6 * c[1] + 1. * c[2]
            tint = c / lum if lum > 0 else np.ones(3)
            f += sheen * ((1 - tint_w) + tint_w * tint) * sw(np.abs(cd))[:, None]
    if cc > 0:
        a = (1 - gloss) * .1 + gloss * .001
        a2 = a * a
        ch = np.abs(h[:, 2])
        D = (a2 - 1) / (np.pi * np.log(a2) * (1 + (a2 - 1) * ch * ch))  # GTR1
        F = .04 + .96 * sw((wo * h).sum(1))
        g = lambda x: 2 / (1 + np.sqrt(.0625 + (1 - .0625) * x * x))  # the separable GGX masking term of the Disney clearcoat, alpha = 0.25
        f += (cc * g(co) * g(ci) * F * D / 4)[:, None]
    return f


def check_disney_closed_form(B, name, prim):
    rng = np.random.default_rng(9)
    a, b = rng.normal(size=(6000, 3)), rng.normal(size=(6000, 3))
    a[:, 2], b[:, 2] = np.abs(a[:, 2]) + 0.02, np.abs(b[:, 2]) + 0.02
    a = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    f, pdf = B.bsdf_eval(prim, a, b)
    want = disney_f(DISNEY[name], a, b)
    assert np.allclose(f, want, rtol=2e-4, atol=2e-6), (name, np.abs(f - want).max())
    # opposite sides of the surface: every lobe is a reflection lobe
    b2 = b.copy()
    b2[:, 2] *= -1
    f2, pdf2 = B.bsdf_eval(prim, a, b2)
    assert (f2 == 0).all() and (pdf2 == 0).all()


def check_reciprocity(B, name, prim):
    rng = np.random.default_rng(5)
    a = rng.normal(size=(4000, 3))
    b = rng.normal(size=(4000, 3))
    a[:, 2], b[:, 2] = np.abs(a[:, 2]) + 0.05, np.abs(b[:, 2]) + 0.05  # both above the surface: the reflection lobes
    a = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    f1, _ = B.bsdf_eval(prim, a, b)
    f2, _ = B.bsdf_eval(prim, b, a)
    assert np.allclose(f1, f2, rtol=2e-3, atol=2e-5), name


def check_specular(B, mirror_prim, glass_prim, kr=0.9, eta=1.5):
    u = stratified(64, 6)
    for wo in WOS:
        wo_r = np.tile(wo, (u.shape[0], 1))
        wi, f, pdf, fl = B.bsdf_sample(mirror_prim, wo_r, u)
        assert np.allclose(wi, [-wo[0], -wo[1], wo[2]], atol=1e-6) and (fl == (BSDF_REFLECTION | BSDF_SPECULAR)).all()
        assert np.allclose(f * np.abs(wi[:, 2:3]) / pdf[:, None], kr, rtol=1e-5)  # FresnelNoOp mirror: throughput = Kr
        # specular glass (FresnelSpecular): reflect with probability F, refract by Snell's law otherwise; expected throughput
        # F R + (1 - F) T (etaI / etaT)^2 for radiance
        wi, f, pdf, fl = B.bsdf_sample(glass_prim, wo_r, u)
        F = float(B.ev("fr_dielectric", [[wo[2], 1.0, eta]])[0, 0])
        refl = (fl & BSDF_REFLECTION) != 0
        assert abs(refl.mean() - F) < 0.02
        assert np.allclose(wi[refl], [-wo[0], -wo[1], wo[2]], atol=1e-6)
        t = wi[~refl]
        assert (t[:, 2] < 0).all()
        sin_i, sin_t = np.sqrt(1 - wo[2] ** 2), np.sqrt(np.maximum(0, 1 - t[:, 2].astype(np.float64) ** 2))
        assert np.allclose(eta * sin_t, sin_i, atol=2e-5)  # Snell
        assert np.allclose(t[:, :2] @ np.array([wo[1], -wo[0]]), 0, atol=2e-5)  # in the plane of incidence
        thr = (f * np.abs(wi[:, 2:3]) / pdf[:, None])[:, 0]
        assert np.allclose(thr[refl], 1.0, rtol=1e-4) and np.allclose(thr[~refl], 1 / eta ** 2, rtol=1e-4)


# ---------------------------------------------------------------------------------------------- BSSRDF radial profile
def check_bssrdf(B, prim):
    for ch in range(3):
        u = ((np.arange(20000) + 0.5) / 20000).astype(np.float32)
        r = B.ev("sss_sample_sr", np.stack([np.full_like(u, ch), u], 1), prim=prim)[:, 0].astype(np.float64)
        ok = r >= 0
        assert ok.mean() > 0.99 and np.all(np.diff(r[ok]) >= -1e-6 * r[ok][1:])  # inverse CDF: monotone in u
        rmax = r[ok].max()
        rr = np.linspace(0, rmax, 200001)[1:]
        pdf = B.ev("sss_pdf_sr", np.stack([np.full_like(rr, ch), rr], 1).astype(np.float32), prim=prim)[:, 0].astype(np.float64)
        assert (pdf >= 0).all()
        mass = np.trapezoid(pdf * 2 * np.pi * rr, rr)
        assert 0.97 < mass < 1.01, (ch, mass)  # density over the plane, out to the largest radius the sampler reaches
        # the sampler follows it: mean radius by sampling and by quadrature
        assert abs(r[ok].mean() - np.trapezoid(pdf * 2 * np.pi * rr * rr, rr)) < 0.02 * r[ok].mean()
        sr = B.ev("sss_sr", rr[:, None].astype(np.float32), prim=prim).astype(np.float64)
        # (within a channel's own table range the profile is non-negative; beyond it the Catmull-Rom extrapolation may ring below
        # zero, and the reference's Clamp(Sr) clamps to [-1, 1], not [0, 1]: kept)
        assert (sr[:, ch] >= 0).all() and (sr <= 1).all() and (sr >= -1e-3).all()
        # Pdf_Sr is Sr normalised by the effective albedo: the two have the same shape
        q = sr[:, ch] / np.maximum(pdf, 1e-30)
        sel = (pdf > 1e-6 * pdf.max()) & (sr[:, ch] < 0.999)
        assert sel.sum() > 1000 and q[sel].std() < 0.02 * q[sel].mean()
        assert 0 < q[sel].mean() < 1  # ... and that albedo is below one
