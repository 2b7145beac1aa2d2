"""More answers from physics (see test_cpu_furnace.py), for the parts a constant environment does not reach:
area lights and their sampling, and transmittance through a homogeneous medium.

* a Lambertian floor under a diffuse disk emitter of radius R at height h: on the axis the irradiance is
  pi L R^2 / (R^2 + h^2), so the floor shows rho L R^2 / (R^2 + h^2);
* under a sphere emitter of radius r whose centre is d above the floor: pi L r^2 / d^2 -> rho L r^2 / d^2;
* an emitter seen through a purely absorbing sphere of medium (sigma_s = 0): L exp(-sigma_a * chord).
Scenes are assembled with FlatSceneBuilder; the oracle is checked here, the HIP path in test_gpu_known_answers.py."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc
from pathtracer_amd import scene as S

W, H, SEED = 48, 36, 9
INTEGRATORS = {"direct": pa.LI_DIRECT, "path": pa.LI_PATH, "pathsampled": pa.LI_PATHSAMPLED, "volpath": pa.LI_VOLPATH}
RHO, L_E = 0.6, 8.0


def light_scene(kind):
    b = pa.FlatSceneBuilder()
    b.rectangle(40, 40, pa.material("matte", kd=(RHO, RHO, RHO)), to_world=S.rotate_x(-90))  # floor y = 0, facing up
    if kind == "disk":
        R, h = 1.0, 2.0
        b.disk(0, R, 0, 360, pa.material("emissive", L=(L_E, L_E, L_E)), to_world=S.translate(0, h, 0) @ S.rotate_x(90))
        want = RHO * L_E * R * R / (R * R + h * h)
    else:
        r, d = 0.5, 2.5
        b.sphere(r, pa.material("emissive", L=(L_E, L_E, L_E)), to_world=S.translate(0, d, 0))
        want = RHO * L_E * r * r / (d * d)
    # a long lens on the axis point: the patch in view is a few centimetres wide, over which the irradiance is flat
    cam = pa.look_at_camera((3, 2.5, -6), (0, 0, 0), (0, 1, 0), 0.6, W / H)
    return b.finish(), cam, want


def medium_scene():
    sa, r = 0.7, 1.0
    b = pa.FlatSceneBuilder()
    b.rectangle(30, 30, pa.material("emissive", L=(1, 1, 1)), to_world=S.translate(0, 0, 5))  # the thing seen, behind the ball
    b.sphere(r, None, to_world=S.translate(0, 0, 0), medium=pa.medium((sa, sa, sa), (0, 0, 0)), no_mat=True)
    cam = pa.look_at_camera((0, 0, -6), (0, 0, 0), (0, 1, 0), 12, W / H)
    return b.finish(), cam, sa, r


SPP_CPU = 128


def check_light(kind, name, got, want):
    """Li_Direct and Li_PathSampled sample the light: exact to the noise.  Li_Path (main.cpp:362-410) is a brute-force
    tracer that finds the emitter by chance: unbiased, noisy.  Li_VolPath loses the BSDF-sampled half of the MIS
    estimate on emitters - IntersectTr reports "nothing hit" when the first thing hit is an emitter
    (primitive.cpp:433-461) and an area light has no radiance for escaping rays - so it is exact for a small light
    (the light-sampling half carries all the weight) and a few per cent low for a large one.  The restatement keeps
    that; what physics pins there is the bracket."""
    if name == "path":
        assert abs(got - want) <= 0.08 * want, (kind, name, got, want)
    elif name == "volpath" and kind == "disk":
        assert 0.90 * want <= got <= 0.99 * want, (kind, name, got, want)
    else:
        assert abs(got - want) <= 0.015 * want, (kind, name, got, want)


def central(img, k=6):
    return img[H // 2 - k:H // 2 + k, W // 2 - k:W // 2 + k]


@pytest.mark.parametrize("kind", ["disk", "sphere"])
def test_oracle_floor_under_an_area_light(kind):
    sc, cam, want = light_scene(kind)
    o = orc.OracleScene(sc.flat, 0)
    for name, integ in INTEGRATORS.items():
        we, acc, _ = o.render(cam, W, H, SEED, 0, SPP_CPU, integ, 8)
        img = orc.image_of(we, acc).reshape(H, W, 3)
        check_light(kind, name, central(img).mean(), want)
    o.close()


def chord_transmittance(o, cam, sa, r):
    oo, od, _ = o.generate_rays(cam, W, H, SEED, 0)
    d = od / np.linalg.norm(od, axis=0)
    b2 = (oo * oo).sum(0) - (oo * d).sum(0) ** 2  # squared distance of the ray from the ball's centre
    chord = 2 * np.sqrt(np.maximum(r * r - b2, 0))
    return np.exp(-sa * chord).reshape(H, W)


def test_oracle_transmittance_through_an_absorbing_ball():
    sc, cam, sa, r = medium_scene()
    o = orc.OracleScene(sc.flat, 0)
    want = chord_transmittance(o, cam, sa, r)
    we, acc, _ = o.render(cam, W, H, SEED, 0, 256, pa.LI_VOLPATH, 8)
    img = orc.image_of(we, acc).reshape(H, W, 3)
    assert central(want).max() < 0.3  # the ball really is in the way (exp(-1.4) = 0.25 through the middle)
    got, exp = central(img).mean(), central(want).mean()
    assert abs(got - exp) <= 0.03 * exp, (got, exp)
    assert np.allclose(img[0, 0], 1.0, atol=1e-5)  # beside the ball: the emitter itself
    o.close()
