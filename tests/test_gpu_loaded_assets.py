"""Assets that came through the loaders, rendered: an OBJ with two material groups, its MTL
library and a particle file go through assets.* into a flat scene; the HIP path is compared with the oracle on
that scene, and with the same geometry handed over as plain arrays."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc
from pathtracer_amd import assets
from pathtracer_amd import scene as S

import common
from test_gpu_parity import check_closest

pytestmark = pytest.mark.gpu
W, H, SEED = 64, 48, 21
FMAX = np.finfo(np.float32).max


def torus(nu, nv, R=1.0, r=0.4):
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    P = np.stack([(R + r * np.cos(vv)) * np.cos(uu), r * np.sin(vv), (R + r * np.cos(vv)) * np.sin(uu)], -1).reshape(-1, 3).astype(np.float32)
    quads = []
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv
            quads.append((a, b, c, d))
    return P, np.array(quads)


def write_assets(tmp_path):
    P, Q = torus(24, 12)
    half = len(Q) // 2
    lines = ["mtllib torus.mtl", "o torus"] + ["v %r %r %r" % tuple(float(x) for x in p) for p in P]
    lines += ["usemtl coat"] + ["f %d %d %d %d" % tuple(q + 1) for q in Q[:half]]
    lines += ["usemtl chalk"] + ["f %d %d %d %d" % tuple(q - len(P)) for q in Q[half:]]  # relative indices
    (tmp_path / "torus.obj").write_text("\n".join(lines) + "\n")
    (tmp_path / "torus.mtl").write_text("newmtl coat\nKd 0.7 0.2 0.1\nKs 0.4 0.4 0.4\nnewmtl chalk\nKs 0.3 0.5 0.7\nnewmtl sun\nKe 12 12 10\n")
    rng = np.random.default_rng(2)
    pts = rng.uniform(-0.3, 0.3, (40, 3)).astype(np.float32) + np.array([0, 0.9, 0], np.float32)
    (tmp_path / "drops.txt").write_text("%d\n" % len(pts) + "".join("%r %r %r %d\n" % (float(a), float(b), float(c), i % 2) for i, (a, b, c) in enumerate(pts)))
    return P, Q, pts


def test_loaded_scene_matches_the_oracle_and_the_array_path(cuda, tmp_path):
    P, Q, pts = write_assets(tmp_path)
    meshes = assets.load_obj(tmp_path / "torus.obj")
    assert [m["mtl_name"] for m in meshes] == ["coat", "chalk"]
    lib = assets.MtlLibrary(meshes[0]["mtl_file"], str(tmp_path) + "/", times=len(meshes))
    (vecs, scal) = assets.load_particles(tmp_path / "drops.txt", "vs")
    assert np.array_equal(vecs[0], pts)
    assert lib.material_numbers(lib.find("coat"))[0] == "plastic" and lib.material_numbers(lib.find("chalk"))[0] == "matte"

    def build(use_loader):
        b = pa.FlatSceneBuilder()
        place = S.translate(0, 0.2, 0) @ S.rotate_x(35)
        for m in meshes:
            mat = lib.material(lib.find(m["mtl_name"]))
            if use_loader:
                b.parsed_mesh(m, mat, to_world=place)
            else:  # the same triangles as plain arrays
                b.mesh(m["p"], m["indices"][:, :, 0], mat, to_world=place)
        b.cloud(vecs[0][scal[0] == 0], 0.05, pa.material("glass", index=1.4))
        b.rectangle(9, 9, pa.material("matte", kd=(0.6, 0.6, 0.6)), to_world=S.translate(0, -0.8, 0) @ S.rotate_x(-90))
        b.rectangle(2, 2, lib.material(lib.find("sun")), to_world=S.translate(0, 3.5, 0) @ S.rotate_x(90))
        return b.finish()

    sc = build(True)
    cam = pa.look_at_camera((0, 1.8, 4.5), (0, 0.2, 0), (0, 1, 0), 42, W / H)
    o = orc.OracleScene(sc.flat, 0)
    films = []
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        oo, od, _ = o.generate_rays(cam, W, H, SEED, 0)
        tmax = np.full(W * H, FMAX, np.float32)
        g = common.gpu_trace_closest(r, oo, od, tmax)
        check_closest(g, o.trace_closest(oo, od, tmax))
        assert (g["prim"] == 0).any() and (g["prim"] == 1).any()  # both material groups are in view
        for integ in (pa.LI_PATH, pa.LI_VOLPATH):
            r.film_init(W, H, SEED)
            r.render(6, integ)
            img = r.image()
            films.append(r.film()[0].copy())
            we, acc, _ = o.render(cam, W, H, SEED, 0, 6, integ, 8)
            ref = orc.image_of(we, acc).reshape(H, W, 3)
            mean_rel, rmse_rel, off = common.image_stats(img, ref)
            # same image tolerances as test_gpu_parity.test_image
            assert mean_rel < 5e-3 and rmse_rel < 5e-2 and off < 0.02, (integ, mean_rel, rmse_rel, off)
        rgb = r.rgb8()
        assert rgb.max() > 128 and rgb.std() > 10  # a lit picture, not a constant one
    with pa.Renderer(0) as r:
        r.upload(build(False))
        r.set_camera(cam)
        for k, integ in enumerate((pa.LI_PATH, pa.LI_VOLPATH)):
            r.film_init(W, H, SEED)
            r.render(6, integ)
            assert np.array_equal(r.film()[0], films[k])  # bit-identical: the loader adds nothing of its own
    o.close()


def test_reference_style_host_program_renders_a_png(cuda, tmp_path):
    """examples/obj_scene.cpp: LoadObj -> MTLParseAll -> MakeMTLMaterial -> Render -> ImageWrite, no C-ABI in sight."""
    import subprocess
    from test_cpu_film_output import decode_png
    from test_cpu_loaders import example_binary
    write_assets(tmp_path)
    png = tmp_path / "output.png"
    run = subprocess.run([example_binary(), str(tmp_path) + "/", "torus.obj", "--render", str(png), "8"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=100)
    assert run.returncode == 0, run.stdout
    assert "Saved PNG" in run.stdout and " * Rendered using 8 samples" in run.stdout
    img = decode_png(png.read_bytes())
    assert img.shape == (120, 160, 3) and img.max() > 128 and img.std() > 10
