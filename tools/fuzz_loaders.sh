#!/bin/bash
# Mutation fuzz of the host loaders (EXR incl. PIZ, PNG, JPEG, PPM, OBJ, MTL, particles) under host ASan + UBSan; no GPU.
#   bash tools/fuzz_loaders.sh [iterations per base file]
# Base files are written by the tests' own writers (tests/test_cpu_exr.py, test_cpu_loaders.py) and Pillow into /tmp/lf.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); N=${1:-1500}
mkdir -p /tmp/lf/m
g++ -std=c++14 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared -o /tmp/libpt_host_asan.so \
    $ROOT/pathtracer_amd/host/pt_scene.cpp $ROOT/pathtracer_amd/host/pt_scenes.cpp $ROOT/pathtracer_amd/host/pt_image.cpp $ROOT/pathtracer_amd/host/pt_loaders.cpp \
    -L$ROOT/pathtracer_amd/lib -lptr_hip -lz -Wl,-rpath,$ROOT/pathtracer_amd/lib
(cd $ROOT/tests && PYTHONPATH=$ROOT python3 - <<'PY'
import numpy as np
import test_cpu_exr as E, test_cpu_loaders as L
from PIL import Image
out = '/tmp/lf/'
n = 0
for comp in (E.NONE, E.RLE, E.ZIPS, E.ZIP, E.PIZ):
    for pt in (E.HALF, E.FLOAT):
        img = E.picture(37, 23, comp * 3 + pt)
        E.write_exr(out + "e%d.exr" % n, {"R": (pt, img[..., 0]), "G": (pt, img[..., 1]), "B": (pt, img[..., 2]), "A": (pt, np.ones((37, 23), np.float32))}, comp)
        n += 1
y, x = np.mgrid[0:64, 0:48]
smooth = np.stack([(x + y) / 128.0, (x * y % 17) / 17.0, np.sin(x / 5.0) * 0.5 + 0.5], -1).astype(np.float32)
E.write_exr(out + "e%d.exr" % n, {"R": (E.HALF, smooth[..., 0]), "G": (E.HALF, smooth[..., 1]), "B": (E.HALF, smooth[..., 2])}, E.PIZ)
a = np.random.default_rng(1).uniform(0, 255, (40, 56, 3)).astype(np.uint8)
Image.fromarray(a).save(out + "p0.png"); Image.fromarray(a[..., 0]).save(out + "p1.png")
Image.fromarray(a).convert("P").save(out + "p2.png"); Image.fromarray(np.dstack([a, a[..., :1]])).save(out + "p3.png")
Image.fromarray(a).save(out + "j0.jpg", quality=80); Image.fromarray(a).save(out + "j1.jpg", quality=80, progressive=True)
Image.fromarray(a).save(out + "j2.jpg", quality=80, subsampling=1, restart_marker_blocks=2); Image.fromarray(a[..., 0]).save(out + "j3.jpg", quality=60, progressive=True, optimize=True)
open(out + "o0.obj", "w").write(L.OBJ); open(out + "things.mtl", "w").write(L.MTL)
open(out + "t0.txt", "w").write("# header\n3\n0 0 0 1\n1.5 2.5 3.5 0\n\n-1 -2 -3 1\n")
open(out + "q0.ppm", "wb").write(b"P6\n4 3\n255\n" + bytes(range(36)))
PY
)
g++ -O1 -g -fsanitize=address,undefined $ROOT/tools/fuzz_loaders.cpp -o /tmp/lf/fuzz /tmp/libpt_host_asan.so -Wl,-rpath,/tmp -Wl,-rpath,$ROOT/pathtracer_amd/lib
ASAN_OPTIONS=detect_leaks=0:exitcode=99 /tmp/lf/fuzz $N > /tmp/lf/out.txt 2> /tmp/lf/err.txt && echo "no sanitizer report" || { echo "FAILED"; grep -n "ERROR\|runtime error\|child died" /tmp/lf/err.txt | head; exit 1; }
