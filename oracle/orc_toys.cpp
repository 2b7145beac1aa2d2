/*
 * orc_toys.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * ProceduralComponent::Map dispatch (src/core/procedural.cpp:380-400).  Only toy id 4
 * ("Terrain", an SDF box over the component's bounds) is restated so far; the origami toys
 * (ids 0-3, procedural.cpp:402-574) are a later row.
 */
#include "orc_scene.h"
#include <cstdio>
#include <cstdlib>

namespace orc {

static Float ExtentOn(const Bounds3 &b, int i) { /* geometry.h:1357-1362 */
    if (i == 0) return Absf(b.pMax.x - b.pMin.x);
    if (i == 1) return Absf(b.pMax.y - b.pMin.y);
    return Absf(b.pMax.z - b.pMin.z);
}

Float SDFMapToy(int id, const Bounds3 &bounds, const V3 &p, int *shapeId) {
    (void)shapeId;
    V3 length(ExtentOn(bounds, 0), ExtentOn(bounds, 1), ExtentOn(bounds, 2));
    V3 half = length * 0.5;
    if (id == 4) { /* SDF::Terrain == SDF::Box(q, half), procedural.cpp:398-400 */
        V3 d = Abs(p) - half;
        return Min(Max(d.x, Max(d.y, d.z)), 0.0) + Length(V3(Max(d.x, 0.f), Max(d.y, 0.f), Max(d.z, 0.f)));
    }
    fprintf(stderr, "oracle: procedural toy id %d is not restated yet\n", id);
    abort();
}

} // namespace orc
