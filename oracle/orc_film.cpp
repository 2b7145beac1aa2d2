/*
 * orc_film.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * CPU restatement of the film resolve: GetPixelRGB (src/core/camera.cpp:36-51), ReinhardMap /
 * NaughtyDogMap / ExponentialMap (camera.cpp:6-33) and GammaCorrect (src/core/geometry.h:73-76,
 * 1663-1670), followed by ImageWrite's narrowing of the ints to bytes (camera.cpp:65-68).
 */
#include "orc_math.h"
#include <cmath>
#include <cstdint>

namespace orc {

static Float GammaCorrect(Float value) { /* geometry.h:73-76 */
    if (value <= 0.0031308f) return 12.92f * value;
    return 1.055f * std::pow(value, (Float)(1.f / 2.4f)) - 0.055f;
}
static V3 GammaCorrect(const V3 &v) { return V3(GammaCorrect(v.x), GammaCorrect(v.y), GammaCorrect(v.z)); }
static V3 AddS(const V3 &v, Float s) { return V3(v.x + s, v.y + s, v.z + s); } /* vec3 + scalar, geometry.h */

static V3 ReinhardMap(V3 value, Float) { /* camera.cpp:6-11 */
    value = divv(value, AddS(value, 1.f));
    return GammaCorrect(value);
}
static V3 NaughtyDogMap(V3 value, Float exposure) { /* camera.cpp:13-27 */
    float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, W = 11.2f;
    value *= exposure;
    value = AddS(divv(AddS(value * AddS(A * value, C * B), D * E), AddS(value * AddS(A * value, B), D * F)), -(E / F));
    float white = ((W * (A * W + C * B) + D * E) / (W * (A * W + B) + D * F)) - E / F;
    value = value / white;
    return GammaCorrect(value);
}
static V3 ExponentialMap(V3 value, Float exposure) { /* camera.cpp:29-33 */
    V3 v = -value * exposure;
    value = V3(1.f) - V3(std::exp(v.x), std::exp(v.y), std::exp(v.z));
    return GammaCorrect(value);
}

} // namespace orc

extern "C" int orc_film_rgb8(int npix, const float *we, const double *acc, float exposure, int algo, unsigned char *rgb) {
    using namespace orc;
    for (int i = 0; i < npix; i++) {
        V3 e(we[3 * i], we[3 * i + 1], we[3 * i + 2]);
        if (acc[i] == 0) { /* the reference asserts; the boundary defines 0 */
            rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0;
            continue;
        }
        Float invNs = 1.0f / acc[i]; /* double division, narrowed (camera.cpp:39) */
        e *= invNs;
        switch (algo) {
        case 0: e = ReinhardMap(e, exposure); break;
        case 1: e = ExponentialMap(e, exposure); break;
        case 2: e = NaughtyDogMap(e, exposure); break;
        default: return 1;
        }
        double c[3] = {255.999 * e.x, 255.999 * e.y, 255.999 * e.z};
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = c[k] == c[k] ? (unsigned char)(int)c[k] : 0;
    }
    return 0;
}
