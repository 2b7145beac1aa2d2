/*
 * pt_image.cpp — film output of the host mirror: ImageWrite (camera.cpp:53-85) and the PPM
 * writer (ppm.cpp).  The per-pixel work of GetPixelRGB (camera.cpp:36-51) — normalise, tone map,
 * gamma, quantise — runs on the device behind ptr_film_rgb8; what is left here is the container.
 * The reference hands the bytes to miniz (tdefl_write_image_to_png_file_in_memory_ex, 3 channels,
 * level 6, flip = MZ_TRUE); this file writes the same PNG structure (8-bit RGB, filter 0 on every
 * scanline, rows bottom-up -> top-down) with zlib's deflate at the same level, so the decoded
 * pixels are identical while the compressed stream need not be.
 */
#include "pt_scene.h"
#include <cstdio>
#include <cstring>
#include <iostream>
#include <zlib.h>

static void put_be32(std::vector<unsigned char> &v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
static void put_chunk(std::vector<unsigned char> &png, const char tag[4], const unsigned char *data, size_t len) {
    put_be32(png, (uint32_t)len);
    size_t at = png.size();
    png.insert(png.end(), tag, tag + 4);
    if (len) png.insert(png.end(), data, data + len);
    put_be32(png, (uint32_t)crc32(0L, png.data() + at, (uInt)(len + 4)));
}

bool EncodePNG(const unsigned char *rgb, int width, int height, bool flip, std::vector<unsigned char> &png) {
    if (!rgb || width <= 0 || height <= 0) return false;
    const size_t row = (size_t)width * 3;
    std::vector<unsigned char> raw((row + 1) * (size_t)height);
    for (int y = 0; y < height; y++) {
        const int src = flip ? height - 1 - y : y;
        raw[(row + 1) * y] = 0; /* filter type None */
        memcpy(&raw[(row + 1) * y + 1], rgb + row * src, row);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    png.assign(sig, sig + 8);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); /* 8-bit, RGB */
    put_chunk(png, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(png, "IDAT", z.data(), zlen);
    put_chunk(png, "IEND", nullptr, 0);
    return true;
}

void ImageRGB8(Image *image, Float exposure, ToneMapAlgorithm algorithm, unsigned char *data) {
    PtrContext *ctx = image->owner ? image->owner->ctx : nullptr;
    if (!ctx) { std::cout << "ImageWrite: image was never rendered" << std::endl; exit(1); }
    int rc = ptr_film_rgb8(ctx, exposure, (int)algorithm, data);
    if (rc) { std::cout << "ptr_film_rgb8: " << ptr_last_error() << std::endl; exit(1); }
}

void ImageWrite(Image *image, const char *path, Float exposure, ToneMapAlgorithm algorithm) {
    std::vector<unsigned char> data((size_t)image->pixels_count * 3), png;
    ImageRGB8(image, exposure, algorithm, data.data());
    if (!EncodePNG(data.data(), image->width, image->height, true, png)) {
        std::cout << "Failed to get PNG" << std::endl;
        return;
    }
    FILE *fp = fopen(path, "wb");
    if (!fp) { std::cout << "Failed to open " << path << std::endl; return; }
    fwrite(png.data(), 1, png.size(), fp);
    fclose(fp);
    std::cout << "Saved PNG " << path << std::endl;
}

/* ppm.cpp:4-32: binary "P6", rows written top-down from the bottom-up pixel array.  The reference's
 * PPMWriteFloat takes a per-pixel handler; the handler here is the device tone map. */
bool ImageWritePPM(Image *image, const char *path, Float exposure, ToneMapAlgorithm algorithm) {
    std::vector<unsigned char> data((size_t)image->pixels_count * 3);
    ImageRGB8(image, exposure, algorithm, data.data());
    FILE *fp = fopen(path, "wb");
    if (!fp) return false;
    fprintf(fp, "P6\n%d %d\n255\n", image->width, image->height);
    const size_t row = (size_t)image->width * 3;
    for (int j = image->height - 1; j >= 0; --j) fwrite(&data[row * j], 1, row, fp);
    fclose(fp);
    return true;
}

/* ppm.cpp:34-68: three header lines, then rows top-down into a bottom-up array */
bool PPMRead(const char *path, unsigned char **values, int &width, int &height) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    char line[256];
    bool ok = fgets(line, sizeof line, fp) && fgets(line, sizeof line, fp) && sscanf(line, "%d %d", &width, &height) == 2 &&
              fgets(line, sizeof line, fp) && width > 0 && height > 0;
    if (!ok) { fclose(fp); return false; }
    const size_t row = (size_t)width * 3;
    unsigned char *pixels = new unsigned char[row * height];
    for (int j = height - 1; j >= 0 && ok; --j) ok = fread(pixels + row * j, 1, row, fp) == row;
    fclose(fp);
    if (!ok) { delete[] pixels; return false; }
    *values = pixels;
    return true;
}

/* util.h:41-61 */
void ConvertPPMtoPNG(const char *path, const char *out) {
    unsigned char *values = nullptr;
    int width = 0, height = 0;
    if (!PPMRead(path, &values, width, height)) {
        printf("Failed to get values for %s\n", path);
        return;
    }
    std::vector<unsigned char> png;
    if (!EncodePNG(values, width, height, true, png)) printf("Failed to get PNG\n");
    else {
        FILE *fp = fopen(out, "wb");
        if (fp) { fwrite(png.data(), 1, png.size(), fp); fclose(fp); printf("Done\n"); }
    }
    delete[] values;
}

/* The reference's periodic backup (graphy.cpp:11,63: temp.ppm) stores tone-mapped bytes, from which a
 * render cannot be resumed.  The checkpoint here stores the accumulators themselves - we (3 x f32),
 * accWeight (f64), samples (i64) per pixel - so that ImageResume + Render continues the same
 * sample sequence (sample indices continue from `samples`, seeds are counter-based). */
static const char kCkptMagic[8] = {'P', 'T', 'R', 'F', 'I', 'L', 'M', '1'};
bool ImageCheckpoint(Image *image, const char *path) {
    PtrContext *ctx = image->owner ? image->owner->ctx : nullptr;
    if (!ctx) return false;
    const size_t n = (size_t)image->pixels_count;
    std::vector<float> we(3 * n);
    std::vector<double> aw(n);
    std::vector<int64_t> sm(n);
    if (ptr_film_read(ctx, we.data(), aw.data(), sm.data())) return false;
    std::string tmp = std::string(path) + ".part";
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) return false;
    int32_t dims[2] = {image->width, image->height};
    bool ok = fwrite(kCkptMagic, 1, 8, fp) == 8 && fwrite(dims, 4, 2, fp) == 2 && fwrite(we.data(), 4, 3 * n, fp) == 3 * n &&
              fwrite(aw.data(), 8, n, fp) == n && fwrite(sm.data(), 8, n, fp) == n;
    ok = (fclose(fp) == 0) && ok;
    if (!ok) { remove(tmp.c_str()); return false; }
    return rename(tmp.c_str(), path) == 0; /* a reader never sees a half-written file */
}
bool ImageResume(Image *image, Aggregator *scene, const char *path) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    char magic[8];
    int32_t dims[2];
    const size_t n = (size_t)image->pixels_count;
    std::vector<float> we(3 * n);
    std::vector<double> aw(n);
    std::vector<int64_t> sm(n);
    bool ok = fread(magic, 1, 8, fp) == 8 && memcmp(magic, kCkptMagic, 8) == 0 && fread(dims, 4, 2, fp) == 2 && dims[0] == image->width &&
              dims[1] == image->height && fread(we.data(), 4, 3 * n, fp) == 3 * n && fread(aw.data(), 8, n, fp) == n &&
              fread(sm.data(), 8, n, fp) == n;
    fclose(fp);
    if (!ok) return false;
    if (ptr_film_write(scene->ctx, we.data(), aw.data(), sm.data(), -1)) return false;
    image->owner = scene;
    return true;
}

/* camera.cpp:86-125.  The per-pixel PixelStats of the reference are kept as whole-image sums on the
 * device (ptr_counters_read); the report has the reference's lines and wording.  lightHitFromMedium
 * is not tracked by the reference's integrators either (it stays 0 there), so that line prints nan
 * exactly when the reference's does: when there were no light hits. */
void ImageStats(Image *image) {
    PtrContext *ctx = image->owner ? image->owner->ctx : nullptr;
    PtrCounters c;
    int64_t samples0 = 0;
    if (!ctx || ptr_counters_read(ctx, &c)) { std::cout << "ImageStats: no rendered film" << std::endl; return; }
    {
        std::vector<int64_t> sm((size_t)image->pixels_count);
        if (ptr_film_read(ctx, nullptr, nullptr, sm.data()) == 0 && !sm.empty()) samples0 = sm[0];
    }
    double totalPaths = (double)c.total_paths, totalHits = (double)(c.medium_hits + c.surface_hits);
    printf(" * Rendered using %ld samples\n", (long)samples0);
    printf(" * Zero radiance paths: %g%%\n", 100 * (double)c.zero_radiance / totalPaths);
    printf(" * Medium hits: %g%%\n", 100 * (double)c.medium_hits / totalHits);
    printf(" * Surface hits: %g%%\n", (double)c.surface_hits / totalHits); /* the reference scales the count, not the ratio (camera.cpp:115) */
    printf(" * Light hits: %g%%\n", 100 * (double)c.light_hits / totalHits);
    printf(" * Light hits from medium: %g%%\n", 100 * 0.0 / (double)c.light_hits);
    printf(" * Misses: %g%%\n", 100 * (double)c.misses / totalPaths);
}

/* ---- plain-C views for the Python mirror (pathtracer_amd/render.py) -------------------------------- */
extern "C" {
/* PNG of an RGB8 array; returns the byte count, or -1 (too small / failed).  cap = 0 asks for a bound. */
long pth_png_encode(const unsigned char *rgb, int width, int height, int flip, unsigned char *out, long cap) {
    if (cap == 0) return (long)compressBound((uLong)(((size_t)width * 3 + 1) * height)) + 64;
    std::vector<unsigned char> png;
    if (!EncodePNG(rgb, width, height, flip != 0, png) || (long)png.size() > cap) return -1;
    memcpy(out, png.data(), png.size());
    return (long)png.size();
}
/* The reference's tail of render() (main.cpp:1185): tone map the film of `ctx` and write `path`
 * (.ppm -> P6, anything else -> PNG).  0 on success. */
int pth_image_write(PtrContext *ctx, int width, int height, const char *path, float exposure, int algorithm) {
    Aggregator agg;
    agg.ctx = ctx;
    Image im;
    im.pixels = nullptr; im.width = width; im.height = height; im.pixels_count = width * height; im.owner = &agg;
    std::vector<unsigned char> data((size_t)im.pixels_count * 3);
    if (ptr_film_rgb8(ctx, exposure, algorithm, data.data())) return 1;
    size_t len = strlen(path);
    if (len > 4 && strcmp(path + len - 4, ".ppm") == 0) return ImageWritePPM(&im, path, exposure, (ToneMapAlgorithm)algorithm) ? 0 : 2;
    std::vector<unsigned char> png;
    if (!EncodePNG(data.data(), width, height, true, png)) return 2;
    remove(path);
    FILE *fp = fopen(path, "wb");
    if (!fp) return 3;
    size_t w = fwrite(png.data(), 1, png.size(), fp);
    return (fclose(fp) == 0 && w == png.size()) ? 0 : 3;
}
int pth_ppm_read(const char *path, unsigned char *out, long cap, int *width, int *height) {
    unsigned char *v = nullptr;
    if (!PPMRead(path, &v, *width, *height)) return 1;
    long n = (long)*width * *height * 3;
    int rc = 0;
    if (out) { if (n <= cap) memcpy(out, v, n); else rc = 2; }
    delete[] v;
    return rc;
}
}
