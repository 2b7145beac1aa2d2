/* Mutation fuzz driver of the host loaders: see tools/fuzz_loaders.sh (host ASan + UBSan build, no GPU). */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <unistd.h>
#include <sys/wait.h>
extern "C" {
void *pth_image_load(const char *path, int *w, int *h, const float **rgb);
void pth_image_free(void *h);
const float *pth_exr_read(const char *path, int *w, int *h);
void pth_exr_free(const float *p);
void *pth_obj_load(const char *path, int split);
int pth_obj_count(void *h);
void pth_obj_free(void *h);
void *pth_mtl_load(const char *file, const char *basedir, int times);
int pth_mtl_count(void *h);
void pth_mtl_free(void *h);
void *pth_particles_load(const char *path, const char *format);
void pth_particles_free(void *h);
}
static std::vector<unsigned char> slurp(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb"); std::vector<unsigned char> d(1 << 22); size_t n = fread(d.data(), 1, d.size(), f); fclose(f); d.resize(n); return d;
}
int main(int argc, char **argv) {
    int iters = atoi(argv[1]); long ok = 0;
    std::vector<std::string> bases;
    for (int i = 0; i < 11; i++) bases.push_back("e" + std::to_string(i) + ".exr");
    for (int i = 0; i < 4; i++) bases.push_back("p" + std::to_string(i) + ".png");
    for (int i = 0; i < 4; i++) bases.push_back("j" + std::to_string(i) + ".jpg");
    bases.push_back("o0.obj"); bases.push_back("things.mtl"); bases.push_back("t0.txt"); bases.push_back("q0.ppm");
    int bi = 0; int skip = argc > 2 ? atoi(argv[2]) : 0;
    for (const std::string &b : bases) {
        if (bi < skip) { bi++; continue; }
        std::vector<unsigned char> d = slurp("/tmp/lf/" + b);
        std::string ext = b.substr(b.find('.'));
        std::string mpath = "/tmp/lf/m/m" + ext;
        srand(++bi * 31);
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> m = d;
            int k = 1 + rand() % 5;
            for (int j = 0; j < k; j++) {
                int mode = rand() % 5; size_t pos = rand() % m.size();
                if (mode == 0) m[pos] = (unsigned char)rand();
                else if (mode == 1) m[pos] ^= 1 << (rand() % 8);
                else if (mode == 2) m.resize(pos + 1);
                else if (mode == 3) m[pos] = (rand() & 1) ? 0xff : 0;
                else { size_t len = 1 + rand() % 8; for (size_t q = 0; q < len && pos + q < m.size(); q++) m[pos + q] = "0123456789-e. /\n"[rand() % 16]; }
            }
            FILE *f = fopen(mpath.c_str(), "wb"); fwrite(m.data(), 1, m.size(), f); fclose(f);
            int w = 0, h = 0;
            if (ext == ".exr") { const float *p = pth_exr_read(mpath.c_str(), &w, &h); if (p) { ok++; volatile float s = p[(size_t)w * h * 3 - 1]; (void)s; pth_exr_free(p); } }
            else if (ext == ".png" || ext == ".ppm" || ext == ".jpg") { const float *rgb; void *hd = pth_image_load(mpath.c_str(), &w, &h, &rgb); if (hd) { ok++; volatile float s = rgb[(size_t)w * h * 3 - 1]; (void)s; pth_image_free(hd); } }
            else if (ext == ".obj" || ext == ".mtl" || ext == ".txt") { fflush(stdout); fflush(stderr); pid_t pid = fork(); if (pid == 0) {
            if (ext == ".obj") { for (int split = 0; split < 2; split++) { void *hd = pth_obj_load(mpath.c_str(), split); if (hd) { ok += pth_obj_count(hd) > 0; pth_obj_free(hd); } } }
            else if (ext == ".mtl") { void *hd = pth_mtl_load("m.mtl", "/tmp/lf/m/", 1); if (hd) { ok += pth_mtl_count(hd) > 0; pth_mtl_free(hd); } }
            else { void *hd = pth_particles_load(mpath.c_str(), "vs"); if (hd) { ok++; pth_particles_free(hd); } }
            _exit(0); } int st = 0; waitpid(pid, &st, 0); if (WIFSIGNALED(st) || (WIFEXITED(st) && WEXITSTATUS(st) > 1)) { fprintf(stderr, "child died: status %d on %s iteration %d\n", st, b.c_str(), it); return 2; } }
        }
        fprintf(stderr, "%s done, decoded so far %ld\n", b.c_str(), ok);
    }
}
