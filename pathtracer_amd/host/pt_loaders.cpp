/*
 * pt_loaders.cpp — OBJ / MTL / particle / texture-image ingest (see pt_loaders.h).  Written from the
 * behaviour of the reference's parsers, including the parts that matter for what ends up in the
 * scene: the number grammar of ParseDouble (mtl.cpp:41-154), first-use vertex compaction per
 * sub-mesh (obj_loader.cpp:172-258), the rule that any non-face line closes a sub-mesh
 * (obj_loader.cpp:320-323), quad fan order (obj_loader.cpp:413-419) and the token stepping of the
 * MTL value list (mtl.cpp:205-221).
 *
 * Argument evaluation order.  The reference reads a vector as
 *     Flip(vec3f(ParseFloat(t), ParseFloat(t), ParseFloat(t)))            (obj_loader.cpp:134-140)
 * and is built with g++, which evaluates the three calls right to left: the first number of the line
 * lands in z, Flip swaps it back, and "v 1 2 3" is (1, 2, 3).  Here the three reads are sequenced
 * explicitly, so the result does not depend on the compiler.
 */
#include "pt_loaders.h"
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <zlib.h>

/* ---- numbers ------------------------------------------------------------------------------------------ */
static inline bool is_digit(char c) { return (unsigned)(c - '0') < 10u; }

bool ParseDouble(const char *s, const char *s_end, Float *result) {
    if (s >= s_end) return false;
    const char *c = s;
    int sign = 1;
    if (*c == '+' || *c == '-') { sign = (*c == '-') ? -1 : 1; ++c; }
    else if (!is_digit(*c)) return false;

    double mantissa = 0.0;
    int ndigits = 0;
    for (; c != s_end && is_digit(*c); ++c, ++ndigits) { mantissa *= 10; mantissa += (int)(*c - '0'); }
    if (ndigits == 0) return false;

    int exponent = 0;
    bool have_exp = false;
    if (c != s_end) {
        if (*c == '.') {
            ++c;
            /* digit k after the point adds d * 10^-k: table up to 1e-7, pow beyond (mtl.cpp:101-108) */
            static const double tenth[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            for (int k = 1; c != s_end && is_digit(*c); ++c, ++k)
                mantissa += (int)(*c - '0') * (k < 8 ? tenth[k] : std::pow(10.0, -k));
            have_exp = c != s_end && (*c == 'e' || *c == 'E');
        } else if (*c == 'e' || *c == 'E') have_exp = true;
    }
    if (have_exp) {
        ++c;
        int esign = 1;
        if (c != s_end && (*c == '+' || *c == '-')) { esign = (*c == '-') ? -1 : 1; ++c; }
        else if (!is_digit(*c)) return false; /* (reads the terminator when the token ends in 'e', as the reference does) */
        int n = 0;
        for (; c != s_end && is_digit(*c); ++c, ++n) { exponent *= 10; exponent += (int)(*c - '0'); }
        exponent *= esign;
        if (n == 0) return false;
    }
    /* 10^e = 5^e * 2^e; the double is narrowed to Float on assignment */
    *result = sign * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

Float ParseFloat(const char **token) {
    (*token) += strspn(*token, " \t");
    const char *end = (*token) + strcspn(*token, " \t\r");
    Float val = 0;
    ParseDouble(*token, end, &val);
    *token = end;
    return val;
}
void ParseV3(vec3f *v, const char **token) {
    Float a = ParseFloat(token);
    Float b = ParseFloat(token);
    Float c = ParseFloat(token);
    *v = vec3f(a, b, c);
}
void ParseV2(vec2f *v, const char **token) {
    Float a = ParseFloat(token);
    Float b = ParseFloat(token);
    *v = vec2f(a, b);
}

/* a line ends at \n, \r\n or a lone \r; end of file ends the last line (obj_loader.cpp:105-122) */
std::istream &GetLine(std::istream &is, std::string &t) {
    t.clear();
    std::istream::sentry guard(is, true);
    if (!guard) return is;
    std::streambuf *sb = is.rdbuf();
    while (true) {
        int ch = sb->sbumpc();
        if (ch == '\n') break;
        if (ch == '\r') { if (sb->sgetc() == '\n') sb->sbumpc(); break; }
        if (ch == EOF) { if (t.empty()) is.setstate(std::ios::eofbit); break; }
        t.push_back((char)ch);
    }
    return is;
}
/* next non-empty, non-comment line, leading blanks skipped; false at end of file */
static bool next_payload_line(std::istream &is, std::string &buf, const char **token) {
    while (is.peek() != -1) {
        GetLine(is, buf);
        if (!buf.empty() && buf.back() == '\n') buf.pop_back();
        if (!buf.empty() && buf.back() == '\r') buf.pop_back();
        if (buf.empty()) continue;
        const char *t = buf.c_str();
        t += strspn(t, " \t");
        if (t[0] == '\0' || t[0] == '#') continue;
        *token = t;
        return true;
    }
    return false;
}
static inline bool is_blank(char c) { return c == ' ' || c == '\t'; }
static inline bool ends_line(char c) { return c == '\r' || c == '\n' || c == '\0'; }

/* ---- OBJ ------------------------------------------------------------------------------------------------ */
namespace {
struct Corner { int v, vt, vn; }; /* indices into the file-wide lists, -1 = absent */

/* 1-based -> 0-based, negative = relative to the current list end, 0 is invalid (obj_loader.cpp:26-47) */
bool resolve_index(int idx, int n, int *out) {
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx < 0) { *out = n + idx; return true; }
    return false;
}
/* "i", "i/j", "i//k", "i/j/k" (obj_loader.cpp:49-101) */
bool parse_corner(const char **tok, int nv, int nvn, int nvt, Corner *out) {
    Corner c = {-1, -1, -1};
    if (!resolve_index(atoi(*tok), nv, &c.v)) return false;
    *tok += strcspn(*tok, "/ \t\r");
    if ((*tok)[0] != '/') { *out = c; return true; }
    ++*tok;
    if ((*tok)[0] == '/') {
        ++*tok;
        if (!resolve_index(atoi(*tok), nvn, &c.vn)) return false;
        *tok += strcspn(*tok, "/ \t\r");
        *out = c;
        return true;
    }
    if (!resolve_index(atoi(*tok), nvt, &c.vt)) return false;
    *tok += strcspn(*tok, "/ \t\r");
    if ((*tok)[0] != '/') { *out = c; return true; }
    ++*tok;
    if (!resolve_index(atoi(*tok), nvn, &c.vn)) return false;
    *tok += strcspn(*tok, "/ \t\r");
    *out = c;
    return true;
}

template <typename T> T *clone_array(const std::vector<T> &v) {
    if (v.empty()) return nullptr;
    T *p = new T[v.size()];
    for (size_t i = 0; i < v.size(); i++) p[i] = v[i];
    return p;
}

/* One sub-mesh out of the corners gathered so far: positions / normals / uvs are renumbered in order of
 * first use, indices become (position, normal, uv) triples per corner (obj_loader.cpp:172-258). */
void close_mesh(ParsedMesh *mesh, const std::vector<vec3f> &v, const std::vector<vec3f> &vn, const std::vector<vec2f> &vt,
                const std::vector<Corner> &corners) {
    std::vector<int> newV(v.size(), -1), newN(vn.size(), -1), newU(vt.size(), -1);
    std::vector<Point3f> P;
    std::vector<Normal3f> N;
    std::vector<Point2f> U;
    std::vector<Point3i> I(corners.size());
    for (size_t i = 0; i < corners.size(); i++) {
        const Corner &c = corners[i];
        if (c.v < 0 || c.v >= (int)v.size() || c.vn >= (int)vn.size() || c.vt >= (int)vt.size()) {
            printf("OBJ: face refers to an element that is not in the file (the reference reads out of bounds here)\n");
            exit(1);
        }
        if (newV[c.v] < 0) { newV[c.v] = (int)P.size(); P.push_back(Point3f(v[c.v].x, v[c.v].y, v[c.v].z)); }
        if (c.vn > -1 && newN[c.vn] < 0) { newN[c.vn] = (int)N.size(); N.push_back(vn[c.vn]); }
        if (c.vt > -1 && newU[c.vt] < 0) { newU[c.vt] = (int)U.size(); U.push_back(Point2f(vt[c.vt].x, vt[c.vt].y)); }
        I[i] = Point3i(newV[c.v], c.vn > -1 ? newN[c.vn] : -1, c.vt > -1 ? newU[c.vt] : -1);
    }
    mesh->p = clone_array(P);
    mesh->n = clone_array(N);
    mesh->uv = clone_array(U);
    mesh->indices = clone_array(I);
    mesh->nTriangles = (int)(corners.size() / 3);
    mesh->nVertices = (int)P.size();
    mesh->nNormals = (int)N.size();
    mesh->nUvs = (int)U.size();
}

ParsedMesh *blank_mesh() {
    ParsedMesh *m = new ParsedMesh();
    m->p = nullptr; m->n = nullptr; m->s = nullptr; m->uv = nullptr; m->indices = nullptr;
    m->nTriangles = m->nVertices = m->nUvs = m->nNormals = 0;
    m->toWorld = Translate(0, 0, 0);
    return m;
}
} // namespace

void FreeParsedMesh(ParsedMesh *m) {
    if (!m) return;
    delete[] m->p; delete[] m->n; delete[] m->s; delete[] m->uv; delete[] m->indices;
    delete m;
}

ParsedMesh *LoadObjOnly(const char *path) {
    std::vector<MeshMtl> mtls;
    std::vector<ParsedMesh *> *meshes = LoadObj(path, &mtls, false);
    ParsedMesh *mesh = meshes->empty() ? nullptr : meshes->at(0); /* the reference's at(0) throws on a file without faces */
    delete meshes;
    return mesh;
}

std::vector<ParsedMesh *> *LoadObj(const char *path, std::vector<MeshMtl> *mtls, bool split_mesh) {
    std::vector<ParsedMesh *> *meshes = new std::vector<ParsedMesh *>();
    printf("Attempting to parse %s\n", path);
    std::ifstream ifs(path);
    if (!ifs) {
        printf("Could not open file %s\n", path);
        return meshes;
    }
    std::vector<vec3f> v, vn;
    std::vector<vec2f> vt;
    std::vector<Corner> corners;
    ParsedMesh *open_mesh = nullptr; /* sub-mesh being gathered */
    std::string mtl_file, mtl_name, line;
    const char *tok = nullptr;
    clock_t start = clock();

    while (next_payload_line(ifs, line, &tok)) {
        /* any line that does not start with 'f' ends the sub-mesh (the test is on the first character only) */
        if (tok[0] != 'f' && open_mesh && split_mesh) {
            close_mesh(open_mesh, v, vn, vt, corners);
            open_mesh = nullptr;
        }
        /* "o name": the name is skipped and the rest of the line goes through the same tests, so an object
         * called "v 1 2 3" is also a vertex (obj_loader.cpp:325-331 has no `continue`) */
        if (tok[0] == 'o' && is_blank(tok[1])) tok += 2;
        const bool is_face = tok[0] == 'f' && is_blank(tok[1]);
        if (tok[0] == 'v' && is_blank(tok[1])) {
            tok += 2;
            vec3f p;
            ParseV3(&p, &tok);
            v.push_back(p);
        } else if (tok[0] == 'v' && tok[1] == 'n' && is_blank(tok[2])) {
            tok += 3;
            vec3f n;
            ParseV3(&n, &tok);
            vn.push_back(n);
        } else if (tok[0] == 'v' && tok[1] == 't' && is_blank(tok[2])) {
            tok += 3;
            vec2f uv;
            ParseV2(&uv, &tok);
            vt.push_back(uv);
        } else if (strncmp(tok, "mtllib", 6) == 0 && is_blank(tok[6])) {
            mtl_file = std::string(tok + 7);
        } else if (strncmp(tok, "usemtl", 6) == 0 && is_blank(tok[6])) {
            mtl_name = std::string(tok + 7);
        } else if (is_face) {
            if (!open_mesh) {
                open_mesh = blank_mesh();
                meshes->push_back(open_mesh);
                MeshMtl m;
                m.file = mtl_file;
                m.name = mtl_name;
                mtls->push_back(m);
                corners.clear();
            }
            tok += 2;
            tok += strspn(tok, " \t");
            Corner face[4];
            int nc = 0;
            while (!ends_line(tok[0])) {
                Corner c;
                if (!parse_corner(&tok, (int)v.size(), (int)vn.size(), (int)vt.size(), &c)) {
                    printf("Failed parsing face\n");
                    break;
                }
                tok += strspn(tok, " \t\r");
                if (nc >= 4) {
                    printf("Error: Not a supported face description\n");
                    exit(0);
                }
                face[nc++] = c;
            }
            if (nc == 3) {
                corners.insert(corners.end(), face, face + 3);
            } else if (nc == 4) { /* fan: (0 1 2) (0 2 3) */
                const Corner quad[6] = {face[0], face[1], face[2], face[0], face[2], face[3]};
                corners.insert(corners.end(), quad, quad + 6);
            } else {
                printf("Warning unsupported face with %d vertices\n", nc);
            }
        }
        /* 'o', 'g', 's' and anything else: no state of their own */
    }
    if (open_mesh) close_mesh(open_mesh, v, vn, vt, corners);
    printf("Took %g seconds, #v [%d] #vn [%d] #vt [%d]. Decomposed in %d meshe(s)\n", (double)(clock() - start) / CLOCKS_PER_SEC, (int)v.size(),
           (int)vn.size(), (int)vt.size(), (int)meshes->size());
    return meshes;
}

/* obj_loader.cpp:445-521: deep copy; flip_x mirrors positions about the vertex centroid */
ParsedMesh *DuplicateMesh(ParsedMesh *mesh, MeshProperties *props) {
    if (!mesh) return nullptr;
    ParsedMesh *d = blank_mesh();
    d->toWorld = mesh->toWorld;
    d->nVertices = mesh->nVertices; d->nTriangles = mesh->nTriangles; d->nUvs = mesh->nUvs; d->nNormals = mesh->nNormals;
    vec3f center;
    if (mesh->p) {
        Float w = 0.f;
        for (int i = 0; i < mesh->nVertices; i++) {
            center = center + vec3f(mesh->p[i].x, mesh->p[i].y, mesh->p[i].z);
            w += 1.f;
        }
        Float invW = 1.f / w;
        center = center * invW;
        d->p = new Point3f[mesh->nVertices];
        for (int i = 0; i < mesh->nVertices; i++) {
            Point3f p = mesh->p[i];
            if (props && props->flip_x) {
                vec3f op = vec3f(p.x, p.y, p.z) - center;
                vec3f wp(-op.x, op.y, op.z);
                p = Point3f(wp.x + center.x, wp.y + center.y, wp.z + center.z);
            }
            d->p[i] = p;
        }
    }
    if (mesh->uv) { d->uv = new Point2f[mesh->nUvs]; for (int i = 0; i < mesh->nUvs; i++) d->uv[i] = mesh->uv[i]; }
    if (mesh->n) { d->n = new Normal3f[mesh->nNormals]; for (int i = 0; i < mesh->nNormals; i++) d->n[i] = mesh->n[i]; }
    if (mesh->indices) {
        d->indices = new Point3i[(size_t)mesh->nTriangles * 3];
        for (int i = 0; i < mesh->nTriangles * 3; i++) d->indices[i] = mesh->indices[i];
    }
    return d;
}

/* ---- MTL ------------------------------------------------------------------------------------------------ */
namespace {
const char *const kMtlKeys[13] = {"Ka", "Kd", "Ks", "Tf", "Ns", "illum", "map_Ka", "map_Kd", "map_Ks", "d", "Ke", "Ni", "map_Bump"};
const bool kMtlKeyNumeric[13] = {true, true, true, true, true, true, false, false, false, true, true, true, false};

bool starts_with(const std::string &s, const char *prefix) { return s.compare(0, strlen(prefix), prefix) == 0; }
inline char at(const std::string &s, size_t i) { return i < s.size() ? s[i] : '\0'; }
inline bool is_sep(char c) { return c == ' ' || c == '\n' || c == '\r'; }

/* the word that starts at or after `pos`; *last when nothing is left (mtl.cpp:162-190) */
std::string word_from(const std::string &s, size_t pos, bool *last) {
    *last = true;
    while (pos < s.size() && is_sep(s[pos])) pos++;
    if (pos >= s.size()) return std::string();
    size_t end = pos;
    while (end < s.size() && !is_sep(s[end])) end++;
    char stop = at(s, end);
    *last = (stop == '\n' || stop == '\r');
    return s.substr(pos, end - pos);
}
/* Values of "key v0 v1 ...".  The cursor advances by len(value) + 1 from the end of the key, not from
 * where the value was found (mtl.cpp:214-220): with single blanks that is the next value, with runs of
 * blanks or tabs the following reads start inside the previous value.  Kept, since it decides what a
 * given file turns into. */
std::vector<std::string> key_values(const std::string &line, size_t key_len) {
    std::vector<std::string> out;
    size_t cur = key_len;
    bool last = false;
    while (!last) {
        std::string w = word_from(line, cur, &last);
        if (!w.empty()) { out.push_back(w); cur += w.size() + 1; }
    }
    return out;
}
} // namespace

bool MTLParse(const char *path, std::vector<MTL *> *materials, const char *basedir) {
    if (!materials) {
        printf("Invalid materials pointer for MTL parsing\n");
        return false;
    }
    std::string filename = std::string(basedir) + std::string(path);
    std::ifstream in(filename.c_str());
    if (!in.is_open()) {
        printf("Could not open file: %s\n", path);
        return false;
    }
    MTL *cur = nullptr;
    bool in_material = false;
    std::string raw;
    while (std::getline(in, raw)) { /* '\n' only: a lone '\r' does not end a line here, unlike the OBJ reader */
        if (!raw.empty() && raw.back() == '\n') raw.pop_back();
        if (!raw.empty() && raw.back() == '\r') raw.pop_back();
        if (raw.empty()) continue;
        const char *t = raw.c_str();
        t += strspn(t, " \t");
        if (t[0] == '#') continue;
        std::string line(t);
        if (line.empty()) continue;
        if (starts_with(line, "newmtl")) {
            bool last;
            cur = new MTL;
            cur->mtlname = word_from(line, 6, &last);
            cur->basedir = std::string(basedir);
            materials->push_back(cur);
            in_material = true;
        } else if (in_material) {
            int k = 0;
            while (k < 13 && !starts_with(line, kMtlKeys[k])) k++; /* first key that prefixes the line, in table order */
            if (k == 13) {
                printf("Could not find token for line: %s\n", line.c_str());
                in_material = false; /* properties are ignored until the next newmtl */
            } else {
                cur->data[kMtlKeys[k]] = key_values(line, strlen(kMtlKeys[k]));
            }
        }
    }
    return true;
}

std::string MTLGetValue(const char *property, MTL *mtl, bool &found) {
    std::string out;
    found = false;
    if (!mtl) return out;
    auto it = mtl->data.find(property);
    if (it == mtl->data.end()) return out;
    found = true;
    for (size_t i = 0; i < it->second.size(); i++) {
        if (i) out += " ";
        out += it->second[i];
    }
    return out;
}

Spectrum MTLGetSpectrum(const char *property, MTL *mtl, bool &found) {
    Spectrum e(0);
    found = false;
    if (!mtl) return e;
    auto it = mtl->data.find(property);
    if (it == mtl->data.end()) return e;
    found = true;
    bool numeric = false;
    for (int k = 0; k < 13; k++)
        if (strcmp(property, kMtlKeys[k]) == 0) numeric = kMtlKeyNumeric[k];
    if (!numeric) {
        printf("Warning: Query for non numeric property (%s)\n", property);
        return e;
    }
    const std::vector<std::string> &vals = it->second;
    for (size_t c = 0; c < 3 && c < vals.size(); c++) { /* stops at the first value that is not a number */
        Float x = 0;
        if (!ParseDouble(vals[c].c_str(), vals[c].c_str() + vals[c].size(), &x)) break;
        e[(int)c] = x;
    }
    return e;
}

MTL *MTLFindMaterial(const char *name, std::vector<MTL *> *materials) {
    for (MTL *m : *materials)
        if (m->mtlname == name) return m;
    printf("Warning: Did not find material (%s)\n", name);
    return materials->at(0);
}

/* one parse per sub-mesh descriptor, so a library used by n sub-meshes is listed n times; lookups take
 * the first entry of a name (mtl.cpp:366-379) */
bool MTLParseAll(std::vector<MTL *> *materials, std::vector<MeshMtl> *desc, const char *basedir) {
    bool ok = true;
    if (desc)
        for (size_t i = 0; i < desc->size() && ok; i++) ok &= MTLParse((*desc)[i].file.c_str(), materials, basedir);
    return ok;
}

/* scene.cpp:484-577: Ke -> light; Kd + Ks (maps win over colours) -> plastic, roughness 0.03; one of the
 * two -> matte when it was Ks, the "broken uber" otherwise; neither -> grey matte */
MaterialDescriptor MakeMTLMaterial(MTL *mtl) {
    bool found = false;
    Spectrum L = MTLGetSpectrum("Ke", mtl, found);
    const bool black = L.x == 0 && L.y == 0 && L.z == 0;
    if (!black && found) return MakeEmissive(L);

    MaterialDescriptor desc;
    memset((void *)&desc, 0, sizeof(desc));
    bool hasKd = false, hasKdMap = false, hasKs = false, hasKsMap = false;
    Spectrum kd = MTLGetSpectrum("Kd", mtl, hasKd);
    std::string kdMap = MTLGetValue("map_Kd", mtl, hasKdMap);
    Spectrum ks = MTLGetSpectrum("Ks", mtl, hasKs);
    std::string ksMap = MTLGetValue("map_Ks", mtl, hasKsMap);
    int i = 0;
    bool kd_colour = hasKd, ks_colour = hasKs;
    if (hasKdMap) {
        ImageData *img = LoadTextureImageData((mtl->basedir + kdMap).c_str());
        if (img) { desc.textures[i++] = MakeTexture(img); kd_colour = false; printf(" * Got Kd map\n"); }
        else hasKdMap = false;
    }
    if (kd_colour) desc.textures[i++] = MakeTexture(kd);
    if (hasKsMap) {
        ImageData *img = LoadTextureImageData((mtl->basedir + ksMap).c_str());
        if (img) { desc.textures[i++] = MakeTexture(img); ks_colour = false; printf(" * Got Ks map\n"); }
        else hasKsMap = false;
    }
    if (ks_colour) desc.textures[i++] = MakeTexture(ks);

    if (i == 2) {
        Float rough = 0.03;
        desc.textures[i++] = MakeTexture(rough);
        desc.type = MaterialType::Plastic;
    } else if (i == 1) {
        if (hasKs || hasKsMap) {
            Float sigma = 0;
            desc.textures[i++] = MakeTexture(sigma);
            desc.type = MaterialType::Matte;
        } else {
            printf("Warning: broken uber material\n");
            desc = MakeUberMaterial(Spectrum(0.001), ks, Spectrum(0), Spectrum(0), 0, 0, Spectrum(1), 1.5f);
        }
    } else {
        printf("Warning: Could not make material for MTL\n");
        desc.textures[i++] = MakeTexture((Float)0.45);
        desc.textures[i++] = MakeTexture((Float)0.);
        desc.type = MaterialType::Matte;
    }
    if (desc.specs_taken == 0) desc.specs_taken = i; /* slots filled above (the uber fallback brings its own count) */
    return desc;
}

/* ---- texture images ------------------------------------------------------------------------------------- */
namespace {
bool read_file(const char *path, std::vector<unsigned char> &out) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(fp);
    return true;
}
uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* 8-bit, non-interlaced PNG -> rows top-down, `nn` interleaved channels (what stbi_load(..., 0) returns) */
bool decode_png(const std::vector<unsigned char> &f, std::vector<unsigned char> &px, int &w, int &h, int &nn) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || memcmp(f.data(), sig, 8) != 0) return false;
    std::vector<unsigned char> idat, plte;
    int ctype = -1;
    size_t at = 8;
    bool end = false;
    while (!end && at + 12 <= f.size()) {
        uint32_t len = be32(&f[at]);
        const unsigned char *tag = &f[at + 4], *body = &f[at + 8];
        if (at + 12 + (size_t)len > f.size()) return false;
        if (crc32(0L, tag, (uInt)(len + 4)) != be32(body + len)) return false;
        if (memcmp(tag, "IHDR", 4) == 0) {
            if (len != 13) return false;
            w = (int)be32(body); h = (int)be32(body + 4);
            ctype = body[9];
            if (body[8] != 8 || body[10] != 0 || body[11] != 0 || body[12] != 0) return false; /* depth 8, no interlace */
        } else if (memcmp(tag, "PLTE", 4) == 0) plte.assign(body, body + len);
        else if (memcmp(tag, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
        else if (memcmp(tag, "IEND", 4) == 0) end = true;
        at += 12 + (size_t)len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!end || ch == 0 || w <= 0 || h <= 0 || (int64_t)w * h > (1 << 28)) return false;
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return false;
    std::vector<unsigned char> img(stride * h);
    for (int y = 0; y < h; y++) { /* undo the scanline filters */
        const unsigned char *in = &raw[(stride + 1) * y + 1];
        unsigned char *cur = &img[stride * y];
        const unsigned char *up = y ? &img[stride * (y - 1)] : nullptr;
        const int ft = raw[(stride + 1) * y];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0, pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return false;
            cur[x] = (unsigned char)(in[x] + pred);
        }
    }
    if (ctype == 3) { /* palette -> RGB */
        px.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            size_t k = (size_t)img[i] * 3;
            if (k + 3 > plte.size()) return false;
            px[3 * i] = plte[k]; px[3 * i + 1] = plte[k + 1]; px[3 * i + 2] = plte[k + 2];
        }
        nn = 3;
    } else {
        px.swap(img);
        nn = ch;
    }
    return true;
}

/* binary PGM (P5) / PPM (P6), maxval 255, rows top-down as stored */
bool decode_pnm(const std::vector<unsigned char> &f, std::vector<unsigned char> &px, int &w, int &h, int &nn) {
    if (f.size() < 7 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return false;
    nn = f[1] == '6' ? 3 : 1;
    size_t at = 2;
    int vals[3], got = 0;
    while (got < 3 && at < f.size()) {
        if (f[at] == '#') { while (at < f.size() && f[at] != '\n') at++; continue; }
        if (isspace(f[at])) { at++; continue; }
        if (!isdigit(f[at])) return false;
        int x = 0;
        while (at < f.size() && isdigit(f[at])) { /* header values are small: stop growing long before an int could overflow */
            if (x < 100000000) x = x * 10 + (f[at] - '0');
            at++;
        }
        vals[got++] = x;
    }
    if (got < 3 || vals[2] != 255 || at >= f.size()) return false;
    at++; /* the single whitespace byte after maxval */
    w = vals[0]; h = vals[1];
    if (w <= 0 || h <= 0 || (int64_t)w * h > (1 << 28)) return false;
    size_t need = (size_t)w * h * nn;
    if (f.size() - at < need) return false;
    px.assign(f.begin() + at, f.begin() + at + need);
    return true;
}

/* ---- JPEG (ITU T.81 baseline and progressive Huffman, 8 bit, 1 or 3 components) -----------------------------
 * Replaces stbi_load(path, &nx, &ny, &nn, 0) of the stb_image the reference vendors (src/third/stb_image.h, v2.0x:
 * jpeg part :1460-3400) for the two .jpg textures its scenes name (main.cpp:633,635).  A JPEG's pixels are not
 * defined by the standard to the last bit: they depend on the decoder's inverse DCT, chroma upsampling and colour
 * conversion.  This decoder is written from the standard, with those three stages in the fixed-point arithmetic
 * stb_image documents for itself - inverse DCT after jidctint's "islow" factorisation with 12-bit constants and two
 * guard bits between the passes (:1905-2003), chroma rows / columns interpolated 3:1 for the 2x factors and repeated
 * for the others (:2848-3038), YCbCr -> RGB with 12-bit constants in a 20-bit fixed point (:3071-3095) - so that a
 * texture decodes to the bytes the reference renders with.  PARITY UNPINNED: the reference ships neither a .jpg nor
 * decoded pixels; tests/test_cpu_jpeg.py holds the decoder to Pillow's (libjpeg) within the +-few levels two
 * correct decoders differ by, and to exact known answers where the arithmetic leaves no freedom.
 * Like stb_image's: no arithmetic coding, no 12-bit, no CMYK / 4 components, no 16-bit quantisation tables. */
struct JpegHuff {
    /* canonical code (T.81 Annex C / F.2.2.3): for each length the first code, its index into `sym`, and the count */
    int first_code[17], first_sym[17], count[17];
    unsigned char sym[256];
    bool defined = false;
    bool build(const int *counts, const unsigned char *symbols, int n) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            first_code[len] = code;
            first_sym[len] = k;
            count[len] = counts[len - 1];
            code += counts[len - 1];
            k += counts[len - 1];
            if (code > (1 << len)) return false; /* more codes of this length than bits allow */
            code <<= 1;
        }
        if (n > 256) return false;
        memcpy(sym, symbols, (size_t)n);
        defined = true;
        return true;
    }
};
/* entropy-coded segment: bits most significant first, FF00 -> FF, any other FFxx is a marker: it ends the data and
 * zeros are fed from there on (what stb_image does, :1566-1581) */
struct JpegBits {
    const unsigned char *p, *end;
    uint32_t acc = 0;
    int n = 0;
    int marker = -1;
    JpegBits(const unsigned char *b, const unsigned char *e) : p(b), end(e) {}
    void reset() { acc = 0; n = 0; marker = -1; }
    void fill() {
        while (n <= 24) {
            int b = 0;
            if (marker < 0 && p < end) {
                b = *p++;
                if (b == 0xff) {
                    int c = p < end ? *p++ : 0;
                    if (c != 0) { marker = c; b = 0; }
                }
            }
            if (marker >= 0) b = 0;
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int bits(int k) { /* k <= 16 */
        if (k == 0) return 0;
        if (n < k) fill();
        int v = (int)(acc >> (32 - k));
        acc <<= k;
        n -= k;
        return v;
    }
    int bit() { return bits(1); }
    /* receive + extend (T.81 F.2.2.1): k magnitude bits -> signed value */
    int receive_extend(int k) {
        if (k == 0) return 0;
        int v = bits(k);
        return v < (1 << (k - 1)) ? v - (1 << k) + 1 : v;
    }
    int decode(const JpegHuff &h) { /* -1: no such code */
        if (n < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (int)(acc >> (32 - len));
            int off = code - h.first_code[len];
            if (off >= 0 && off < h.count[len]) {
                acc <<= len;
                n -= len;
                return h.sym[h.first_sym[len] + off];
            }
        }
        return -1;
    }
};
struct JpegComp {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int px = 0, py = 0;     /* samples this component really has */
    int w2 = 0, h2 = 0;     /* plane size: whole MCUs */
    int bw = 0;             /* blocks per plane row */
    int dc_pred = 0;
    std::vector<unsigned char> plane;
    std::vector<int16_t> coef; /* progressive: 64 per block, natural order */
};
/* position k of the zig-zag sequence -> row-major index; built once (T.81 figure A.6); beyond 63 a corrupt stream lands on 63 */
struct JpegZigzag {
    unsigned char at[64 + 16];
    JpegZigzag() {
        int r = 0, c = 0;
        for (int k = 0; k < 64; k++) {
            at[k] = (unsigned char)(r * 8 + c);
            if ((r + c) % 2 == 0) { /* moving up-right */
                if (c == 7) r++;
                else if (r == 0) c++;
                else { r--; c++; }
            } else { /* moving down-left */
                if (r == 7) c++;
                else if (c == 0) r++;
                else { r++; c--; }
            }
        }
        for (int k = 64; k < 80; k++) at[k] = 63;
    }
};
const JpegZigzag &jpeg_zigzag() { static JpegZigzag z; return z; }

/* one 8-point inverse DCT pass, jidctint's factorisation; constants are round(x * 4096).  in[] are eight samples, the
 * result is the even part e[0..3] and the odd part o[0..3], both scaled by 4096: out[i] = e[i] + o[3-i], out[7-i] = e[i] - o[3-i].
 * 32-bit two's-complement arithmetic that wraps (unsigned here, so that a corrupt file's huge coefficients are defined too) */
typedef uint32_t ju32;
inline int jpeg_sar(ju32 x, int n) { return (int)((int32_t)x >> n); }
inline void jpeg_idct8(const int *in, ju32 *e, ju32 *o) {
    const ju32 K0_541 = 2217, K1_847 = (ju32)-7567, K0_765 = 3135, K1_175 = 4816, K0_298 = 1223, K2_053 = 8410, K3_072 = 12586, K1_501 = 6149,
               K0_899 = (ju32)-3685, K2_562 = (ju32)-10497, K1_961 = (ju32)-8034, K0_390 = (ju32)-1597;
    const ju32 s0 = (ju32)in[0], s1 = (ju32)in[1], s2 = (ju32)in[2], s3 = (ju32)in[3], s4 = (ju32)in[4], s5 = (ju32)in[5], s6 = (ju32)in[6], s7 = (ju32)in[7];
    const ju32 z = (s2 + s6) * K0_541;
    const ju32 a2 = z + s6 * K1_847, a3 = z + s2 * K0_765;
    const ju32 a0 = (s0 + s4) * 4096u, a1 = (s0 - s4) * 4096u;
    e[0] = a0 + a3; e[3] = a0 - a3; e[1] = a1 + a2; e[2] = a1 - a2;
    const ju32 q3 = s7 + s3, q4 = s5 + s1, q1 = s7 + s1, q2 = s5 + s3;
    const ju32 q5 = (q3 + q4) * K1_175;
    const ju32 m1 = q5 + q1 * K0_899, m2 = q5 + q2 * K2_562, m3 = q3 * K1_961, m4 = q4 * K0_390;
    o[3] = s1 * K1_501 + m1 + m4;
    o[2] = s3 * K3_072 + m2 + m3;
    o[1] = s5 * K2_053 + m2 + m4;
    o[0] = s7 * K0_298 + m1 + m3;
}
inline unsigned char jpeg_clamp(int x) { return (unsigned char)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
/* dequantised coefficients (row-major) -> 8x8 samples.  Columns first, kept with two extra bits; a column whose AC
 * terms are all zero is its DC term times four (exactly what the full formula gives); then rows, rounded, +128, clamped */
void jpeg_idct_block(unsigned char *out, int out_stride, const int16_t *d) {
    int tmp[64];
    for (int c = 0; c < 8; c++) {
        int col[8];
        bool ac = false;
        for (int r = 0; r < 8; r++) { col[r] = d[r * 8 + c]; if (r && col[r]) ac = true; }
        if (!ac) {
            for (int r = 0; r < 8; r++) tmp[r * 8 + c] = col[0] * 4;
            continue;
        }
        ju32 e[4], o[4];
        jpeg_idct8(col, e, o);
        for (int i = 0; i < 4; i++) {
            tmp[i * 8 + c] = jpeg_sar(e[i] + 512u + o[3 - i], 10);
            tmp[(7 - i) * 8 + c] = jpeg_sar(e[i] + 512u - o[3 - i], 10);
        }
    }
    for (int r = 0; r < 8; r++) {
        ju32 e[4], o[4];
        jpeg_idct8(tmp + r * 8, e, o);
        unsigned char *row = out + (size_t)r * out_stride;
        const ju32 bias = 65536u + (128u << 17);
        for (int i = 0; i < 4; i++) {
            row[i] = jpeg_clamp(jpeg_sar(e[i] + bias + o[3 - i], 17));
            row[7 - i] = jpeg_clamp(jpeg_sar(e[i] + bias - o[3 - i], 17));
        }
    }
}

struct JpegDecoder {
    const std::vector<unsigned char> &f;
    size_t at = 0;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    int restart_interval = 0;
    JpegComp comp[3];
    unsigned char quant[4][64];
    JpegHuff hdc[4], hac[4];
    /* the scan being decoded */
    int scan_n = 0, order[3] = {0, 0, 0}, ss = 0, se = 63, ah = 0, al = 0, eob_run = 0;

    explicit JpegDecoder(const std::vector<unsigned char> &file) : f(file) { memset(quant, 0, sizeof quant); }
    bool have(size_t n) const { return at + n <= f.size(); }
    int u8() { return at < f.size() ? f[at++] : 0; }
    int u16() { int a = u8(); return (a << 8) | u8(); }
    /* next marker code, skipping fill bytes; -1 at the end of the file or when no FF is where one must be */
    int next_marker() {
        if (!have(2) || f[at] != 0xff) return -1;
        while (at < f.size() && f[at] == 0xff) at++;
        return at < f.size() ? f[at++] : -1;
    }
    bool read_dqt() {
        int len = u16() - 2;
        while (len > 0) {
            int q = u8();
            if ((q >> 4) != 0 || (q & 15) > 3) return false; /* 8-bit tables only */
            for (int i = 0; i < 64; i++) quant[q & 15][jpeg_zigzag().at[i]] = (unsigned char)u8();
            len -= 65;
        }
        return len == 0;
    }
    bool read_dht() {
        int len = u16() - 2;
        while (len > 0) {
            int q = u8(), counts[16], n = 0;
            if ((q >> 4) > 1 || (q & 15) > 3) return false;
            for (int i = 0; i < 16; i++) { counts[i] = u8(); n += counts[i]; }
            if (n > 256 || !have((size_t)n)) return false;
            unsigned char symbols[256];
            for (int i = 0; i < n; i++) symbols[i] = (unsigned char)u8();
            JpegHuff &h = (q >> 4) ? hac[q & 15] : hdc[q & 15];
            if (!h.build(counts, symbols, n)) return false;
            len -= 17 + n;
        }
        return len == 0;
    }
    bool read_segment(int m) { /* tables and ignorable segments between the frame / scan headers */
        if (m == 0xdb) return read_dqt();
        if (m == 0xc4) return read_dht();
        if (m == 0xdd) {
            if (u16() != 4) return false;
            restart_interval = u16();
            return true;
        }
        if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
            int len = u16();
            if (len < 2 || !have((size_t)len - 2)) return false;
            at += (size_t)len - 2;
            return true;
        }
        return false;
    }
    bool read_frame() {
        int len = u16();
        if (len < 11 || u8() != 8) return false;
        height = u16();
        width = u16();
        ncomp = u8();
        if (height == 0 || width == 0 || (ncomp != 1 && ncomp != 3) || len != 8 + 3 * ncomp) return false;
        if ((int64_t)width * height > (1 << 28)) return false; /* the PNG route's limit */
        for (int i = 0; i < ncomp; i++) {
            JpegComp &c = comp[i];
            c.id = u8();
            if (c.id != i + 1 && c.id != i) return false; /* JFIF numbering, or the from-zero numbering some writers use */
            int q = u8();
            c.h = q >> 4; c.v = q & 15; c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        mcux = (width + hmax * 8 - 1) / (hmax * 8);
        mcuy = (height + vmax * 8 - 1) / (vmax * 8);
        for (int i = 0; i < ncomp; i++) {
            JpegComp &c = comp[i];
            c.px = (width * c.h + hmax - 1) / hmax;
            c.py = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcux * c.h * 8;
            c.h2 = mcuy * c.v * 8;
            c.bw = c.w2 / 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coef.assign((size_t)c.w2 * c.h2, 0);
        }
        return true;
    }
    bool read_scan_header() {
        int len = u16();
        scan_n = u8();
        if (scan_n < 1 || scan_n > ncomp || len != 6 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; i++) {
            int id = u8(), q = u8(), which = 0;
            while (which < ncomp && comp[which].id != id) which++;
            if (which == ncomp) return false;
            comp[which].hd = q >> 4; comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return false;
            order[i] = which;
        }
        ss = u8(); se = u8();
        int a = u8();
        ah = a >> 4; al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return false;
        } else {
            if (ss != 0 || ah != 0 || al != 0) return false;
            se = 63;
        }
        return true;
    }
    /* sequential block: DC difference, then (run, size) pairs; coefficients leave dequantised, in 16 bits */
    bool block_sequential(JpegBits &br, JpegComp &c, int16_t *d) {
        const unsigned char *zz = jpeg_zigzag().at, *q = quant[c.tq];
        const JpegHuff &dc = hdc[c.hd], &ac = hac[c.ha];
        memset(d, 0, 64 * sizeof(int16_t));
        int t = br.decode(dc);
        if (t < 0 || t > 16) return false;
        c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)br.receive_extend(t)); /* wraps on a corrupt file instead of overflowing */
        d[0] = (int16_t)((unsigned)c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            int rs = br.decode(ac);
            if (rs < 0) return false;
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                int pos = zz[k++];
                d[pos] = (int16_t)(br.receive_extend(s) * q[pos]);
            }
        }
        return true;
    }
    bool block_prog_dc(JpegBits &br, JpegComp &c, int16_t *d) {
        if (se != 0) return false; /* DC scans carry no AC */
        if (ah == 0) {
            memset(d, 0, 64 * sizeof(int16_t));
            int t = br.decode(hdc[c.hd]);
            if (t < 0 || t > 16) return false;
            c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)br.receive_extend(t));
            d[0] = (int16_t)((unsigned)c.dc_pred << al);
        } else if (br.bit()) d[0] = (int16_t)(d[0] + (1 << al));
        return true;
    }
    /* a refinement bit for a coefficient that is already non-zero (T.81 G.1.2.3) */
    static void refine(JpegBits &br, int16_t *p, int bitv) {
        if (br.bit() && (*p & bitv) == 0) *p = (int16_t)(*p > 0 ? *p + bitv : *p - bitv);
    }
    bool block_prog_ac(JpegBits &br, JpegComp &c, int16_t *d) {
        if (ss == 0) return false;
        const unsigned char *zz = jpeg_zigzag().at;
        const JpegHuff &ac = hac[c.ha];
        if (ah == 0) { /* first pass over this band */
            if (eob_run) { eob_run--; return true; }
            for (int k = ss; k <= se;) {
                int rs = br.decode(ac);
                if (rs < 0) return false;
                int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { /* EOBn: this block and 2^r + bits - 1 more end here */
                        eob_run = (1 << r) - 1;
                        if (r) eob_run += br.bits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    d[zz[k++]] = (int16_t)((unsigned)br.receive_extend(s) << al);
                }
            }
            return true;
        }
        const int bitv = 1 << al;
        if (eob_run) {
            eob_run--;
            for (int k = ss; k <= se; k++) {
                int16_t *p = &d[zz[k]];
                if (*p != 0) refine(br, p, bitv);
            }
            return true;
        }
        /* (run, 1) pairs: each places one new coefficient of magnitude 1 << al after `run` still-zero positions; every
         * non-zero coefficient passed on the way takes a correction bit.  EOBn ends the new coefficients of this block and of
         * n - 1 blocks after it: what is left of the band only takes correction bits. */
        for (int k = ss; k <= se;) {
            const int rs = br.decode(ac);
            if (rs < 0) return false;
            const int run = rs >> 4, size = rs & 15;
            if (size == 0 && run < 15) {
                eob_run = (1 << run) - 1;
                if (run) eob_run += br.bits(run);
                for (; k <= se; k++) {
                    int16_t *p = &d[zz[k]];
                    if (*p != 0) refine(br, p, bitv);
                }
                return true;
            }
            if (size > 1) return false;
            /* size 1: a new coefficient, its sign comes first in the stream; size 0 with run 15 (ZRL): sixteen zero positions */
            const int value = size ? (br.bit() ? bitv : -bitv) : 0;
            int zeros_to_pass = run;
            while (k <= se) {
                int16_t *p = &d[zz[k++]];
                if (*p != 0) { refine(br, p, bitv); continue; }
                if (zeros_to_pass == 0) { *p = (int16_t)value; break; } /* (ZRL writes the zero that is there already) */
                zeros_to_pass--;
            }
        }
        return true;
    }
    bool decode_scan() {
        JpegBits br(f.data() + at, f.data() + f.size());
        for (int i = 0; i < ncomp; i++) comp[i].dc_pred = 0;
        eob_run = 0;
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        bool stop = false;
        /* after each MCU: at a restart interval's end the next marker must be RSTn, else the scan stops there */
        auto mcu_done = [&]() {
            if (--todo > 0) return;
            if (br.n < 24) br.fill();
            if (br.marker < 0xd0 || br.marker > 0xd7) { stop = true; return; }
            br.reset();
            for (int i = 0; i < ncomp; i++) comp[i].dc_pred = 0;
            eob_run = 0;
            todo = restart_interval ? restart_interval : 0x7fffffff;
        };
        int16_t blk[64];
        auto one_block = [&](JpegComp &c, int bx, int by) -> bool {
            if (!progressive) {
                if (!hdc[c.hd].defined || !hac[c.ha].defined) return false;
                if (!block_sequential(br, c, blk)) return false;
                jpeg_idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, blk);
                return true;
            }
            int16_t *d = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (ss == 0) return hdc[c.hd].defined && block_prog_dc(br, c, d);
            return hac[c.ha].defined && block_prog_ac(br, c, d);
        };
        if (scan_n == 1) { /* one component: its blocks in raster order, as many as its real size needs */
            JpegComp &c = comp[order[0]];
            const int bw = (c.px + 7) >> 3, bh = (c.py + 7) >> 3;
            for (int by = 0; by < bh && !stop; by++)
                for (int bx = 0; bx < bw && !stop; bx++) {
                    if (!one_block(c, bx, by)) return false;
                    mcu_done();
                }
        } else {
            if (progressive && ss != 0) return false; /* AC scans are never interleaved */
            for (int my = 0; my < mcuy && !stop; my++)
                for (int mx = 0; mx < mcux && !stop; mx++) {
                    for (int k = 0; k < scan_n; k++) {
                        JpegComp &c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++)
                                if (!one_block(c, mx * c.h + x, my * c.v + y)) return false;
                    }
                    mcu_done();
                }
        }
        /* where the entropy-coded data ended: the marker the bit reader met, or the next one in the file (zero bytes between
         * the data and the marker are passed over) */
        at = (size_t)(br.p - f.data());
        if (br.marker >= 0) pending = br.marker;
        else {
            pending = -1;
            while (at < f.size()) {
                int x = f[at++];
                if (x == 0xff) { pending = at < f.size() ? f[at++] : -1; break; }
                if (x != 0) return false;
            }
        }
        return true;
    }
    int pending = -1; /* marker met at the end of the last scan */
    void finish_progressive() {
        for (int i = 0; i < ncomp; i++) {
            JpegComp &c = comp[i];
            const int bw = (c.px + 7) >> 3, bh = (c.py + 7) >> 3;
            const unsigned char *q = quant[c.tq];
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    int16_t *d = &c.coef[((size_t)by * c.bw + bx) * 64];
                    for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * q[k]);
                    jpeg_idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, d);
                }
        }
    }
    bool decode() {
        if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8) return false;
        at = 2;
        int m = next_marker();
        while (m != 0xc0 && m != 0xc1 && m != 0xc2) {
            if (m < 0 || !read_segment(m)) return false;
            m = next_marker();
            while (m < 0 && at < f.size()) { at++; m = next_marker(); } /* padding after a segment */
        }
        progressive = m == 0xc2;
        if (!read_frame()) return false;
        m = next_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                if (!read_scan_header() || !decode_scan()) return false;
                m = pending;
                if (m < 0) return false;
                continue;
            }
            if (m < 0 || !read_segment(m)) return false;
            m = next_marker();
        }
        if (progressive) finish_progressive();
        return true;
    }
};

/* one output row of a component at full width: chroma interpolated 3:1 between the nearer and the farther source row /
 * column for the factor-2 cases (centred, JFIF siting), samples repeated otherwise */
const unsigned char *jpeg_upsample_row(unsigned char *out, const unsigned char *near_row, const unsigned char *far_row, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near_row;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (unsigned char)((3 * near_row[i] + far_row[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        const unsigned char *in = near_row;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < w - 1; i++) {
            out[2 * i] = (unsigned char)((3 * in[i] + in[i - 1] + 2) >> 2);
            out[2 * i + 1] = (unsigned char)((3 * in[i] + in[i + 1] + 2) >> 2);
        }
        out[2 * (w - 1)] = (unsigned char)((in[w - 2] + 3 * in[w - 1] + 2) >> 2);
        out[2 * (w - 1) + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) { /* vertical 3:1 first (kept times four), then horizontal 3:1: sixteenths */
        int cur = 3 * near_row[0] + far_row[0];
        if (w == 1) { out[0] = out[1] = (unsigned char)((cur + 2) >> 2); return out; }
        out[0] = (unsigned char)((cur + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int prev = cur;
            cur = 3 * near_row[i] + far_row[i];
            out[2 * i - 1] = (unsigned char)((3 * prev + cur + 8) >> 4);
            out[2 * i] = (unsigned char)((3 * cur + prev + 8) >> 4);
        }
        out[2 * w - 1] = (unsigned char)((cur + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)
        for (int j = 0; j < hs; j++) out[i * hs + j] = near_row[i];
    return out;
}

/* JPEG -> rows top-down, 1 (greyscale file) or 3 interleaved channels: stbi_load(..., 0) */
bool decode_jpeg(const std::vector<unsigned char> &f, std::vector<unsigned char> &px, int &w, int &h, int &nn) {
    if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8) return false;
    JpegDecoder z(f);
    if (!z.decode()) return false;
    w = z.width; h = z.height; nn = z.ncomp;
    px.assign((size_t)w * h * nn, 0);
    struct Up { int hs, vs, w_lores, ystep, ypos; const unsigned char *line0, *line1; std::vector<unsigned char> buf; } up[3];
    for (int k = 0; k < nn; k++) {
        Up &r = up[k];
        r.hs = z.hmax / z.comp[k].h;
        r.vs = z.vmax / z.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (w + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = z.comp[k].plane.data();
        r.buf.assign((size_t)w + 8, 0);
    }
    const int CR_R = 5743 << 8, CR_G = -(2925 << 8), CB_G = -(1410 << 8), CB_B = 7258 << 8; /* round(x * 4096) << 8 of 1.402, 0.71414, 0.34414, 1.772 */
    for (int j = 0; j < h; j++) {
        const unsigned char *row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < nn; k++) {
            Up &r = up[k];
            const bool lower = r.ystep >= (r.vs >> 1); /* the output row lies in the lower half of its source row: the next one is the nearer */
            row[k] = jpeg_upsample_row(r.buf.data(), lower ? r.line1 : r.line0, lower ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < z.comp[k].py) r.line1 += z.comp[k].w2;
            }
        }
        unsigned char *out = &px[(size_t)j * w * nn];
        if (nn == 1) { memcpy(out, row[0], (size_t)w); continue; }
        for (int i = 0; i < w; i++) {
            const int yf = (row[0][i] << 20) + (1 << 19), cb = row[1][i] - 128, cr = row[2][i] - 128;
            int r = yf + cr * CR_R;
            int g = yf + cr * CR_G + (int)((unsigned)(cb * CB_G) & 0xffff0000u);
            int b = yf + cb * CB_B;
            out[3 * i] = jpeg_clamp(r >> 20);
            out[3 * i + 1] = jpeg_clamp(g >> 20);
            out[3 * i + 2] = jpeg_clamp(b >> 20);
        }
    }
    return true;
}
} // namespace

/* ---- OpenEXR, the subset environment maps come in ---------------------------------------------------------
 * Single-part scan-line files; compression NONE, RLE, ZIPS, ZIP, PIZ; HALF, FLOAT or UINT channels; channels R, G,
 * B (or one luminance channel, replicated) anywhere in the channel list.  Replaces tinyexr's LoadEXR as the
 * reference uses it (image_util.cpp:21-45): rows top-down in data-window order, Spectrum(R, G, B) per pixel
 * (the reference's Flip(Spectrum(out[it++], out[it++], out[it++])) is R, G, B under g++'s right-to-left
 * evaluation), alpha dropped.  PXR24 / B44 / DWA, tiles, deep data and multi-part files are reported. */
namespace {
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal half: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
struct ExrChannel { std::string name; int type; int xs, ys; };
int32_t rd_i32(const unsigned char *p) { int32_t v; memcpy(&v, p, 4); return v; } /* files are little-endian, so is the host */
uint64_t rd_u64(const unsigned char *p) { uint64_t v; memcpy(&v, p, 8); return v; }

bool exr_unrle(const unsigned char *in, size_t n_in, std::vector<unsigned char> &out, size_t want) {
    out.clear();
    size_t i = 0;
    while (i < n_in) {
        int c = (signed char)in[i++];
        if (c < 0) {
            size_t n = (size_t)(-c);
            if (i + n > n_in) return false;
            out.insert(out.end(), in + i, in + i + n);
            i += n;
        } else {
            if (i >= n_in) return false;
            out.insert(out.end(), (size_t)c + 1, in[i++]);
        }
        if (out.size() > want) return false;
    }
    return out.size() == want;
}
/* the byte predictor and the even / odd split that ZIP and RLE blocks are stored with */
void exr_unfilter(std::vector<unsigned char> &t, std::vector<unsigned char> &out) {
    for (size_t i = 1; i < t.size(); i++) t[i] = (unsigned char)(t[i - 1] + t[i] - 128);
    out.resize(t.size());
    const size_t half = (t.size() + 1) / 2;
    for (size_t i = 0, a = 0, b = half; i < t.size(); ) {
        out[i++] = t[a++];
        if (i < t.size()) out[i++] = t[b++];
    }
}
/* ---- PIZ (OpenEXR ImfPizCompressor / ImfHuf / ImfWav): a block of up to 32 lines, channel-planar 16-bit words ->
 * value range compacted through a bitmap -> 2-D Haar-like wavelet per channel (per 16-bit half of a 32-bit channel) ->
 * canonical Huffman code with a run-length symbol.  Decoding runs the three steps backwards. */
struct BitReader {
    const unsigned char *p, *end;
    uint64_t c = 0;
    int lc = 0;
    bool ok = true;
    BitReader(const unsigned char *b, const unsigned char *e) : p(b), end(e) {}
    unsigned get(int n) { /* most significant bit first */
        while (lc < n) {
            if (p >= end) { ok = false; return 0; }
            c = (c << 8) | *p++;
            lc += 8;
        }
        lc -= n;
        return (unsigned)((c >> lc) & ((1ull << n) - 1));
    }
};
bool piz_huf_uncompress(const unsigned char *in, size_t n_in, std::vector<uint16_t> &out, size_t n_out) {
    out.assign(n_out, 0);
    if (n_out == 0) return n_in == 0 || n_in >= 20;
    if (n_in < 20) return false;
    const uint32_t im = (uint32_t)rd_i32(in), iM = (uint32_t)rd_i32(in + 4), n_bits = (uint32_t)rd_i32(in + 12);
    const uint32_t ENC = (1u << 16) + 1;
    if (im >= ENC || iM >= ENC || im > iM) return false;
    /* the code lengths im .. iM, 6 bits each; 59..62 = a run of 2..5 zero lengths, 63 + 8 bits = a run of 6..261 */
    std::vector<uint64_t> code(ENC, 0);
    BitReader tb(in + 20, in + n_in);
    for (uint32_t i = im; i <= iM; i++) {
        const unsigned l = tb.get(6);
        if (!tb.ok) return false;
        if (l == 63 || l >= 59) {
            unsigned zerun = l == 63 ? tb.get(8) + 6 : l - 59 + 2;
            if (!tb.ok || i + zerun > iM + 1) return false;
            i += zerun - 1; /* the lengths stay 0 */
        } else code[i] = l;
    }
    const unsigned char *data = tb.p; /* the table ends on a byte boundary of what was consumed */
    if ((size_t)(in + n_in - data) * 8 < n_bits) return false;
    /* canonical codes: within a length, in symbol order; the longest codes start at 0 */
    uint64_t n[59] = {};
    for (uint32_t i = 0; i < ENC; i++) n[code[i]]++;
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) {
        const uint64_t nc = (c + n[i]) >> 1;
        n[i] = c;
        c = nc;
    }
    uint64_t first[59], count[59] = {};
    std::vector<uint32_t> sym_at[59];
    for (int l = 1; l <= 58; l++) first[l] = n[l];
    for (uint32_t i = 0; i < ENC; i++) {
        const unsigned l = (unsigned)code[i];
        if (l > 0) { sym_at[l].push_back(i); count[l]++; }
    }
    BitReader br(data, in + n_in);
    uint64_t left = n_bits;
    size_t o = 0;
    const uint32_t rlc = iM;
    while (left > 0) {
        uint64_t acc = 0;
        int l = 0;
        uint32_t sym = 0;
        bool found = false;
        while (l < 58 && left > 0) {
            acc = (acc << 1) | br.get(1);
            left--;
            l++;
            if (!br.ok) return false;
            if (count[l] && acc >= first[l] && acc - first[l] < count[l]) { sym = sym_at[l][(size_t)(acc - first[l])]; found = true; break; }
        }
        if (!found) { /* trailing pad bits of the last byte */
            break;
        }
        if (sym == rlc) {
            if (left < 8) return false;
            const unsigned run = br.get(8);
            left -= 8;
            if (!br.ok || o == 0 || o + run > n_out) return false;
            for (unsigned k = 0; k < run; k++) { out[o] = out[o - 1]; o++; }
        } else {
            if (o >= n_out) return false;
            out[o++] = (uint16_t)sym;
        }
    }
    return o == n_out;
}
/* Inverse of PIZ's two-dimensional integer wavelet over one 16-bit plane (element (x, y) at v[x * sx + y * sy]).
 * Written from the format's definition, level by level, as two separable sweeps:
 *   a level of step p joins samples p apart into (average, difference) pairs; going back, the pair (l, h) gives
 *   14-bit data:  a = l + (h & 1) + (h >> 1),  b = a - h            (signed 16-bit arithmetic)
 *   16-bit data:  b = l - (h >> 1),  a = h + b - 0x8000              (modulo 2^16)
 * Along an axis of n samples, level p pairs (k 2p, k 2p + p) for k < n / 2p; a sample at (n / 2p) 2p is "alone" when bit p of n
 * is set: it takes part in the other axis' sweep only.  Rows are undone before columns wherever both apply, and the two sweeps
 * touch disjoint samples where only one applies, so each sweep can run over the whole plane before the next. */
struct PizUnlift {
    bool narrow; /* every value of the block fits 14 bits */
    void operator()(uint16_t &lo, uint16_t &hi) const {
        if (narrow) {
            const int d = (int16_t)hi, s = (int16_t)lo + (d & 1) + (d >> 1);
            lo = (uint16_t)(int16_t)s;
            hi = (uint16_t)(int16_t)(s - d);
        } else {
            const unsigned d = hi, second = (lo - (d >> 1)) & 0xffffu, first = (d + second - 0x8000u) & 0xffffu;
            lo = (uint16_t)first;
            hi = (uint16_t)second;
        }
    }
};
/* the positions along an axis of n samples that the OTHER axis' sweep of level p visits: both members of every pair, then the lone one */
static void piz_level_lines(int n, int p, std::vector<int> &lines) {
    lines.clear();
    const int pairs = n / (2 * p);
    for (int k = 0; k < pairs; k++) { lines.push_back(2 * p * k); lines.push_back(2 * p * k + p); }
    if (n & p) lines.push_back(2 * p * pairs);
}
void piz_wavelet_inverse(uint16_t *v, int nx, ptrdiff_t sx, int ny, ptrdiff_t sy, uint16_t max_value) {
    const PizUnlift unlift = {max_value < (1 << 14)};
    int top = 1;
    while (2 * top <= (nx < ny ? nx : ny)) top *= 2; /* largest power of two <= the shorter side */
    std::vector<int> cols, rows;
    for (int p = top / 2; p >= 1; p /= 2) {
        piz_level_lines(nx, p, cols);
        piz_level_lines(ny, p, rows);
        const int ypairs = ny / (2 * p), xpairs = nx / (2 * p);
        for (int x : cols) /* sweep along y */
            for (int k = 0; k < ypairs; k++) {
                uint16_t *q = v + x * sx + (ptrdiff_t)(2 * p * k) * sy;
                unlift(q[0], q[p * sy]);
            }
        for (int y : rows) /* sweep along x */
            for (int k = 0; k < xpairs; k++) {
                uint16_t *q = v + y * sy + (ptrdiff_t)(2 * p * k) * sx;
                unlift(q[0], q[p * sx]);
            }
    }
}
/* one PIZ block -> the raw scan-line layout the other compressions give (line by line, channel by channel) */
bool exr_unpiz(const unsigned char *src, size_t packed, const std::vector<ExrChannel> &chans, int w, int lines, std::vector<unsigned char> &raw) {
    if (packed < 4) return false;
    uint16_t min_nz, max_nz;
    memcpy(&min_nz, src, 2);
    memcpy(&max_nz, src + 2, 2);
    std::vector<unsigned char> bitmap(8192, 0);
    size_t at = 4;
    if (max_nz >= 8192) return false;
    if (min_nz <= max_nz) {
        const size_t nb = (size_t)max_nz - min_nz + 1;
        if (at + nb > packed) return false;
        memcpy(&bitmap[min_nz], src + at, nb);
        at += nb;
    }
    std::vector<uint16_t> lut(65536, 0);
    unsigned k = 0;
    for (unsigned i = 0; i < 65536; i++)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (at + 4 > packed) return false;
    const int32_t len = rd_i32(src + at);
    at += 4;
    if (len < 0 || at + (size_t)len > packed) return false;
    size_t total = 0;
    std::vector<size_t> start(chans.size());
    std::vector<int> size(chans.size());
    for (size_t c = 0; c < chans.size(); c++) {
        size[c] = chans[c].type == 1 ? 1 : 2;
        start[c] = total;
        total += (size_t)w * lines * size[c];
    }
    std::vector<uint16_t> buf;
    if (!piz_huf_uncompress(src + at, (size_t)len, buf, total)) return false;
    for (size_t c = 0; c < chans.size(); c++)
        for (int j = 0; j < size[c]; j++) piz_wavelet_inverse(&buf[start[c] + j], w, size[c], lines, (ptrdiff_t)w * size[c], max_value);
    for (size_t i = 0; i < total; i++) buf[i] = lut[buf[i]];
    raw.resize(total * 2);
    size_t o = 0;
    std::vector<size_t> pos(start);
    for (int y = 0; y < lines; y++)
        for (size_t c = 0; c < chans.size(); c++) {
            const size_t n = (size_t)w * size[c];
            memcpy(&raw[o], &buf[pos[c]], n * 2);
            o += n * 2;
            pos[c] += n;
        }
    return true;
}
bool exr_fail(const char *path, const char *why) {
    printf("Fail read %s: %s\n", path, why);
    return false;
}
bool exr_decode(const char *path, const std::vector<unsigned char> &f, std::vector<float> &rgb, int &w, int &h) {
    if (f.size() < 16 || f[0] != 0x76 || f[1] != 0x2f || f[2] != 0x31 || f[3] != 0x01) return exr_fail(path, "not an OpenEXR file");
    const uint32_t version = (uint32_t)rd_i32(&f[4]);
    if ((version & 0xff) != 2) return exr_fail(path, "unknown OpenEXR version");
    if (version & 0x1a00) return exr_fail(path, "tiled, deep or multi-part files are not supported");
    size_t at = 8;
    std::vector<ExrChannel> chans;
    int compression = -1, line_order = 0, box[4] = {0, 0, -1, -1};
    auto cstr = [&](std::string *out) {
        size_t e = at;
        while (e < f.size() && f[e]) e++;
        if (e >= f.size()) return false;
        out->assign((const char *)&f[at], e - at);
        at = e + 1;
        return true;
    };
    while (true) { /* attributes: name\0 type\0 size data, ended by an empty name */
        std::string name, type;
        if (!cstr(&name)) return exr_fail(path, "truncated header");
        if (name.empty()) break;
        if (!cstr(&type) || at + 4 > f.size()) return exr_fail(path, "truncated header");
        const int32_t size = rd_i32(&f[at]);
        at += 4;
        if (size < 0 || at + (size_t)size > f.size()) return exr_fail(path, "truncated header");
        const unsigned char *d = &f[at];
        if (name == "channels") {
            size_t c = 0;
            while (c < (size_t)size && d[c]) {
                ExrChannel ch;
                size_t e = c;
                while (e < (size_t)size && d[e]) e++;
                if (e + 17 > (size_t)size) return exr_fail(path, "bad channel list");
                ch.name.assign((const char *)d + c, e - c);
                ch.type = rd_i32(d + e + 1);
                ch.xs = rd_i32(d + e + 9);
                ch.ys = rd_i32(d + e + 13);
                chans.push_back(ch);
                c = e + 17;
            }
        } else if (name == "compression" && size >= 1) compression = d[0];
        else if (name == "dataWindow" && size >= 16) { for (int k = 0; k < 4; k++) box[k] = rd_i32(d + 4 * k); }
        else if (name == "lineOrder" && size >= 1) line_order = d[0];
        at += (size_t)size;
    }
    (void)line_order; /* every block carries its y, so the order blocks come in does not matter */
    const int64_t w64 = (int64_t)box[2] - box[0] + 1, h64 = (int64_t)box[3] - box[1] + 1; /* the corners are untrusted ints */
    if (w64 <= 0 || h64 <= 0 || w64 > (1 << 28) || h64 > (1 << 28) || w64 * h64 > (1 << 28)) return exr_fail(path, "bad data window");
    w = (int)w64;
    h = (int)h64;
    int lines_per_block;
    switch (compression) {
    case 0: case 1: case 2: lines_per_block = 1; break;
    case 3: lines_per_block = 16; break;
    case 4: lines_per_block = 32; break;
    default: return exr_fail(path, "compression not supported (NONE, RLE, ZIPS, ZIP and PIZ are)");
    }
    int ci[3] = {-1, -1, -1}, n_colour = 0;
    size_t line_bytes = 0;
    std::vector<size_t> ch_off(chans.size());
    for (size_t c = 0; c < chans.size(); c++) {
        if (chans[c].xs != 1 || chans[c].ys != 1) return exr_fail(path, "subsampled channels are not supported");
        if (chans[c].type < 0 || chans[c].type > 2) return exr_fail(path, "unknown pixel type");
        ch_off[c] = line_bytes;
        line_bytes += (size_t)w * (chans[c].type == 1 ? 2 : 4);
        if (chans[c].name == "R") ci[0] = (int)c;
        else if (chans[c].name == "G") ci[1] = (int)c;
        else if (chans[c].name == "B") ci[2] = (int)c;
    }
    for (int k = 0; k < 3; k++) n_colour += ci[k] >= 0;
    if (n_colour == 0) { /* a luminance image: its first non-alpha channel feeds all three */
        for (size_t c = 0; c < chans.size() && ci[0] < 0; c++)
            if (chans[c].name != "A") ci[0] = ci[1] = ci[2] = (int)c;
        if (ci[0] < 0) return exr_fail(path, "no colour channel");
    }
    const int n_blocks = (h + lines_per_block - 1) / lines_per_block;
    if (at + 8 * (size_t)n_blocks > f.size()) return exr_fail(path, "truncated offset table");
    rgb.assign((size_t)w * h * 3, 0.f);
    std::vector<unsigned char> raw, tmp;
    for (int b = 0; b < n_blocks; b++) {
        const uint64_t off = rd_u64(&f[at + 8 * (size_t)b]);
        if (off > f.size() || f.size() - off < 8) return exr_fail(path, "block offset outside the file"); /* (off + 8 could wrap) */
        const int64_t y64 = (int64_t)rd_i32(&f[off]) - box[1];
        const int32_t packed = rd_i32(&f[off + 4]);
        const int y0 = (y64 < 0 || y64 >= h) ? -1 : (int)y64;
        if (y0 < 0 || y0 >= h || packed < 0 || (uint64_t)packed > f.size() - off - 8) return exr_fail(path, "bad block");
        const int lines = y0 + lines_per_block <= h ? lines_per_block : h - y0;
        const size_t want = line_bytes * lines;
        const unsigned char *src = &f[off + 8];
        if ((size_t)packed == want || compression == 0) { /* stored as is when packing did not help */
            if ((size_t)packed != want) return exr_fail(path, "bad block size");
            raw.assign(src, src + want);
        } else if (compression == 4) {
            if (!exr_unpiz(src, (size_t)packed, chans, w, lines, raw) || raw.size() != want) return exr_fail(path, "bad PIZ block");
        } else if (compression == 1) {
            if (!exr_unrle(src, (size_t)packed, tmp, want)) return exr_fail(path, "bad RLE block");
            exr_unfilter(tmp, raw);
        } else {
            tmp.resize(want);
            uLongf got = (uLongf)want;
            if (uncompress(tmp.data(), &got, src, (uLong)packed) != Z_OK || got != want) return exr_fail(path, "bad ZIP block");
            exr_unfilter(tmp, raw);
        }
        for (int l = 0; l < lines; l++)
            for (int k = 0; k < 3; k++) {
                if (ci[k] < 0) continue; /* missing colour channel stays 0 */
                const ExrChannel &ch = chans[ci[k]];
                const unsigned char *p = &raw[line_bytes * l + ch_off[ci[k]]];
                float *out = &rgb[((size_t)(y0 + l) * w) * 3 + k];
                for (int x = 0; x < w; x++) {
                    float v;
                    if (ch.type == 1) { uint16_t hv; memcpy(&hv, p + 2 * x, 2); v = half_to_float(hv); }
                    else if (ch.type == 2) memcpy(&v, p + 4 * x, 4);
                    else { uint32_t u; memcpy(&u, p + 4 * x, 4); v = (float)u; }
                    out[3 * (size_t)x] = v;
                }
            }
    }
    return true;
}
} // namespace

Spectrum *ReadImageEXR(const char *path, int &width, int &height) {
    std::vector<unsigned char> file;
    if (!read_file(path, file)) {
        printf("Fail read %s: cannot open\n", path);
        return nullptr;
    }
    std::vector<float> rgb;
    if (!exr_decode(path, file, rgb, width, height)) return nullptr;
    Spectrum *out = new Spectrum[(size_t)width * height];
    for (size_t i = 0; i < (size_t)width * height; i++) out[i] = Spectrum(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    return out;
}

/* texture.cpp:29-64: bytes / 255.f into Spectrum, missing channels stay 0, alpha dropped */
ImageData *LoadTextureImageData(const char *path) {
    std::vector<unsigned char> file, px;
    int nx = 0, ny = 0, nn = 0;
    const size_t plen = strlen(path);
    if (plen >= 4 && strcmp(path + plen - 4, ".exr") == 0) { /* texture.cpp:32-34 */
        Spectrum *data = ReadImageEXR(path, nx, ny);
        if (!data) return nullptr;
        ImageData *img = new ImageData();
        img->data = data; img->width = nx; img->height = ny; img->is_valid = 1;
        return img;
    }
    if (!read_file(path, file)) {
        printf("Could not open image %s\n", path);
        return nullptr;
    }
    if (!decode_png(file, px, nx, ny, nn) && !decode_jpeg(file, px, nx, ny, nn) && !decode_pnm(file, px, nx, ny, nn)) {
        printf("No decoder for image %s (PNG 8-bit non-interlaced, JPEG baseline / progressive 8-bit and binary PGM/PPM are supported)\n", path);
        return nullptr;
    }
    ImageData *img = new ImageData();
    img->data = new Spectrum[(size_t)nx * ny];
    img->width = nx;
    img->height = ny;
    size_t at = 0;
    for (size_t i = 0; i < (size_t)nx * ny; i++) {
        Float r = 0, g = 0, b = 0;
        if (nn > 0) r = px[at++] / 255.f;
        if (nn > 1) g = px[at++] / 255.f;
        if (nn > 2) b = px[at++] / 255.f;
        if (nn == 4) at++;
        img->data[i] = Spectrum(r, g, b);
    }
    img->is_valid = 1;
    return img;
}

/* ---- particles (part_loader.cpp) ---------------------------------------------------------------------- */
enum { PP_INITIALIZED = 1, PP_ALLOCATED = 2 };
static inline bool fmt_vec(char c) { return c == 'v' || c == 'V'; }
static inline bool fmt_scalar(char c) { return c == 's' || c == 'S'; }

void PParserInitialize(PParser *parser, const char *format) {
    parser->format = std::string(format);
    parser->data = nullptr;
    parser->state = PP_INITIALIZED;
}

static void pp_allocate(PParser *parser, int count) {
    int nv = 0, ns = 0;
    for (char c : parser->format) {
        if (fmt_vec(c)) nv++;
        else if (fmt_scalar(c)) ns++;
        else printf("Skipping unknown format: %c\n", c);
    }
    PParsedData *d = new PParsedData();
    d->vector_data = nv ? new vec3f *[nv] : nullptr;
    d->scalar_data = ns ? new Float *[ns] : nullptr;
    for (int i = 0; i < nv; i++) d->vector_data[i] = new vec3f[count]();
    for (int i = 0; i < ns; i++) d->scalar_data[i] = new Float[count]();
    d->nvectors = nv; d->nscalars = ns; d->count = count;
    parser->data = d;
    parser->state = PP_ALLOCATED;
}

/* First payload line: particle count (read as a float, truncated).  Every further line: one value group
 * per format letter, 'v' = three floats, 's' = one.  True when the file held exactly `count` records
 * (part_loader.cpp:55-130).  Records beyond `count` are a malformed file: reported instead of written
 * past the arrays. */
bool PParserParse(PParser *parser, const char *filename) {
    if (!(parser->state == PP_INITIALIZED && parser->data == nullptr)) {
        printf("Invalid parser setup\n");
        return false;
    }
    std::ifstream ifs(filename);
    if (!ifs) {
        printf("Failed to open file: %s\n", filename);
        return false;
    }
    std::string line;
    const char *tok = nullptr;
    int count = 0, records = 0;
    bool first = true;
    clock_t start = clock();
    while (next_payload_line(ifs, line, &tok)) {
        if (first) {
            count = (int)ParseFloat(&tok);
            printf("Particle count %d\n", count);
            first = false;
            if (count < 1) return false;
            pp_allocate(parser, count);
            continue;
        }
        if (records >= count) {
            printf("Particle file holds more records than its header says (%d)\n", count);
            return false;
        }
        int vi = 0, si = 0;
        for (char c : parser->format) {
            if (fmt_vec(c)) {
                vec3f v;
                ParseV3(&v, &tok);
                parser->data->vector_data[vi++][records] = v;
            } else if (fmt_scalar(c)) {
                parser->data->scalar_data[si++][records] = ParseFloat(&tok);
            }
        }
        records++;
    }
    if (first) return false;
    printf("Took %g seconds, #p [%d] #v [%d] #s [%d].\n", (double)(clock() - start) / CLOCKS_PER_SEC, records, parser->data->nvectors,
           parser->data->nscalars);
    return records == count;
}

void PParserCleanup(PParser *parser) {
    if (parser->state != PP_ALLOCATED) return;
    PParsedData *d = parser->data;
    for (int i = 0; i < d->nvectors; i++) delete[] d->vector_data[i];
    for (int i = 0; i < d->nscalars; i++) delete[] d->scalar_data[i];
    delete[] d->vector_data;
    delete[] d->scalar_data;
    delete d;
    parser->data = nullptr;
    parser->state = PP_INITIALIZED;
}

vec3f *PParserGetVectorPtr(PParser *parser, int index, int *count) {
    *count = 0;
    if (parser->state != PP_ALLOCATED || index < 0 || parser->data->nvectors <= index) return nullptr;
    *count = parser->data->count;
    return parser->data->vector_data[index];
}
Float *PParserGetScalarPtr(PParser *parser, int index, int *count) {
    *count = 0;
    if (parser->state != PP_ALLOCATED || index < 0 || parser->data->nscalars <= index) return nullptr;
    *count = parser->data->count;
    return parser->data->scalar_data[index];
}

/* ---- plain-C views for the Python mirror (pathtracer_amd/assets.py) -------------------------------------- */
namespace {
struct ObjFile {
    std::vector<ParsedMesh *> *meshes;
    std::vector<MeshMtl> mtls;
};
struct MtlFile { std::vector<MTL *> mats; };
long copy_out(const std::string &s, char *buf, long cap) {
    if (buf && cap > 0) {
        long n = (long)s.size() < cap - 1 ? (long)s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (long)s.size();
}
} // namespace

extern "C" {
int pth_parse_double(const char *s, float *out) { return ParseDouble(s, s + strlen(s), out) ? 1 : 0; }

void *pth_obj_load(const char *path, int split) {
    ObjFile *f = new ObjFile();
    f->meshes = LoadObj(path, &f->mtls, split != 0);
    return f;
}
int pth_obj_count(void *h) { return (int)((ObjFile *)h)->meshes->size(); }
/* counts = {nTriangles, nVertices, nNormals, nUvs}; arrays stay owned by the handle */
void pth_obj_mesh(void *h, int i, int *counts, const float **p, const float **n, const float **uv, const int **indices) {
    ParsedMesh *m = ((ObjFile *)h)->meshes->at(i);
    counts[0] = m->nTriangles; counts[1] = m->nVertices; counts[2] = m->nNormals; counts[3] = m->nUvs;
    *p = (const float *)m->p; *n = (const float *)m->n; *uv = (const float *)m->uv; *indices = (const int *)m->indices;
}
long pth_obj_mtl(void *h, int i, int which, char *buf, long cap) {
    const MeshMtl &m = ((ObjFile *)h)->mtls.at(i);
    return copy_out(which ? m.name : m.file, buf, cap);
}
void pth_obj_free(void *h) {
    ObjFile *f = (ObjFile *)h;
    for (ParsedMesh *m : *f->meshes) FreeParsedMesh(m);
    delete f->meshes;
    delete f;
}

void *pth_mtl_load(const char *file, const char *basedir, int times) {
    MtlFile *f = new MtlFile();
    for (int t = 0; t < times; t++)
        if (!MTLParse(file, &f->mats, basedir)) { delete f; return nullptr; }
    return f;
}
int pth_mtl_count(void *h) { return (int)((MtlFile *)h)->mats.size(); }
long pth_mtl_name(void *h, int i, char *buf, long cap) { return copy_out(((MtlFile *)h)->mats.at(i)->mtlname, buf, cap); }
int pth_mtl_find(void *h, const char *name) {
    MtlFile *f = (MtlFile *)h;
    MTL *m = MTLFindMaterial(name, &f->mats);
    for (size_t i = 0; i < f->mats.size(); i++)
        if (f->mats[i] == m) return (int)i;
    return -1;
}
int pth_mtl_value(void *h, int i, const char *key, char *buf, long cap) {
    bool found = false;
    std::string v = MTLGetValue(key, ((MtlFile *)h)->mats.at(i), found);
    copy_out(v, buf, cap);
    return found ? 1 : 0;
}
int pth_mtl_spectrum(void *h, int i, const char *key, float *rgb) {
    bool found = false;
    Spectrum s = MTLGetSpectrum(key, ((MtlFile *)h)->mats.at(i), found);
    rgb[0] = s.x; rgb[1] = s.y; rgb[2] = s.z;
    return found ? 1 : 0;
}
/* MakeMTLMaterial as plain numbers: head = {type, is_emissive}; tex[16][6] = {kind, r, g, b, fval, image width} */
void pth_mtl_material(void *h, int i, int *head, float *tex) {
    MaterialDescriptor d = MakeMTLMaterial(((MtlFile *)h)->mats.at(i));
    head[0] = (int)d.type; head[1] = d.is_emissive;
    for (int t = 0; t < 16; t++) {
        const TextureDescriptor &x = d.textures[t];
        float *o = tex + 6 * t;
        o[0] = (float)x.type; o[1] = x.spec_val.x; o[2] = x.spec_val.y; o[3] = x.spec_val.z; o[4] = x.fval;
        o[5] = (x.type == 2 && x.image) ? (float)x.image->width : 0.f;
    }
}
void pth_mtl_free(void *h) {
    MtlFile *f = (MtlFile *)h;
    for (MTL *m : f->mats) delete m;
    delete f;
}

/* LoadTextureImageData: *rgb points at width*height*3 floats owned by the handle */
void *pth_image_load(const char *path, int *width, int *height, const float **rgb) {
    ImageData *img = LoadTextureImageData(path);
    if (!img) return nullptr;
    *width = img->width; *height = img->height; *rgb = (const float *)img->data;
    return img;
}
/* unit hook for tests: the JPEG decoder's inverse DCT on one block of dequantised coefficients (row-major) -> 64 samples */
void pth_jpeg_idct(const short *coef, unsigned char *out) { jpeg_idct_block(out, 8, (const int16_t *)coef); }
/* ReadImageEXR: width*height*3 floats the caller releases with pth_exr_free */
const float *pth_exr_read(const char *path, int *width, int *height) { return (const float *)ReadImageEXR(path, *width, *height); }
void pth_exr_free(const float *p) { delete[] (Spectrum *)p; }
void pth_image_free(void *h) {
    ImageData *img = (ImageData *)h;
    if (img) { delete[] img->data; delete img; }
}

void *pth_particles_load(const char *path, const char *format) {
    PParser *p = new PParser();
    PParserInitialize(p, format);
    if (!PParserParse(p, path)) { PParserCleanup(p); delete p; return nullptr; }
    return p;
}
const float *pth_particles_get(void *h, int scalar, int index, int *count) {
    return scalar ? (const float *)PParserGetScalarPtr((PParser *)h, index, count) : (const float *)PParserGetVectorPtr((PParser *)h, index, count);
}
void pth_particles_free(void *h) {
    PParser *p = (PParser *)h;
    PParserCleanup(p);
    delete p;
}
}
