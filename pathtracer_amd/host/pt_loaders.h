/*
 * pt_loaders.h — asset ingest of the host mirror: the reference's own OBJ
 * loader (src/third/obj_loader.h:8-42), MTL parser (src/third/mtl.h:7-60), MakeMTLMaterial
 * (src/core/scene.h:154), particle parser (src/third/part_loader.h:16-53) and texture image
 * loader (src/core/texture.h:92), same names and argument meaning.  Meshes come back as host
 * arrays (the reference hands out managed memory); MakeMesh / InsertPrimitive take them as is.
 */
#pragma once
#include "pt_scene.h"
#include <istream>
#include <map>
#include <string>
#include <vector>

struct MeshProperties { int flip_x; };
typedef struct { std::string file; std::string name; } MeshMtl;
typedef struct {
    std::string mtlname;
    std::string basedir;
    std::map<std::string, std::vector<std::string>> data;
} MTL;

/* obj_loader.h */
std::vector<ParsedMesh *> *LoadObj(const char *path, std::vector<MeshMtl> *mtls, bool split_mesh = true);
ParsedMesh *LoadObjOnly(const char *path);
ParsedMesh *DuplicateMesh(ParsedMesh *mesh, MeshProperties *props = nullptr);
void FreeParsedMesh(ParsedMesh *mesh); /* the reference never frees; long-lived hosts can */
std::istream &GetLine(std::istream &is, std::string &t);
Float ParseFloat(const char **token);
void ParseV3(vec3f *v, const char **token);
void ParseV2(vec2f *v, const char **token);

/* mtl.h */
bool ParseDouble(const char *s, const char *s_end, Float *result);
bool MTLParse(const char *path, std::vector<MTL *> *materials, const char *basedir);
Spectrum MTLGetSpectrum(const char *property, MTL *mtl, bool &found);
std::string MTLGetValue(const char *property, MTL *mtl, bool &found);
MTL *MTLFindMaterial(const char *name, std::vector<MTL *> *materials);
bool MTLParseAll(std::vector<MTL *> *materials, std::vector<MeshMtl> *desc, const char *basedir);
MaterialDescriptor MakeMTLMaterial(MTL *mtl);

/* texture.h:92.  Decoders present: PNG (8-bit, non-interlaced) and binary PPM/PGM; the reference
 * goes through stb_image, whose JPEG/BMP/TGA/... paths are not rebuilt; ".exr" goes to ReadImageEXR.  nullptr when the file cannot
 * be decoded (the reference asserts). */
ImageData *LoadTextureImageData(const char *path);
/* image_util.h: RGB of an OpenEXR file, rows top-down (scan-line files, NONE / RLE / ZIPS / ZIP); nullptr on failure */
Spectrum *ReadImageEXR(const char *path, int &width, int &height);

/* part_loader.h */
typedef struct {
    vec3f **vector_data;
    Float **scalar_data;
    int nvectors;
    int nscalars;
    int count;
} PParsedData;
typedef struct {
    std::string format;
    PParsedData *data;
    int state;
} PParser;
void PParserInitialize(PParser *parser, const char *format);
bool PParserParse(PParser *parser, const char *filename);
void PParserCleanup(PParser *parser);
vec3f *PParserGetVectorPtr(PParser *parser, int index, int *count);
Float *PParserGetScalarPtr(PParser *parser, int index, int *count);
