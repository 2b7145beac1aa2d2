/*
 * obj_scene.cpp — a scene function in the reference's own style (compare src/main.cpp:670-711,
 * 1092-1190) compiled against the host mirror: OBJ + MTL through LoadObj / MTLParseAll /
 * MakeMTLMaterial, the render loop, ImageWrite.  Nothing in it names HIP or the C-ABI.
 *
 *   obj_scene <mesh folder/> <file.obj> [--render out.png [spp]]
 *
 * Without --render it stops after flattening (runs on a machine without a GPU) and prints what
 * PrepareSceneForRendering would upload.
 */
#include "../pathtracer_amd/host/pt_loaders.h"
#include <cstdlib>
#include <iostream>

static void ObjScene(Camera *camera, Float aspect, const std::string &folder, const std::string &file) {
    camera->Config(Point3f(0.f, 1.8f, 4.5f), Point3f(0.f, 0.2f, 0.f), vec3f(0.f, 1.f, 0.f), 42.f, aspect);

    MaterialDescriptor white = MakeMatteMaterial(Spectrum(.6f));
    RectDescriptor floor = MakeRectangle(Translate(0, -0.8f, 0) * RotateX(-90), 9, 9);
    InsertPrimitive(floor, white);

    std::vector<MeshMtl> mtls;
    std::vector<MTL *> materials;
    std::vector<ParsedMesh *> *meshes = LoadObj((folder + file).c_str(), &mtls, true);
    bool rv = MTLParseAll(&materials, &mtls, folder.c_str());
    Transform place = Translate(0, 0.2f, 0) * RotateX(35);
    for (int i = 0; i < (int)meshes->size(); i++) {
        ParsedMesh *m = meshes->at(i);
        if (m->nTriangles > 0) {
            m->toWorld = place;
            MaterialDescriptor mat = (rv && !materials.empty()) ? MakeMTLMaterial(MTLFindMaterial(mtls[i].name.c_str(), &materials))
                                                               : MakePlasticMaterial(Spectrum(.7f, .2f, .1f), Spectrum(.4f), 0.03f);
            InsertPrimitive(MakeMesh(m), mat);
        }
    }
    MaterialDescriptor matEm = MakeEmissive(Spectrum(0.992, 0.964, 0.390) * 10);
    RectDescriptor light = MakeRectangle(Translate(0, 3.5f, 0) * RotateX(90), 2, 2);
    InsertPrimitive(light, matEm);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cout << "usage: obj_scene <folder/> <file.obj> [--render out.png [spp]]" << std::endl;
        return 2;
    }
    const bool render = argc >= 5 && std::string(argv[3]) == "--render";
    const int width = 160, height = 120, spp = argc >= 6 ? atoi(argv[5]) : 16;
    Aggregator scene;
    Camera camera;
    BeginScene(render ? &scene : nullptr);
    ObjScene(&camera, (Float)width / (Float)height, argv[1], argv[2]);
    if (!render) {
        FlatSceneHolder *h = FlattenScene();
        const PtrFlatScene *fs = FlatSceneGet(h);
        long tris = 0;
        for (int i = 0; i < fs->n_meshes; i++) tris += fs->meshes[i].n_triangles;
        printf("flat scene: %d primitives, %d meshes, %ld triangles\n", fs->n_primitives, fs->n_meshes, tris);
        FlatSceneFree(h);
        return 0;
    }
    PrepareSceneForRendering(&scene);
    Image *image = CreateImage(width, height);
    SetupPixels(image, 1234);
    for (int done = 0; done < spp; done += 4) Render(image, &scene, &camera, spp - done < 4 ? spp - done : 4);
    ImageWrite(image, argv[4], 2.f, ToneMapAlgorithm::Exponential);
    ImageStats(image);
    ImageFree(image);
    return 0;
}
