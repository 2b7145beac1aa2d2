"""surface-area estimate of the geometry tree's traversal cost, host SAH against device LBVH (needs a GPU for the latter)"""
import sys
import pathtracer_amd as pa

for fam, p0 in (("knot", 660), ("room", 224), ("fluid", 4000000), ("subsurface", 387)):
    sc = pa.BuiltScene(fam, 256, 144, p0, 0)
    for mode in ("host", "device"):
        pa.set_bvh_build_mode(mode)
        r = pa.Renderer(0)
        r.upload(sc)
        i = r.bvh_build_info()
        print("%-10s %-12s build %8.1f ms depth %2d  sah boxes %7.2f items %6.2f leaves %6.2f  nodes %d" % (
            fam, i["builder"], i["build_ms"], i["depth"], i["sah_boxes"], i["sah_items"], i["sah_leaves"], r.bvh_info()["nodes"]), flush=True)
        r.close()
    sc.close()
