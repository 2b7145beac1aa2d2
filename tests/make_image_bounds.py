"""tests/image_bounds.json from the residuals test_image measured on the GPU:
    PT_RECORD_IMAGE_STATS=gpurun_out/image_stats.jsonl python -m pytest tests/test_gpu_parity.py -m gpu -k test_image   (on the MI355X)
    python tests/make_image_bounds.py gpurun_out/image_stats.jsonl                                                        (here)
bound = 10 x the largest residual seen for the case (several runs may be appended to the file)."""
import json
import os
import sys

worst = {}
for path in sys.argv[1:]:
    for line in open(path):
        d = json.loads(line)
        w = worst.setdefault(d["case"], [0.0, 0.0, 0.0])
        for i, k in enumerate(("mean_rel", "rmse_rel", "off")):
            w[i] = max(w[i], d[k])
out = {k: [10.0 * v for v in w] for k, w in sorted(worst.items())}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_bounds.json")
json.dump({"_measured_x10": True, **out}, open(dst, "w"), indent=0, sort_keys=True)
print("wrote", dst, len(out), "cases")
