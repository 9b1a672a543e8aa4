"""keep_aspect / strict_nms / precision options of ObjectDetector.predict on a trained model (usage: weights.npz vocdir)."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import pytoolkit as tk  # noqa: E402

wpath, vocdir = sys.argv[1], sys.argv[2]
X, y = tk.data.voc.load_07_test(vocdir)
for kw in ({}, {"keep_aspect": True}, {"strict_nms": True}, {"precision": "mixed"}, {"keep_aspect": True, "strict_nms": True, "precision": "mixed"}):
    od = tk.dl.od.ObjectDetector.load_voc(16, (320, 320), weights=wpath, use_multi_gpu=False, **kw)
    s = tk.data.voc.evaluate(y, od.predict(list(X)))
    print(kw, f'mAP={s["mAP"] * 100:.1f} mAP(VOC2007)={s["mAP_VOC"] * 100:.1f}', flush=True)
    del od
