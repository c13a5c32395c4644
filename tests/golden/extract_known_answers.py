"""Re-types the reference's own known-answer vectors as DATA (tests/golden/reference_known_answers.json).

Runs in the authoring container only.  The reference's unit tests tests/test_nms.py:11-217 (Caffe2
UtilsNMSTest.TestNMS / TestNMS1) and tests/test_box_coder.py:11-105 (UtilsBoxesTest.TestBboxTransformRandom) are
executed with the operator under test and numpy's assert helpers wrapped, so every (inputs, expected output)
pair they hold is captured as numbers.  No source text of the reference is stored."""
import importlib.util
import json
import os
import sys
import unittest

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()


def load_test_module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    out = {"nms": [], "box_decode": []}
    pending = {}

    # ---- NMS ------------------------------------------------------------------------------------
    tn = load_test_module("/root/reference/tests/test_nms.py", "ref_test_nms")
    real_nms = tn.box_nms

    def nms_spy(boxes, scores, thresh):
        pending.update(boxes=boxes.numpy().tolist(), scores=scores.numpy().tolist(), thresh=float(thresh))
        return real_nms(boxes, scores, thresh)

    def equal_spy(actual, expected, *a, **k):
        np.testing.assert_array_equal.__wrapped__(actual, expected)
        out["nms"].append(dict(pending, keep=np.asarray(expected).tolist()))

    wrapped_eq = np.testing.assert_array_equal
    equal_spy_fn = equal_spy
    np.testing.assert_array_equal = equal_spy_fn
    np.testing.assert_array_equal.__wrapped__ = wrapped_eq
    tn.box_nms = nms_spy
    suite = unittest.defaultTestLoader.loadTestsFromModule(tn)
    res = unittest.TextTestRunner(verbosity=0).run(suite)
    assert res.wasSuccessful()
    np.testing.assert_array_equal = wrapped_eq

    # ---- BoxCoder.decode ------------------------------------------------------------------------
    tb = load_test_module("/root/reference/tests/test_box_coder.py", "ref_test_box_coder")
    real_decode = tb.BoxCoder.decode

    def decode_spy(self, rel_codes, boxes):
        pending.clear()
        pending.update(deltas=rel_codes.numpy().tolist(), boxes=boxes.numpy().tolist(), weights=list(self.weights))
        return real_decode(self, rel_codes, boxes)

    wrapped_close = np.testing.assert_allclose

    def close_spy(actual, desired, *a, **k):
        wrapped_close(actual, desired, *a, **k)
        out["box_decode"].append(dict(pending, expected=np.asarray(desired).tolist(), atol=k.get("atol", 0.0)))

    tb.BoxCoder.decode = decode_spy
    np.testing.assert_allclose = close_spy
    res = unittest.TextTestRunner(verbosity=0).run(unittest.defaultTestLoader.loadTestsFromModule(tb))
    assert res.wasSuccessful()
    np.testing.assert_allclose = wrapped_close
    tb.BoxCoder.decode = real_decode

    path = os.path.join(HERE, "reference_known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
