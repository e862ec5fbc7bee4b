"""Golden vectors for nesti-net_amd/evaluate.py from the reference's own utils/evaluate.py.

Runs ONLY in the build container (needs /root/reference).  Builds a small synthetic dataset (three shapes; dense and
already-sparse predictions; a flipped, a noisy and a bad normal field), runs the reference script on it unmodified
(runpy, its own argparse flags) and stores the inputs plus the seven summary lines it writes per dataset list in
tests/golden/eval_ref.npz.

The script's two module-level imports that cannot load here -- `visualization` (imports tensorflow) and `utils` (imports
h5py) -- are only used under EXPORT = True (utils/evaluate.py:30, 78-99, 160-185); with EXPORT = False (the script's own
setting) none of their attributes is touched, so they are replaced by empty modules for this run.  The metric code
(utils/evaluate.py:106-198) runs as written.  Under Python 3 / numpy 2 the per-shape lists print as
[np.float64(..), ..]; the test compares the numbers, not the list formatting."""
import io
import os
import runpy
import sys
import tempfile
import types
from contextlib import redirect_stdout

import numpy as np

REF = "/root/reference/utils/evaluate.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eval_ref.npz")


def make_dataset(rng):
    shapes = {}
    for name, n, kind, sparse_pred in (("boxy", 300, "noisy", False), ("blob", 240, "flipped", True), ("disc", 180, "bad", False)):
        xyz = rng.normal(size=(n, 3)).astype(np.float32)
        gt = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
        pidx = np.sort(rng.choice(n, n // 4, replace=False))
        pred = gt + {"noisy": 0.05, "flipped": 0.15, "bad": 0.6}[kind] * rng.normal(size=gt.shape)
        if kind == "flipped":
            pred[::2] *= -1.7                      # unoriented metric must not care; un-normalised on purpose
        shapes[name] = dict(xyz=xyz.astype(np.float64), gt=gt.astype(np.float64), pidx=pidx,
                            pred=(pred[pidx] if sparse_pred else pred).astype(np.float64))
    lists = {"setA": ["boxy", "blob", "disc"], "setB": ["disc", "boxy"]}
    return shapes, lists


def write_dataset(root, shapes, lists):
    data, res = os.path.join(root, "data") + "/", os.path.join(root, "log", "results") + "/"
    os.makedirs(data)
    os.makedirs(res)
    for name, s in shapes.items():
        np.savetxt(data + name + ".xyz", s["xyz"])
        np.savetxt(data + name + ".normals", s["gt"])
        np.savetxt(data + name + ".pidx", s["pidx"], fmt="%d")
        np.savetxt(res + name + ".normals", s["pred"])
    for ln, names in lists.items():
        with open(data + ln + ".txt", "w") as f:
            f.write("\n".join(names) + "\n\n")
    return data, res


def main():
    shapes, lists = make_dataset(np.random.RandomState(20260210))
    with tempfile.TemporaryDirectory() as root:
        data, res = write_dataset(root, shapes, lists)
        for stub in ("visualization", "utils"):
            sys.modules[stub] = types.ModuleType(stub)
        argv = sys.argv
        sys.argv = [REF, "--normal_results_path", res, "--data_path", data, "--dataset_list"] + list(lists)
        try:
            with redirect_stdout(io.StringIO()):
                runpy.run_path(REF, run_name="__main__")
        finally:
            sys.argv = argv
        out = {}
        for ln in lists:
            out["summary_" + ln] = np.array(open(os.path.join(res, "summary", ln + "_evaluation_results.txt")).read())
    arrays = {}
    for name, s in shapes.items():
        for k, v in s.items():
            arrays["%s_%s" % (name, k)] = v
    np.savez_compressed(OUT, shape_names=np.array(list(shapes)), list_names=np.array(list(lists)),
                        **{"list_" + ln: np.array(v) for ln, v in lists.items()}, **arrays, **out)
    for k, v in out.items():
        print(k, "\n" + str(v))


if __name__ == "__main__":
    main()
