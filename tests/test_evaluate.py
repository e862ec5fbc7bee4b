import os
import re

import numpy as np

from conftest import GOLDEN


def test_shape_metrics_definitions():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd.evaluate import shape_metrics
    gt = np.array([[0, 0, 1.0], [0, 0, 1.0], [1.0, 0, 0], [0, 1.0, 0]])
    pred = np.array([[0, 0, -2.0],                                        # flipped: 0 deg unoriented, 180 oriented
                     [0, np.sin(np.radians(8)), np.cos(np.radians(8))],   # 8 deg
                     [np.cos(np.radians(3)), np.sin(np.radians(3)), 0],   # 3 deg
                     [1.0, 0, 0]])                                        # 90 deg
    m = shape_metrics(pred, gt)
    ang = np.array([0, 8, 3, 90.0])
    assert abs(m["rms"] - np.sqrt(np.mean(ang ** 2))) < 1e-3
    assert m["pgp10"] == 0.75 and m["pgp5"] == 0.5
    assert abs(m["rms_o"] - np.sqrt(np.mean(np.array([180, 8, 3, 90.0]) ** 2))) < 1e-2


def test_evaluate_cli_writes_reference_summary(tmp_path):
    """utils/evaluate.py's file contract: <results>/summary/<dataset>_evaluation_results.txt with its seven log lines;
    .pidx selects the evaluated subset (both when the predictions are dense and when they are already sparse)."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd.evaluate import main
    rng = np.random.RandomState(0)
    data, res = tmp_path / "data", tmp_path / "res"
    data.mkdir(); res.mkdir()
    for name, sparse_pred in (("a", False), ("b", True)):
        gt = rng.normal(size=(50, 3)); gt /= np.linalg.norm(gt, axis=1, keepdims=True)
        idx = np.arange(0, 50, 5)
        pred = gt + 0.01 * rng.normal(size=gt.shape)
        np.savetxt(str(data / (name + ".normals")), gt)
        np.savetxt(str(data / (name + ".xyz")), gt)
        np.savetxt(str(data / (name + ".pidx")), idx, fmt="%d")
        np.savetxt(str(res / (name + ".normals")), pred[idx] if sparse_pred else pred)
    (data / "myset.txt").write_text("a\nb\n\n")
    out = main(["--normal_results_path", str(res) + "/", "--data_path", str(data) + "/", "--dataset_list", "myset"])
    assert 0 < out["myset"]["rms"] < 2 and out["myset"]["pgp5"] == 1.0
    lines = (res / "summary" / "myset_evaluation_results.txt").read_text().splitlines()
    assert len(lines) == 7 and lines[0].startswith("RMS per shape: [") and lines[1].startswith("RMS not oriented (shape average): ")
    assert lines[6].startswith("PGP5 average: ")


def test_evaluate_matches_the_reference_script_output(tmp_path):
    """tests/golden/eval_ref.npz holds a synthetic dataset and the summary files the reference's OWN utils/evaluate.py
    wrote for it (scripts/make_golden_eval.py, run in the build container): our command-line twin must write the same
    seven lines per dataset list, byte for byte once numpy 2's ``np.float32(..)`` scalar wrappers are removed from the
    reference's list lines (the reference's interpreter prints bare numbers there)."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd.evaluate import main
    g = np.load(os.path.join(GOLDEN, "eval_ref.npz"))
    data, res = str(tmp_path / "data") + "/", str(tmp_path / "res") + "/"
    os.makedirs(data)
    os.makedirs(res)
    for name in g["shape_names"]:
        np.savetxt(data + name + ".xyz", g[name + "_xyz"])
        np.savetxt(data + name + ".normals", g[name + "_gt"])
        np.savetxt(data + name + ".pidx", g[name + "_pidx"], fmt="%d")
        np.savetxt(res + name + ".normals", g[name + "_pred"])
    lists = [str(x) for x in g["list_names"]]
    for ln in lists:
        with open(data + ln + ".txt", "w") as f:
            f.write("\n".join(str(x) for x in g["list_" + ln]) + "\n\n")
    main(["--normal_results_path", res, "--data_path", data, "--dataset_list"] + lists)
    for ln in lists:
        want = re.sub(r"np\.float(?:32|64)\(([^)]*)\)", r"\1", str(g["summary_" + ln]))
        got = open(os.path.join(res, "summary", ln + "_evaluation_results.txt")).read()
        assert got == want, (got, want)
        assert len(want.splitlines()) == 7
