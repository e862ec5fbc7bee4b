import numpy as np


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
