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
