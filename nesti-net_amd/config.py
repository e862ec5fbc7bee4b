"""Hyper-parameters of the hot path (the reference's ``parameters.p`` / ``gmm.p``,
``test_n_est_w_experts.py:46-54,201``), as one dataclass."""
import ctypes
import json
from dataclasses import dataclass, field
from typing import Dict, List

MAX_SCALES = 4
MAX_EXPERTS = 8
DTYPES = {"f32": 0, "bf16": 1, "f16": 2, "bf16x3": 3, "f16x3": 4, "f16x3c": 5, "f16x8": 6, "f16x8c": 7}   # include/nesti_hip.h: NESTI_F32 ... NESTI_F16X8C
CASCADE_DTYPES = ("f16x3c", "f16x8c")     # the two-stage gate (gate margin, cascade statistics)
PAIR_DTYPES = ("f16x3", "bf16x3", "f16x3c", "f16x8", "f16x8c")   # activations as 16-bit (hi, lo) pairs
ARCH_EXPERTS, ARCH_SINGLE, ARCH_MULTI, ARCH_SWITCH = 0, 1, 2, 3
SWITCH_NOISE_THRESHOLD = 0.015   # models/ms_sw_n_est.py:80

# train_n_est_w_experts.py:62 (JSON-in-JSON there; plain dict here)
TRAINED_EXPERT_DICT = {0: [0], 1: [0], 2: [1], 3: [1], 4: [2], 5: [2], 6: [0, 1, 2]}


class CConfig(ctypes.Structure):
    """Mirror of ``nesti_config_t`` (include/nesti_hip.h)."""
    _fields_ = [("arch", ctypes.c_int),
                ("n_scales", ctypes.c_int),
                ("points_per_scale", ctypes.c_int),
                ("grid_n", ctypes.c_int),
                ("variance", ctypes.c_double),
                ("n_experts", ctypes.c_int),
                ("expert_scale_lo", ctypes.c_int * MAX_EXPERTS),
                ("expert_scale_cnt", ctypes.c_int * MAX_EXPERTS)]


@dataclass
class NestiConfig:
    """Defaults = the PUBLISHED Nesti-Net configuration BASELINE.json and SURVEY.md 8(d) name (radii 0.01/0.03/0.05,
    8^3 Gaussians, variance 0.0156 -- the '--num_gaussians 8' setting train_n_est_w_experts.py:55 recommends, with
    num_point 512, 7 experts and the expert_dict of :59-62).  They are NOT the argparse defaults of the training
    script (radii 0.005/0.01/0.03, num_gaussians 3, variance 0.111, train_n_est_w_experts.py:48,55-56); a trained
    model's values come from its parameters.p (tf_ckpt.load_reference_model)."""
    patch_radius: List[float] = field(default_factory=lambda: [0.01, 0.03, 0.05])
    num_point: int = 512
    n_gaussians: int = 8          # per axis
    gmm_variance: float = 0.0156
    n_experts: int = 7
    expert_dict: Dict[int, List[int]] = field(default_factory=lambda: dict(TRAINED_EXPERT_DICT))
    arch: int = ARCH_EXPERTS

    @property
    def n_scales(self):
        return len(self.patch_radius)

    @property
    def n_towers(self):
        """Normal-regression towers: E experts, 1 for the single-tower ablations, small+large for ms_sw_n_est."""
        return {ARCH_EXPERTS: self.n_experts, ARCH_SWITCH: 2}.get(self.arch, 1)

    @property
    def n_gate_out(self):
        """Columns of the gate output: E probabilities, the noise estimate for ms_sw_n_est, none otherwise."""
        return {ARCH_EXPERTS: self.n_experts, ARCH_SWITCH: 1}.get(self.arch, 0)

    @staticmethod
    def for_model(name):
        """The configuration this build uses for ``--model name`` when no parameters.p is given: the 8^3-grid setting
        above, with the scale count each model takes (ss_norm_est one radius, ms_sw_n_est exactly two,
        models/ms_sw_n_est.py:50).  These are not the argparse defaults of the reference's training scripts."""
        if name == "experts_n_est":
            return NestiConfig()
        if name == "ss_norm_est":
            return NestiConfig(patch_radius=[0.05], n_experts=1, expert_dict={0: [0]}, arch=ARCH_SINGLE)
        if name == "ms_norm_est":
            return NestiConfig(n_experts=1, expert_dict={0: [0, 1, 2]}, arch=ARCH_MULTI)
        if name == "ms_sw_n_est":
            return NestiConfig(patch_radius=[0.01, 0.05], n_experts=2, expert_dict={0: [0], 1: [1]}, arch=ARCH_SWITCH)
        raise ValueError("unknown model %r" % (name,))

    def default_expert_dict(self):
        """``models/experts_n_est.py:83-96`` when expert_dict is None."""
        n_rads, E = self.n_scales, self.n_experts
        out = []
        for i in range(n_rads):
            out += [[i]] * (E // n_rads)
        out += [list(range(n_rads))] * (E % n_rads)
        return {i: out[i] for i in range(E)}

    def to_c(self):
        c = CConfig()
        c.arch = self.arch
        c.n_scales = self.n_scales
        c.points_per_scale = self.num_point
        c.grid_n = self.n_gaussians
        c.variance = float(self.gmm_variance)
        c.n_experts = self.n_experts
        ed = self.expert_dict if self.expert_dict is not None else self.default_expert_dict()
        if len(ed) != self.n_experts:
            raise ValueError("Incompatible expert assignment values in variable expert_dict")
        for i in range(self.n_experts):
            scales = list(ed[i])
            lo = min(scales)                      # models/experts_n_est.py:100
            c.expert_scale_lo[i] = lo
            c.expert_scale_cnt[i] = len(scales)   # models/experts_n_est.py:101
        return c

    def to_json(self):
        d = dict(self.__dict__)
        d["expert_dict"] = None if self.expert_dict is None else {str(k): v for k, v in self.expert_dict.items()}
        return json.dumps(d)

    @staticmethod
    def from_json(s):
        d = json.loads(s)
        if d.get("expert_dict") is not None:
            d["expert_dict"] = {int(k): v for k, v in d["expert_dict"].items()}
        return NestiConfig(**d)
