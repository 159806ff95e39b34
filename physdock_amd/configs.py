"""Model / data configuration for the PhysDock sampler hot path.

Mirrors the shape contract of the reference's ``PhysDockConfig`` factory
(reference: PhysDock/configs.py:4-195): same keyword names, same model sizes
("toy" .. "full"), same nested keys read by the model constructor
(reference: PhysDock/models/model.py:56-67).  The returned object supports both
attribute and item access like ``ml_collections.ConfigDict`` but has no
third-party dependency.
"""
from __future__ import annotations


class ConfigDict(dict):
    """dict with recursive attribute access (stand-in for ml_collections.ConfigDict)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            v = ConfigDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}


_DEPTHS = {
    # name: (atom, evoformer, pairformer, dit, heads)   reference configs.py:65-96
    "toy": (2, 2, 2, 2, 2),
    "tiny": (2, 2, 8, 4, 2),
    "small": (2, 3, 16, 8, 2),
    "medium": (3, 4, 24, 12, 3),
    "full": (3, 4, 48, 24, 4),
}


def PhysDockConfig(
        inference_mode=True,
        model_name="medium",
        num_augmentation_sample=48,
        crop_size=256,
        atom_crop_size=256 * 8,
        inf=1e9,
        eps=1e-8,
        max_msa_clusters=128,
        token_bond_threshold=2.4,
        sigma_data=16.,
        # extension (not in the reference): override channel widths / depths, used by
        # the small-dimension parity fixtures.  Keys: c_m c_s c_z c_a c_ap and
        # no_blocks_atom no_blocks_evoformer no_blocks_pairformer no_blocks_dit
        overrides=None,
        **unused_training_kwargs,
):
    if model_name not in _DEPTHS:
        raise ValueError("Unknown model name")
    nb_atom, nb_evo, nb_pair, nb_dit, nb_heads = _DEPTHS[model_name]
    dims = dict(c_m=256, c_s=512, c_z=128, c_a=128, c_ap=16,
                no_blocks_atom=nb_atom, no_blocks_evoformer=nb_evo,
                no_blocks_pairformer=nb_pair, no_blocks_dit=nb_dit)
    if overrides:
        dims.update(overrides)
    for k in ("c_m", "c_s", "c_z", "c_a"):
        assert dims[k] % 32 == 0, f"{k} must be a multiple of the head width 32"
    ref_dim, target_dim, msa_dim = 167, 65, 34
    cfg = {
        "inference_mode": inference_mode,
        "sigma_data": sigma_data,
        "data": {
            "crop_size": crop_size,
            "atom_crop_size": atom_crop_size,
            "max_msa_clusters": max_msa_clusters,
            "token_bond_threshold": token_bond_threshold,
        },
        "model": {
            "c_z": dims["c_z"],
            "num_augmentation_sample": num_augmentation_sample,
            "diffusion_conditioning": {
                "ref_dim": ref_dim, "target_dim": target_dim, "msa_dim": msa_dim,
                "c_a": dims["c_a"], "c_ap": dims["c_ap"], "c_s": dims["c_s"],
                "c_m": dims["c_m"], "c_z": dims["c_z"], "inf": inf, "eps": eps,
                "no_blocks_atom": dims["no_blocks_atom"],
                "no_blocks_evoformer": dims["no_blocks_evoformer"],
                "no_blocks_pairformer": dims["no_blocks_pairformer"],
            },
            "dit": {
                "c_a": dims["c_a"], "c_ap": dims["c_ap"], "c_s": dims["c_s"],
                "c_z": dims["c_z"], "inf": inf, "eps": eps,
                "no_blocks_atom": dims["no_blocks_atom"],
                "no_blocks_dit": dims["no_blocks_dit"],
                "sigma_data": sigma_data,
            },
            "confidence_module": {
                "c_a": dims["c_a"], "c_ap": dims["c_ap"], "c_s": dims["c_s"], "c_z": dims["c_z"],
                "inf": inf, "eps": eps,
                "no_blocks_heads": dims.get("no_blocks_heads", nb_heads),
                "no_blocks_atom": dims["no_blocks_atom"],
            },
        },
    }
    return ConfigDict(cfg)


#: the small configuration used for the committed golden fixtures
SMALL_OVERRIDES = dict(c_a=32, c_ap=8, c_s=64, c_z=32, c_m=32,
                       no_blocks_atom=2, no_blocks_evoformer=2,
                       no_blocks_pairformer=2, no_blocks_dit=2)


def small_config():
    return PhysDockConfig(model_name="toy", overrides=SMALL_OVERRIDES)


def ffn_hidden(dim: int) -> int:
    """SwiGLU hidden width (reference feed_forward.py:18-25): 128*ceil(floor(8*dim/3)/128)."""
    h = int(2 * (4 * dim) / 3)
    return 128 * ((h + 127) // 128)
