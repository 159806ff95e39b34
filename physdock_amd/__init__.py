"""physdock_amd - MI355X-native sampler for PhysDock's redocking hot path.

    from physdock_amd import PhysDock, PhysDockConfig
    model = PhysDock(PhysDockConfig(model_name="medium")).to("cuda")
    model.load_state_dict(reference_state_dict)          # reference parameter names
    x = model.sample_diffusion(batch, num_sample=64, steps=40, karras_noise_schedule_power=1000)

Importing the package does not load the HIP library; the first kernel launch does, and
raises if `physdock_amd/libphysdock_hip.so` has not been built (python -m physdock_amd.build).
"""
from .configs import PhysDockConfig, small_config  # noqa: F401
from .import_weights import import_state_dict, import_unicore_ckpt  # noqa: F401
from .model import PhysDock, weighted_rigid_align  # noqa: F401
from .confidence import ConfidenceModule  # noqa: F401  (reference layers/confidence_module.py)
from .params import param_shapes, seeded_state_dict  # noqa: F401
from .driver import redock, redock_many  # noqa: F401  (multi-round caller of the sampler, reference redocking.py:156-342)
