import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch
cfg = PhysDockConfig(model_name="medium")
model = PhysDock(cfg); model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0)); model = model.cuda().eval()
dbatch = {k: v.cuda() for k, v in cfg1_batch(0).items()}
eng = model.engine(torch.device("cuda", 0))
pb = model._prepare_batch(dbatch)
eng.conditioning(pb); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    eng.conditioning(pb); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
