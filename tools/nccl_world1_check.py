# GPU check: RCCL initialises and all-gathers at world size 1; the sharded driver round-trips on it
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
import numpy as np
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth, sharded
V, keys, Ch = synth.scene(200000, 10, 59, 7)
kd = torch.from_numpy(keys.view(np.int64)).cuda(); Cd = torch.from_numpy(Ch).cuda()
sh = sharded.ShardedRaht(kd, 30, prefix_bits=9)
x = torch.ones(4, 59, device=dev)
out = torch.empty(4, 59, device=dev)
dist.all_gather_into_tensor(out, x)
print("nccl world-1 all_gather ok", out.sum().item(), "roundtrip", sh.roundtrip_error(Cd))
dist.barrier()
dist.destroy_process_group()
