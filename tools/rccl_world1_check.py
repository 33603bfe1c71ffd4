"""RCCL smoke on a 1-GPU box: the collective code paths (FedAvg / FedBN all-reduce of the flat arena, sharded style
statistics, the secondary bench) with backend nccl and world_size 1.  python tools/rccl_world1_check.py"""
import os, sys, types, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench_resnet
from ccst_amd import fed, style
from ccst_amd.nets import models
args = types.SimpleNamespace(mode="fedavg", dg_method="no_DG")
m = models.get_network("resnet18")(args, pretrained=False, classes=7).to(dev)
before = fed.FlatParams.of(m).flat.clone()
fed.communication_distributed(args, m, 1.0)
torch.cuda.synchronize()
print("fedavg all-reduce (RCCL, world 1) max diff", float((fed.FlatParams.of(m).flat - before).abs().max()))
srv = models.get_network("resnet18")(args, pretrained=False, classes=7).to(dev)
fed.communication_distributed(types.SimpleNamespace(mode="fedbn"), m, 1.0, server_model=srv)
print("fedbn ok", float((fed.FlatParams.of(srv).flat - before).abs().max()))
acc = style.StyleStatAccumulator(); acc.update(torch.rand(2, 512, 8, 8, device=dev).contiguous(memory_format=torch.channels_last)); acc.all_reduce(); print("stats all-reduce ok", acc.count)
out = bench_resnet.run(dev, world=1, steps=3, warmup=1)
print(out["value"])
dist.barrier(); dist.destroy_process_group()
