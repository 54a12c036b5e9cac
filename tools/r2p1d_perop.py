import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import torch
from protoasnet_amd import model_builder, synth
cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="resnet2p1d_18", backbone_last_layer_num=-3, pretrained=False, prototype_shape="(40, 256, 1, 1, 1)", num_classes=4, img_size=112)
m = model_builder.build(cfg); synth.load_synth(m)
m = m.to('cuda').eval().set_compute_dtype(torch.bfloat16)
x = synth.echo_clips((8,3,32,112,112)).to('cuda').bfloat16()
with torch.no_grad():
    for _ in range(3): m(x)
trunk = m.cnn_backbone
plan = trunk.plan_for(x)
probe = {i: [] for i in range(len(plan.ops))}
trunk._timers = probe
with torch.no_grad():
    for _ in range(3): m(x)
torch.cuda.synchronize()
trunk._timers = None
tot=0
for i, evs in probe.items():
    ms = sum(a.elapsed_time(b) for a,b in evs)/len(evs); tot+=ms
    mt = plan.meta[i]
    print(f"{i:3d} {mt['kernel']:36s} {mt.get('shape',''):46s} {ms*1e3:8.1f} us {mt['flops']/ms/1e9:7.1f} TF/s {mt['bytes']/ms/1e6:7.1f} GB/s")
print('total ms', tot)
