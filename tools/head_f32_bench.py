"""cfg-1 head (image XProtoNet, fp32: 8 x 512 x 7 x 7 features, D = 512, P = 40) through pasn_xproto_head_fwd: us per call.
   python tools/head_f32_bench.py          (PASN_NO_PWTINY=1 for the previous routing)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoasnet_amd import model_builder, synth

DT = os.environ.get("DT", "f32")
cfg = dict(checkpoint_path="", name="XProtoNet", base_architecture="resnet18", pretrained=False, prototype_shape="(40, 512, 1, 1)", num_classes=4, img_size=224)
m = model_builder.build(cfg); synth.load_synth(m); m = m.to("cuda").eval()
if DT == "bf16": m.set_compute_dtype(torch.bfloat16)
x = synth.echo_clips((8, 3, 224, 224)).to("cuda")
if DT == "bf16": x = x.bfloat16()
with torch.no_grad():
    feat = m.cnn_backbone(x)
    orig = m.cnn_backbone
    class Fixed(torch.nn.Module):
        def forward(self, _x): return feat
    m.cnn_backbone = Fixed()
    for _ in range(10): out = m(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): out = m(x)
    b.record(); torch.cuda.synchronize()
    print(f"head only: {a.elapsed_time(b) / 200 * 1e3:.1f} us per call; logits {out[0][0].tolist()}")
    m.cnn_backbone = orig
    for _ in range(5): m(x)
    torch.cuda.synchronize(); a.record()
    for _ in range(50): m(x)
    b.record(); torch.cuda.synchronize()
    print(f"whole model fp32, 8 images: {a.elapsed_time(b) / 50:.3f} ms per call")
