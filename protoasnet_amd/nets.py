"""The three prototype networks of the reference, on the MI355X HIP path.

Same constructor kwargs, attributes, method names, return tuples and ``state_dict`` keys as
``/root/reference/src/models/{ProtoPNet,XProtoNet,Video_XProtoNet}.py``, so the
reference's agents / push / explain code can sit on top.  What differs is underneath: the trunk runs as
a compiled list of fused HIP launches (``plan.py``), and everything after the trunk is one C-ABI call
(``pasn_l2_head_fwd`` / ``pasn_xproto_head_fwd``).  Eval mode serves inference, ``push_forward`` and
``compute_occurence_map`` under ``torch.no_grad()``; train mode (``model.train()``) of the head-B models on the X3D trunks runs
the compiled forward + backward launch lists of ``train.py`` under autograd.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import XProtoDesc
from .backbones import base_architecture_to_features
from .plan import Act, HipTrunk, PlanBuilder, channels_last_rows, logical_view, pack_conv_weight, round_up
from .receptive_field import compute_proto_layer_rf_info_v2


# --------------------------------------------------------------------------------------------------
class PointwiseChain(nn.Sequential):
    """``add_on_layers`` / ``occurrence_module``: a Sequential of 1x1(x1) convs and activations.

    Holds the parameters under the reference's keys (``add_on_layers.0.weight`` ...).  Called as a module it
    runs every conv on the MFMA conv kernel with the activation fused (PPNet path); ``packed()`` hands the
    packed weights to the fused head-B entry point (XProtoNet path).
    """

    def __init__(self, *mods):
        super().__init__(*mods)
        self._cache = {}

    def convs(self):
        return [m for m in self if isinstance(m, (nn.Conv2d, nn.Conv3d))]

    def _steps(self):
        """[(conv, activation-name)] -- the activation module that follows each conv, if any."""
        mods, out = list(self), []
        for i, m in enumerate(mods):
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                act = "relu" if isinstance(nxt, nn.ReLU) else "sigmoid" if isinstance(nxt, nn.Sigmoid) else "none"
                out.append((m, act))
        return out

    def _sig(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed(self, cin_p: int, dtype: torch.dtype, frag: bool = False):
        """[(weight [rows][1][kc] in dtype, bias fp32 [rows] or None)] per conv, cached until a parameter changes.
        ``frag``: weights fragment-major, [rows / 32][kc / 16][64 lanes][8] (bf16; what ``pasn_xproto_chain_fwd`` reads)."""
        key, sig = ("packed", cin_p, dtype, frag), self._sig()
        hit = self._cache.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        out, cp = [], cin_p
        for conv in self.convs():
            w, kc, rows = pack_conv_weight(conv.weight, cp, dtype)
            if frag:
                w = w.view(rows // 32, 32, kc // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
            b = None
            if conv.bias is not None:
                b = torch.zeros(rows, dtype=torch.float32, device=w.device)
                b[: conv.out_channels] = conv.bias.detach().float()
            out.append((w, b))
            cp = round_up(conv.out_channels, 8)
        self._cache[key] = (sig, out)
        return out

    def run_rows(self, rows: torch.Tensor, shape5, c: int, dtype: torch.dtype) -> torch.Tensor:
        """rows: channels-last [N][S][Cp] storage of a (N,c,T,H,W) map -> channels-last storage [N][T][H][W][Dp]."""
        n, t, h, w = shape5
        key, sig = ("plan", tuple(rows.shape), tuple(shape5), c, dtype, rows.device, _lib.tuning_epoch()), self._sig()
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            pb = PlanBuilder(rows.device, dtype, dtype)
            x_in = Act(n, t, h, w, c, rows.shape[2], pb._new_buf(rows.numel() * rows.element_size(), external=True))
            a = x_in
            for conv, act in self._steps():
                a = pb.conv(a, conv, None, act)
            hit = (sig, pb.finish(x_in, a))
            self._cache[key] = hit
        return hit[1].run(rows)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("PointwiseChain: the HIP forward has no autograd backward yet; use torch.no_grad()")
        dtype = next(self.parameters()).dtype
        rows, s, cp = channels_last_rows(x, dtype)
        video = x.dim() == 5
        t, h, w = (x.shape[2], x.shape[3], x.shape[4]) if video else (1, x.shape[2], x.shape[3])
        y = self.run_rows(rows, (x.shape[0], t, h, w), x.shape[1], dtype)
        return logical_view(y, self.convs()[-1].out_channels, video)


def _f32(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# --------------------------------------------------------------------------------------------------
class PPNet(nn.Module):
    """ProtoPNet baseline (reference src/models/ProtoPNet.py:57-340) -- head A."""

    def __init__(self, features, img_size, prototype_shape, proto_layer_rf_info, num_classes, init_weights=True,
                 prototype_activation_function="log", add_on_layers_type="bottleneck"):
        super().__init__()
        self.img_size = img_size
        self.prototype_shape = prototype_shape
        self.num_prototypes = prototype_shape[0]
        self.num_classes = num_classes
        self.epsilon = 1e-4  # ProtoPNet.py:74
        self.prototype_activation_function = prototype_activation_function
        self.prototype_class_identity = self.get_prototype_class_identity()
        self.proto_layer_rf_info = proto_layer_rf_info
        self.features = features  # key name kept for checkpoint loading (ProtoPNet.py:84-85)
        self.compute_dtype: Optional[torch.dtype] = None

        cin = self.get_cnn_backbone_out_channels(self.features)
        depth = self.prototype_shape[1]
        mods = []
        if add_on_layers_type == "bottleneck":  # ProtoPNet.py:91-115: halve until the prototype depth is reached
            cur = cin
            while cur > depth or not mods:
                nxt = max(depth, cur // 2)
                mods += [nn.Conv2d(cur, nxt, 1), nn.ReLU(), nn.Conv2d(nxt, nxt, 1)]
                if nxt > depth:
                    mods.append(nn.ReLU())
                else:
                    assert nxt == depth
                    mods.append(nn.Sigmoid())
                cur = cur // 2
        else:  # 'regular', ProtoPNet.py:117-130
            mods = [nn.Conv2d(cin, depth, 1), nn.ReLU(), nn.Conv2d(depth, depth, 1), nn.Sigmoid()]
        self.add_on_layers = PointwiseChain(*mods)

        self.prototype_vectors = nn.Parameter(torch.rand(self.prototype_shape), requires_grad=True)
        self.ones = nn.Parameter(torch.ones(self.prototype_shape), requires_grad=False)
        self.last_layer = nn.Linear(self.num_prototypes, self.num_classes, bias=False)
        if init_weights:
            self._initialize_weights(self.add_on_layers)
            self.set_last_layer_incorrect_connection(incorrect_strength=-0.5)

    # ---- helpers shared with the subclasses (reference keeps them on PPNet too) ----------------------
    @staticmethod
    def get_cnn_backbone_out_channels(features) -> int:
        """Reference sniffs ``str(module).upper()`` (ProtoPNet.py:152-162); X3D is this build's addition."""
        name = str(features).upper()
        if "RESNET2P1D" in name or name.startswith("X3D"):
            return [m for m in features.modules() if isinstance(m, nn.Conv3d)][-1].out_channels
        if name.startswith("VGG") or name.startswith("RES"):
            return [m for m in features.modules() if isinstance(m, nn.Conv2d)][-1].out_channels
        if name.startswith("DENSE"):
            return [m for m in features.modules() if isinstance(m, nn.BatchNorm2d)][-1].num_features
        raise Exception("other base base_architecture NOT implemented")

    def get_prototype_class_identity(self) -> torch.Tensor:
        assert self.num_prototypes % self.num_classes == 0
        per_class = self.num_prototypes // self.num_classes
        ident = torch.zeros(self.num_prototypes, self.num_classes)
        ident[torch.arange(self.num_prototypes), torch.arange(self.num_prototypes) // per_class] = 1
        return ident

    def set_last_layer_incorrect_connection(self, incorrect_strength) -> None:
        own = torch.t(self.prototype_class_identity)
        self.last_layer.weight.data.copy_(1 * own + incorrect_strength * (1 - own))

    def _initialize_weights(self, layer) -> None:
        for m in layer.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def set_compute_dtype(self, dtype: Optional[torch.dtype]) -> "PPNet":
        """Activations / packed weights in ``dtype`` (fp32 accumulate) while parameters stay as they are."""
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, HipTrunk):
                m.compute_dtype = dtype
        return self

    def _dtype(self) -> torch.dtype:
        return self.compute_dtype or self.prototype_vectors.dtype

    def _guard(self, x: torch.Tensor) -> None:
        if self.training:
            raise NotImplementedError("only forward() (and, for the XProtoNet models, compute_occurence_map()) run in train mode; "
                                      "call .eval() for push_forward / conv_features / prototype_distances")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("the eval-mode HIP forward records no autograd graph: run it under torch.no_grad(), "
                                      "or call .train() for the differentiable (batch-statistics) pass")
        if not x.is_cuda:
            raise RuntimeError("protoasnet_amd models run on the GPU only; there is no CPU fallback")

    # ---- reference surface --------------------------------------------------------------------------
    def _conv_rows(self, x: torch.Tensor):
        """trunk -> add-on (+Sigmoid): channels-last rows [N][S][Dp] plus the spatial shape."""
        self._guard(x)
        feat = self.features(x)
        dtype = self._dtype()
        rows, s, cp = channels_last_rows(feat, dtype)
        n, c, h, w = feat.shape
        z = self.add_on_layers.run_rows(rows, (n, 1, h, w), c, dtype)  # [N][1][H][W][Dp]
        return z, (n, h, w)

    def conv_features(self, x: torch.Tensor) -> torch.Tensor:
        z, _ = self._conv_rows(x)
        return logical_view(z, self.prototype_shape[1], video=False)

    def _head(self, z: torch.Tensor, n: int, s: int, want_dist: bool):
        P, D, K = self.num_prototypes, self.prototype_shape[1], self.num_classes
        assert self.prototype_shape[2] == 1 and self.prototype_shape[3] == 1, "prototypes are 1x1 patches"
        dev = z.device
        dist = torch.empty((n, P, s), dtype=torch.float32, device=dev) if want_dist else None
        min_d = torch.empty((n, P), dtype=torch.float32, device=dev)
        logits = torch.empty((n, K), dtype=torch.float32, device=dev)
        act = self.prototype_activation_function
        protos, fcw = _f32(self.prototype_vectors), _f32(self.last_layer.weight)
        _lib.check(
            _lib.lib().pasn_l2_head_fwd(
                z.data_ptr(), protos.data_ptr(), fcw.data_ptr(), _lib.ptr(dist), min_d.data_ptr(), 0, logits.data_ptr(),
                n, s, D, z.shape[-1], P, K, _lib.dtype_code(z.dtype), 1 if act == "linear" else 0, float(self.epsilon),
                _lib.current_stream(),
            )
        )
        if act not in ("log", "linear"):
            # a callable activation (ProtoPNet.py:222-223): the kernel supplies the distance map / global minima, the callable and
            # the (P x K) last layer run as plain torch ops on the (N, P) minima
            logits = torch.nn.functional.linear(act(min_d), fcw)
        return logits, min_d, dist

    @staticmethod
    def _weighted_l2_convolution(input, filter, weights):  # noqa: A002 -- reference argument names
        """Weighted squared distance of every 1x1 patch to every prototype (reference ProtoPNet.py:165-187; no caller anywhere in the
        reference, so it is not on the hot path and has no kernel: three dense contractions in torch on whatever device the operands
        are on).  input (N,c,h,w), filter and weights (P,c,1,1) -> (N,P,h,w), clamped at zero like the reference's ``F.relu``."""
        if filter.shape[2:] != (1, 1) or weights.shape != filter.shape:
            raise ValueError("prototypes and weights are (P, c, 1, 1) patches")
        wf = weights.flatten(1)  # (P, c)
        ff = filter.flatten(1)
        x2w = torch.einsum("nchw,pc->nphw", input * input, wf)
        xpw = torch.einsum("nchw,pc->nphw", input, ff * wf)
        p2w = (ff * ff * wf).sum(1).view(1, -1, 1, 1)
        return torch.relu(x2w - 2 * xpw + p2w)

    def _l2_convolution(self, x: torch.Tensor) -> torch.Tensor:
        """Distance map of a logical (N,D,H,W) feature tensor (reference ProtoPNet.py:189-207)."""
        rows, s, cp = channels_last_rows(x, self._dtype())
        _, _, dist = self._head(rows, x.shape[0], s, want_dist=True)
        return dist.view(x.shape[0], self.num_prototypes, x.shape[2], x.shape[3])

    def prototype_distances(self, x: torch.Tensor) -> torch.Tensor:
        z, (n, h, w) = self._conv_rows(x)
        _, _, dist = self._head(z, n, h * w, want_dist=True)
        return dist.view(n, self.num_prototypes, h, w)

    def distance_2_similarity(self, distances: torch.Tensor) -> torch.Tensor:
        if self.prototype_activation_function == "log":
            return torch.log((distances + 1) / (distances + self.epsilon))
        if self.prototype_activation_function == "linear":
            return -distances
        return self.prototype_activation_function(distances)

    def _train_pass_a(self, x: torch.Tensor):
        """Train-mode forward of the ProtoPNet model (differentiable; ``train.TrainRunner`` with head A)."""
        from .train import TrainRunner

        if not x.is_cuda:
            raise RuntimeError("protoasnet_amd models run on the GPU only; there is no CPU fallback")
        if x.shape[1] != 3:
            raise NotImplementedError("the training pass takes the reference's 3-channel clip; single-channel (grey) input is an "
                                      "eval-mode path (protoasnet_amd.data.DeviceClipPipeline)")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        runners = self.__dict__.setdefault("_train_runners", {})
        key = (tuple(x.shape), x.dtype, self._dtype(), "A", hash(tuple((p.data_ptr(), p.requires_grad) for p in self.parameters())),
               tuple(self.prototype_shape), _lib.tuning_epoch())
        runner = runners.get(key)
        if runner is None:
            for stale in [k for k in runners if k[:4] == key[:4]]:
                del runners[stale]
            runner = runners[key] = TrainRunner(self, x, 0, head="A")
        return runner(x)

    def forward(self, x: torch.Tensor):
        if self.training:
            return self._train_pass_a(x)
        z, (n, h, w) = self._conv_rows(x)
        logits, min_d, _ = self._head(z, n, h * w, want_dist=False)
        return logits, min_d

    def push_forward(self, x: torch.Tensor):
        z, (n, h, w) = self._conv_rows(x)
        _, _, dist = self._head(z, n, h * w, want_dist=True)
        conv_output = logical_view(z, self.prototype_shape[1], video=False)
        if conv_output.dtype != torch.float32:
            conv_output = conv_output.float()  # callers do .cpu().numpy() (push_ProtoPNet.py:179)
        return conv_output, dist.view(n, self.num_prototypes, h, w)

    def prune_prototypes(self, prototypes_to_prune) -> None:
        keep = sorted(set(range(self.num_prototypes)) - set(prototypes_to_prune))
        self.prototype_vectors = nn.Parameter(self.prototype_vectors.data[keep, ...], requires_grad=True)
        self.prototype_shape = list(self.prototype_vectors.size())
        self.num_prototypes = self.prototype_shape[0]
        self.last_layer.in_features = self.num_prototypes
        self.last_layer.out_features = self.num_classes
        self.last_layer.weight.data = self.last_layer.weight.data[:, keep]
        self.ones = nn.Parameter(self.ones.data[keep, ...], requires_grad=False)
        self.prototype_class_identity = self.prototype_class_identity[keep, :]

    def __repr__(self):
        return (
            "PPNet(\n\tfeatures: {},\n\timg_size: {},\n\tprototype_shape: {},\n\tproto_layer_rf_info: {},\n"
            "\tnum_classes: {},\n\tepsilon: {}\n)"
        ).format(self.features, self.img_size, self.prototype_shape, self.proto_layer_rf_info, self.num_classes, self.epsilon)


# --------------------------------------------------------------------------------------------------
def _occurrence_module(conv, cin: int, depth: int, num_prototypes: int) -> PointwiseChain:
    return PointwiseChain(
        conv(cin, depth, kernel_size=1), nn.ReLU(), conv(depth, depth // 2, kernel_size=1), nn.ReLU(),
        conv(depth // 2, num_prototypes, kernel_size=1, bias=False),
    )


class _XProtoHeadMixin:
    """Head B shared by the image and video models: one ``pasn_xproto_head_fwd`` call after the trunk."""

    def _train_pass(self, x: torch.Tensor, mode: int):
        """Train-mode pass (batch-statistics norm, differentiable): one compiled forward + backward launch list per input
        shape (``train.TrainRunner``), driven by autograd -- the path ``loss.backward()`` of the reference's agents takes
        (Video_XProtoNet_e2e.py:118-141; ``compute_occurence_map`` is called with gradients by loss.py:302)."""
        from .train import TrainRunner

        if not x.is_cuda:
            raise RuntimeError("protoasnet_amd models run on the GPU only; there is no CPU fallback")
        if x.shape[1] != 3:
            raise NotImplementedError("the training pass takes the reference's 3-channel clip; single-channel (grey) input is an "
                                      "eval-mode path (protoasnet_amd.data.DeviceClipPipeline)")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        runners = self.__dict__.setdefault("_train_runners", {})
        # the launch lists hold the parameters' device addresses: a model moved / cast / pruned since gets a fresh compilation
        key = (tuple(x.shape), x.dtype, self._dtype(), mode, hash(tuple((p.data_ptr(), p.requires_grad) for p in self.parameters())),
               tuple(self.prototype_shape), _lib.tuning_epoch())
        runner = runners.get(key)
        if runner is None:
            for stale in [k for k in runners if k[:4] == key[:4]]:
                del runners[stale]
            if len(self.add_on_layers.convs()) != 2 or len(self.occurrence_module.convs()) != 3:
                raise NotImplementedError("the training path supports the 2-conv add-on and 3-conv occurrence module of the shipped configs")
            runner = runners[key] = TrainRunner(self, x, mode)
        return runner(x)

    def _xproto(self, x: torch.Tensor, mode: int):
        if self.training:
            out = self._train_pass(x, mode)
            return (out[0], out[1], out[2], None) if mode == 0 else (None, None, out[0], None)
        self._guard(x)
        feat = self.cnn_backbone(x)
        dtype = self._dtype()
        rows, S, cbp = channels_last_rows(feat, dtype)
        N, Cb = feat.shape[0], feat.shape[1]
        P, D, K = self.num_prototypes, self.prototype_shape[1], self.num_classes
        a_convs, o_convs = self.add_on_layers.convs(), self.occurrence_module.convs()
        if len(a_convs) != 2 or len(o_convs) != 3:
            raise NotImplementedError(
                "the fused head supports the 2-conv add-on ('regular') and 3-conv occurrence module of the shipped configs"
            )
        d = XProtoDesc(N=N, S=S, Cb=Cb, Cbp=cbp, D=D, Dp=round_up(D, 8), Hd=D // 2, Hp=round_up(D // 2, 8), P=P,
                       Pp=round_up(P, 8), K=K, mode=mode)
        lib = _lib.lib()
        code = _lib.dtype_code(dtype)
        # the chained head (one launch for the five convs and the pooling, intermediates in LDS) where the shape allows: bf16, D = 256,
        # trunk channel stride <= 256 -- the X3D heads and R(2+1)D-18[:-3]; everything else takes the seven-launch path
        chain = bool(lib.pasn_xproto_chain_supported(ctypes.byref(d), code))
        (a1, a1b), (a2, a2b) = self.add_on_layers.packed(cbp, dtype, frag=chain)
        (o1, o1b), (o2, o2b), (o3, _) = self.occurrence_module.packed(cbp, dtype, frag=chain)
        ws_bytes = int(lib.pasn_xproto_chain_workspace_bytes(ctypes.byref(d)) if chain else lib.pasn_xproto_head_workspace_bytes(ctypes.byref(d), code))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=rows.device)
        dev = rows.device
        occ = torch.empty((N, P, S), dtype=torch.float32, device=dev)
        featx = sim = logits = None
        protos = fcw = None
        if mode == 0:
            featx = torch.empty((N, P, D), dtype=torch.float32, device=dev)
            sim = torch.empty((N, P), dtype=torch.float32, device=dev)
            logits = torch.empty((N, K), dtype=torch.float32, device=dev)
            protos, fcw = _f32(self.prototype_vectors), _f32(self.last_layer.weight)
        _lib.check(
            (lib.pasn_xproto_chain_fwd if chain else lib.pasn_xproto_head_fwd)(
                rows.data_ptr(), a1.data_ptr(), a1b.data_ptr(), a2.data_ptr(), a2b.data_ptr(), o1.data_ptr(), o1b.data_ptr(),
                o2.data_ptr(), o2b.data_ptr(), o3.data_ptr(), _lib.ptr(protos), _lib.ptr(fcw), occ.data_ptr(), _lib.ptr(featx),
                _lib.ptr(sim), _lib.ptr(logits), ws.data_ptr(), ctypes.byref(d), code, _lib.current_stream(),
            )
        )
        occ = occ.view((N, P, 1) + tuple(feat.shape[2:]))
        return logits, sim, occ, featx

    def forward(self, x: torch.Tensor):
        logits, sim, occ, _ = self._xproto(x, 0)
        return logits, sim, occ

    def compute_occurence_map(self, x: torch.Tensor) -> torch.Tensor:
        return self._xproto(x, 1)[2]

    # The reference's models call the next two on the TRUNK OUTPUT inside forward / push_forward (XProtoNet.py:55,75-85,
    # Video_XProtoNet.py:88,106-109); here the fused head computes the map, so nothing inside this package calls them.  They exist for
    # code written against the reference classes (a subclass, a notebook): the occurrence module alone on a logical (N,C,[T,]H,W)
    # feature tensor, through the same pointwise-conv launches as head A's add-on layers, no autograd graph.
    def _occurrence_logits(self, features: torch.Tensor) -> torch.Tensor:
        if not features.is_cuda:
            raise RuntimeError("protoasnet_amd models run on the GPU only; there is no CPU fallback")
        with torch.no_grad():
            return self.occurrence_module(features).float()  # (N, P, [T,] H, W)

    def get_occurence_map_absolute_val(self, x: torch.Tensor) -> torch.Tensor:
        return torch.abs(self._occurrence_logits(x)).unsqueeze(2)  # (N, P, 1, [T,] H, W)

    def get_occurence_map_softmaxed(self, x: torch.Tensor) -> torch.Tensor:
        """XProtoNet.py:75-80 (softmax over the flattened positions); the video class of the reference has no such method, the mixin
        gives it the same definition over (T, H, W)."""
        om = self._occurrence_logits(x)
        n, p = om.shape[:2]
        return self.om_softmax(om.reshape(n, p, -1)).reshape(om.shape).unsqueeze(2)

    def forward_pair(self, x: torch.Tensor, x_transformed: torch.Tensor):
        """``(forward(x), compute_occurence_map(x_transformed))`` of the reference's loss recipe (Video_XProtoNet_e2e.py:84 + loss.py:302) from
        ONE pass over the 2N clips.  Eval mode: running statistics, so a 2N batch gives every clip what two N-clip passes give.  Train mode:
        one compiled pass with TWO statistics groups (``train.TrainRunner`` mode 2) -- each half is normalised with its own batch statistics,
        the running estimates move twice, in order, and ``num_batches_tracked`` by 2: the state two passes leave.  Both halves are differentiable;
        the add-on path of the second half receives no gradient (its logits / similarities are not returned), as in the reference."""
        if x.shape != x_transformed.shape:
            raise ValueError("forward_pair takes the clips and their transformed copies: same shape")
        nb = x.shape[0]
        both = torch.cat([x, x_transformed])
        if self.training:
            logits, sim, occ = self._train_pass(both, 2)
        else:
            logits, sim, occ, _ = self._xproto(both, 0)
        return (logits[:nb], sim[:nb], occ[:nb]), occ[nb:]

    def push_forward(self, x: torch.Tensor):
        if self.training:
            raise RuntimeError("push_forward runs in eval mode (the reference pushes under model.eval(): push_abs_revision.py:210)")
        logits, sim, occ, feats = self._xproto(x, 0)
        return feats, 1 - sim, occ, logits


class XProtoNet(_XProtoHeadMixin, PPNet):
    """Image ProtoASNet / XProtoNet (reference src/models/XProtoNet.py:8-106) -- head B on a 2-D trunk."""

    def __init__(self, **kwargs):
        PPNet.__init__(self, **kwargs)
        self.cnn_backbone = self.features
        del self.features
        cin = self.get_cnn_backbone_out_channels(self.cnn_backbone)
        self.add_on_layers = PointwiseChain(*list(self.add_on_layers.children())[:-1])  # Sigmoid stripped, XProtoNet.py:17
        self._initialize_weights(self.add_on_layers)
        self.occurrence_module = _occurrence_module(nn.Conv2d, cin, self.prototype_shape[1], self.prototype_shape[0])
        self._initialize_weights(self.occurrence_module)
        self.last_layer = nn.Linear(self.num_prototypes, self.num_classes, bias=False)
        self.set_last_layer_incorrect_connection(incorrect_strength=0)
        self.om_softmax = nn.Softmax(dim=-1)  # XProtoNet.py:48-49: parameter-free attributes of the reference class
        self.cosine_similarity = nn.CosineSimilarity(dim=2)  # (the fused head computes the cosine itself, torch semantics: head_xproto.hip)

    def __repr__(self):
        return (
            "PPNet(\n\tcnn_backbone: {},\n\timg_size: {},\n\tprototype_shape: {},\n\tproto_layer_rf_info: {},\n"
            "\tnum_classes: {},\n\tepsilon: {}\n)"
        ).format(self.cnn_backbone, self.img_size, self.prototype_shape, self.proto_layer_rf_info, self.num_classes, self.epsilon)


class Video_XProtoNet(_XProtoHeadMixin, PPNet):  # noqa: N801 -- reference class name
    """Video ProtoASNet (reference src/models/Video_XProtoNet.py:8-130) -- head B on a 3-D trunk.

    Like the reference it bypasses ``PPNet.__init__`` and therefore has no ``epsilon`` /
    ``prototype_activation_function`` attributes.
    """

    def __init__(self, cnn_backbone, img_size, prototype_shape, proto_layer_rf_info, num_classes, init_weights=True, **kwargs):
        nn.Module.__init__(self)
        self.img_size = img_size
        self.prototype_shape = prototype_shape
        self.num_prototypes = prototype_shape[0]
        self.num_classes = num_classes
        self.prototype_class_identity = self.get_prototype_class_identity()
        self.proto_layer_rf_info = proto_layer_rf_info
        self.compute_dtype = None
        self.cnn_backbone = cnn_backbone
        cin = self.get_cnn_backbone_out_channels(self.cnn_backbone)
        depth = self.prototype_shape[1]
        self.add_on_layers = PointwiseChain(nn.Conv3d(cin, depth, kernel_size=1), nn.ReLU(), nn.Conv3d(depth, depth, kernel_size=1))
        self.occurrence_module = _occurrence_module(nn.Conv3d, cin, depth, self.prototype_shape[0])
        self.om_softmax = nn.Softmax(dim=-1)  # Video_XProtoNet.py:64-65
        self.cosine_similarity = nn.CosineSimilarity(dim=2)
        self.prototype_vectors = nn.Parameter(torch.rand(self.prototype_shape), requires_grad=True)
        self.ones = nn.Parameter(torch.ones(self.prototype_shape), requires_grad=False)
        self.last_layer = nn.Linear(self.num_prototypes, self.num_classes, bias=False)
        if init_weights:
            self._initialize_weights(self.add_on_layers)
            self._initialize_weights(self.occurrence_module)
            self.set_last_layer_incorrect_connection(incorrect_strength=0)

    def __repr__(self):
        return (
            "PPNet(\n\tcnn_backbone: {},\n\timg_size: {},\n\tprototype_shape: {},\n\tproto_layer_rf_info: {},\n"
            "\tnum_classes: {},\n)"
        ).format(self.cnn_backbone, self.img_size, self.prototype_shape, self.proto_layer_rf_info, self.num_classes)


# --------------------------------------------------------------------------------------------------
def _rf_info(features, img_size, prototype_shape):
    sizes, strides, paddings = features.conv_info()
    return compute_proto_layer_rf_info_v2(img_size, sizes, strides, paddings, prototype_shape[2])


def construct_PPNet(base_architecture, pretrained=True, img_size=224, prototype_shape=(2000, 512, 1, 1), num_classes=200,
                    prototype_activation_function="log", add_on_layers_type="bottleneck"):
    """reference src/models/ProtoPNet.py:343-370"""
    features = base_architecture_to_features[base_architecture](pretrained=pretrained)
    return PPNet(features=features, img_size=img_size, prototype_shape=prototype_shape,
                 proto_layer_rf_info=_rf_info(features, img_size, prototype_shape), num_classes=num_classes, init_weights=True,
                 prototype_activation_function=prototype_activation_function, add_on_layers_type=add_on_layers_type)


def construct_XProtoNet(base_architecture, pretrained=True, img_size=224, prototype_shape=(2000, 512, 1, 1), num_classes=200,
                        prototype_activation_function="log", add_on_layers_type="bottleneck"):
    """reference src/models/XProtoNet.py:132-159"""
    features = base_architecture_to_features[base_architecture](pretrained=pretrained)
    return XProtoNet(features=features, img_size=img_size, prototype_shape=prototype_shape,
                     proto_layer_rf_info=_rf_info(features, img_size, prototype_shape), num_classes=num_classes,
                     init_weights=True, prototype_activation_function=prototype_activation_function,
                     add_on_layers_type=add_on_layers_type)


def construct_Video_XProtoNet(base_architecture, pretrained=True, img_size=224, prototype_shape=(40, 256, 1, 1, 1), num_classes=4,
                              backbone_last_layer_num=-3):
    """reference src/models/Video_XProtoNet.py:154-178"""
    cnn_backbone = base_architecture_to_features[base_architecture](pretrained=pretrained, last_layer_num=backbone_last_layer_num)
    return Video_XProtoNet(cnn_backbone=cnn_backbone, img_size=img_size, prototype_shape=prototype_shape,
                           proto_layer_rf_info=None, num_classes=num_classes, init_weights=True)
