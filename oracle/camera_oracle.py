"""CPU restatement (plain PyTorch fp32) of the camera branch that fills HM-ViT's camera slot with the CVT encoder:
ResnetEncoder (opencood/models/backbones/resnet_ms.py:8-89), CrossViewModule (opencood/models/sub_modules/cvt_modules.py:
283-331) and the up-sampling NaiveDecoder (opencood/models/sub_modules/naive_decoder.py:8-92), assembled the way
FaxFusedTransformer assembles its camera branch (encoder -> cross-view module -> decoder, fax_fused_transformer.py:37-57).
TEST INFRASTRUCTURE ONLY.

Parity status: CrossViewAttention inside is pinned by g11 and NaiveDecoder's convolution stack by g8; the ResNet
BasicBlock / Bottleneck arithmetic lives in torchvision (absent here, version unpinned in the reference), so ResnetEncoder
and the Bottleneck layers follow torchvision's published block definitions (conv-bn-relu-conv-bn + identity, v1.5 stride
placement for Bottleneck is irrelevant here: all strides are 1) - PARITY UNPINNED for those."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from . import cvt_oracle as CO

_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}
_EXPANSION = {18: 1, 34: 1, 50: 4, 101: 4, 152: 4}


def _bn(x, sd, p, eps=1e-5):
    tr = CO.BN_TRAINING[0]          # training-mode restatement: batch statistics + running-buffer update (cvt_oracle.batch_statistics)
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"], tr, 0.1 if tr else 0.0, eps)


def resnet_features(images: Tensor, sd: Dict[str, Tensor], num_layers: int, prefix: str = "encoder") -> List[Tensor]:
    """images (n, 3, H, W) -> [layer1, layer2, layer3, layer4] outputs (torchvision ResNet with BasicBlock)."""
    p = prefix
    x = F.relu(_bn(F.conv2d(images, sd[f"{p}.conv1.weight"], stride=2, padding=3), sd, f"{p}.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nb in enumerate(_BLOCKS[num_layers]):
        for bi in range(nb):
            q = f"{p}.layer{li + 1}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            idt = x
            if _EXPANSION[num_layers] == 4:
                # torchvision Bottleneck (published definition, v1.5: the stride sits on the 3x3 convolution)
                y = F.relu(_bn(F.conv2d(x, sd[f"{q}.conv1.weight"]), sd, f"{q}.bn1"))
                y = F.relu(_bn(F.conv2d(y, sd[f"{q}.conv2.weight"], stride=stride, padding=1), sd, f"{q}.bn2"))
                y = _bn(F.conv2d(y, sd[f"{q}.conv3.weight"]), sd, f"{q}.bn3")
            else:
                y = F.relu(_bn(F.conv2d(x, sd[f"{q}.conv1.weight"], stride=stride, padding=1), sd, f"{q}.bn1"))
                y = _bn(F.conv2d(y, sd[f"{q}.conv2.weight"], padding=1), sd, f"{q}.bn2")
            if f"{q}.downsample.0.weight" in sd:
                idt = _bn(F.conv2d(x, sd[f"{q}.downsample.0.weight"], stride=stride), sd, f"{q}.downsample.1")
            x = F.relu(y + idt)
        outs.append(x)
    return outs


def resnet_encoder(input_images: Tensor, sd, params: dict, prefix: str = "encoder"):
    """ResnetEncoder.forward: (B, L, M, H, W, 3) -> features (B, L, M, C, h, w) for params['id_pick']."""
    b, l, m, h, w, c = input_images.shape
    x = input_images.reshape(b * l * m, h, w, c).permute(0, 3, 1, 2).contiguous()
    res = [f.reshape(b, l, m, *f.shape[1:]) for f in resnet_features(x, sd, params["num_layers"], prefix)]
    pick = params["id_pick"]
    return [res[i] for i in pick] if isinstance(pick, list) else res[pick]


def bottleneck(x: Tensor, sd, p: str) -> Tensor:
    """torchvision Bottleneck(c, c // 4) without downsample (cvt_modules.py:13)."""
    y = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv1.weight"]), sd, f"{p}.bn1"))
    y = F.relu(_bn(F.conv2d(y, sd[f"{p}.conv2.weight"], padding=1), sd, f"{p}.bn2"))
    y = _bn(F.conv2d(y, sd[f"{p}.conv3.weight"]), sd, f"{p}.bn3")
    return F.relu(y + x)


def cross_view_module(features: List[Tensor], intrinsic: Tensor, extrinsic: Tensor, sd, cfg: dict, prefix: str = "cvm") -> Tensor:
    """CrossViewModule.forward (cvt_modules.py:312-331).  features: list of (b, l, n, C, h, w); intrinsic (b, l, n, 3, 3);
    extrinsic (b, l, n, 4, 4).  Returns (b, l, dim, Hq, Wq)."""
    b, l, n = features[0].shape[:3]
    I_inv = intrinsic.reshape(b * l, n, 3, 3).inverse()
    E = extrinsic.reshape(b * l, n, 4, 4)
    be = cfg["bev_embedding"]
    grid = CO.bev_grid(be["bev_height"], be["bev_width"], be["h_meters"], be["w_meters"], be["offset"], len(be["decoder_blocks"]))
    x = sd[f"{prefix}.bev_embedding.learned_features"][None].repeat(b * l, 1, 1, 1)
    for i, (feature, num_layers) in enumerate(zip(features, cfg["middle"])):
        feat = feature.reshape(b * l, n, *feature.shape[3:])
        sub = {k[len(f"{prefix}.cross_views.{i}."):]: v for k, v in sd.items() if k.startswith(f"{prefix}.cross_views.{i}.")}
        x = CO.cross_view_attention(x, grid, feat, I_inv, E, sub, cfg["cross_view"])
        for j in range(num_layers):
            x = bottleneck(x, sd, f"{prefix}.layers.{i}.{j}")
    return x.reshape(b, l, *x.shape[1:])


def naive_decoder_up(x: Tensor, sd, prefix: str, num_layer: int) -> Tensor:
    """NaiveDecoder.forward with use_upsample=True (naive_decoder.py:63-92): (n, C, H, W) -> (n, C', H 2^L, W 2^L)."""
    k = 0
    for _ in range(num_layer):
        x = F.relu(_bn(F.conv2d(x, sd[f"{prefix}.decoder.{k}.weight"], sd[f"{prefix}.decoder.{k}.bias"], padding=1), sd, f"{prefix}.decoder.{k + 1}"))
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.relu(_bn(F.conv2d(x, sd[f"{prefix}.decoder.{k + 3}.weight"], sd[f"{prefix}.decoder.{k + 3}.bias"], padding=1), sd, f"{prefix}.decoder.{k + 4}"))
        k += 6
    return x


def camera_encoder(batch: dict, sd, cfg: dict) -> Tensor:
    """images (N, M, H, W, 3), intrinsic (N, M, 3, 3), extrinsic (N, M, 4, 4) -> BEV features (N, C, Hb, Wb)."""
    cam = batch["camera"][None]                                   # b = 1, l = N agents
    feats = resnet_encoder(cam, sd, cfg["encoder"], prefix="encoder.encoder")
    x = cross_view_module(feats, batch["intrinsic"][None], batch["extrinsic"][None], sd, cfg["cvm"])
    x = naive_decoder_up(x[0], sd, "decoder", cfg["decoder"]["num_layer"])
    return x


# ---- seeded configuration / parameters / inputs ----

def make_config(image=64, num_layers=18, dim=128, bev=32, small=True):
    """A reduced CVT camera encoder: `image` x `image` cameras, ResNet-`num_layers`, pyramid levels 1 and 3 (as the shipped
    id_pick), BEV queries (bev / 8)^2, decoder 128 -> [256, 256] with two x2 upsamplings (bev / 2 output)."""
    enc = {"num_layers": num_layers, "pretrained": False, "image_height": image, "image_width": image, "id_pick": [1, 3]}
    c1, c3 = 128, 512
    h1, h3 = image // 8, image // 32
    cvm = {"dim": dim, "middle": [2, 2], "backbone_output_shape": [[1, 1, 4, c1, h1, h1], [1, 1, 4, c3, h3, h3]],
           "bev_embedding": {"sigma": 1.0, "bev_height": bev, "bev_width": bev, "h_meters": 100.0, "w_meters": 100.0, "offset": 0.0,
                             "decoder_blocks": [128, 128, 64]},
           "cross_view": CO.make_config(heads=4, dim_head=32, image=image)}
    dec = {"input_dim": dim, "num_layer": 2, "num_ch_dec": [256, 256]}
    return {"encoder": enc, "cvm": cvm, "decoder": dec}


def random_state_dict(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    rs = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    sd: Dict[str, Tensor] = {}

    def conv(name, co, ci, k, bias=False):
        bnd = 1.0 / math.sqrt(ci * k * k)
        sd[f"{name}.weight"] = t(rs.uniform(-bnd, bnd, (co, ci, k, k)) * 1.7)
        if bias:
            sd[f"{name}.bias"] = t(rs.uniform(-bnd, bnd, co))

    def bn(name, c):
        sd[f"{name}.weight"] = t(1 + 0.1 * rs.standard_normal(c)); sd[f"{name}.bias"] = t(0.1 * rs.standard_normal(c))
        sd[f"{name}.running_mean"] = t(0.1 * rs.standard_normal(c)); sd[f"{name}.running_var"] = t(rs.uniform(0.6, 1.4, c))

    # ResNet (torchvision names)
    conv("encoder.encoder.conv1", 64, 3, 7); bn("encoder.encoder.bn1", 64)
    cin = 64
    for li, (nb, co) in enumerate(zip(_BLOCKS[cfg["encoder"]["num_layers"]], (64, 128, 256, 512))):
        for bi in range(nb):
            q = f"encoder.encoder.layer{li + 1}.{bi}"
            if _EXPANSION[cfg["encoder"]["num_layers"]] == 4:
                conv(f"{q}.conv1", co, cin, 1); bn(f"{q}.bn1", co)
                conv(f"{q}.conv2", co, co, 3); bn(f"{q}.bn2", co)
                conv(f"{q}.conv3", 4 * co, co, 1); bn(f"{q}.bn3", 4 * co)
                if bi == 0:
                    conv(f"{q}.downsample.0", 4 * co, cin, 1); bn(f"{q}.downsample.1", 4 * co)
                cin = 4 * co
                continue
            conv(f"{q}.conv1", co, cin, 3); bn(f"{q}.bn1", co)
            conv(f"{q}.conv2", co, co, 3); bn(f"{q}.bn2", co)
            if bi == 0 and (li > 0):
                conv(f"{q}.downsample.0", co, cin, 1); bn(f"{q}.downsample.1", co)
            cin = co
    # cross view module
    cvm = cfg["cvm"]
    dim = cvm["dim"]
    be = cvm["bev_embedding"]
    hq = be["bev_height"] // (2 ** len(be["decoder_blocks"]))
    wq = be["bev_width"] // (2 ** len(be["decoder_blocks"]))
    sd["cvm.bev_embedding.learned_features"] = t(be["sigma"] * 0.5 * rs.standard_normal((dim, hq, wq)))
    for i, shape in enumerate(cvm["backbone_output_shape"]):
        sub = CO.random_state_dict(shape[3], dim, cvm["cross_view"], seed=seed + 10 + i)
        sd.update({f"cvm.cross_views.{i}.{k}": v for k, v in sub.items()})
        for j in range(cvm["middle"][i]):
            q = f"cvm.layers.{i}.{j}"
            conv(f"{q}.conv1", dim // 4, dim, 1); bn(f"{q}.bn1", dim // 4)
            conv(f"{q}.conv2", dim // 4, dim // 4, 3); bn(f"{q}.bn2", dim // 4)
            conv(f"{q}.conv3", dim, dim // 4, 1); bn(f"{q}.bn3", dim)
    # decoder (NaiveDecoder module list order: layer num_layer-1 first)
    dec = cfg["decoder"]
    k = 0
    for i in range(dec["num_layer"] - 1, -1, -1):
        ci = dec["input_dim"] if i == dec["num_layer"] - 1 else dec["num_ch_dec"][i + 1]
        co = dec["num_ch_dec"][i]
        conv(f"decoder.decoder.{k}", co, ci, 3, bias=True); bn(f"decoder.decoder.{k + 1}", co)
        conv(f"decoder.decoder.{k + 3}", co, co, 3, bias=True); bn(f"decoder.decoder.{k + 4}", co)
        k += 6
    return sd


def synthetic_batch(n_agents: int, cfg: dict, seed: int = 0) -> dict:
    image = cfg["encoder"]["image_height"]
    rs = np.random.RandomState(seed)
    cam = torch.from_numpy(rs.standard_normal((n_agents, 4, image, image, 3)).astype(np.float32))
    _, _, I_inv, E_inv = CO.synthetic_inputs(n_agents, 4, 8, 2, 2, 8, 2, 2, seed=seed + 1, image=image)
    return {"camera": cam, "intrinsic": I_inv.inverse().contiguous(), "extrinsic": E_inv}
