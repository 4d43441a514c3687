"""Oracle UNet2DModel forward (fp32 torch, functional over a diffusers-layout state dict).

Restates diffusers>=0.21 `UNet2DModel.forward` [3P, absent from /root/reference; SURVEY.md Appendix A.2] as the
reference configures it (ldm/train_unconditional.py:237-242, ldm/configs/RangeLDM.yaml:17-24) after the surgery of
ldm/utils.py:125-203 (every conv circular-W / zero-H; Downsample2D(padding=1)).
Block analogues that ARE in the reference and pin the arithmetic: sgm ResnetBlock (model.py:342-362),
AttnBlock (model.py:391-412), Model.forward (model.py:654-701).
"""
import torch
import torch.nn.functional as F
from rangeldm_amd.config import UNetConfig
from . import ops


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def resnet_block(sd, p, x, temb, groups, eps):
    """h=conv1(silu(GN(x))); h+=Linear(silu(temb)); h=conv2(silu(GN(h))); x=conv1x1(x) if Cin!=Cout; x+h."""
    h = ops.group_norm_silu(x, _t(sd, p + ".norm1.weight"), _t(sd, p + ".norm1.bias"), groups, eps)
    h = ops.circ_conv2d(h, _t(sd, p + ".conv1.weight"), _t(sd, p + ".conv1.bias"))
    if temb is not None and (p + ".time_emb_proj.weight") in sd:
        h = h + F.linear(F.silu(temb), _t(sd, p + ".time_emb_proj.weight"), _t(sd, p + ".time_emb_proj.bias"))[:, :, None, None]
    h = ops.group_norm_silu(h, _t(sd, p + ".norm2.weight"), _t(sd, p + ".norm2.bias"), groups, eps)
    h = ops.circ_conv2d(h, _t(sd, p + ".conv2.weight"), _t(sd, p + ".conv2.bias"))
    if (p + ".conv_shortcut.weight") in sd:
        x = ops.circ_conv2d(x, _t(sd, p + ".conv_shortcut.weight"), _t(sd, p + ".conv_shortcut.bias"), 1, 0)
    return x + h


def attention_block(sd, p, x, groups, eps, head_dim):
    """diffusers Attention + AttnProcessor2_0 with residual_connection=True, rescale_output_factor=1 [3P; A.2]."""
    B, C, W, H = x.shape
    y = F.group_norm(x, groups, _t(sd, p + ".group_norm.weight"), _t(sd, p + ".group_norm.bias"), eps)
    y = y.view(B, C, W * H).transpose(1, 2)                                  # (B, L, C), token = w*H + h
    q = F.linear(y, _t(sd, p + ".to_q.weight"), _t(sd, p + ".to_q.bias"))
    k = F.linear(y, _t(sd, p + ".to_k.weight"), _t(sd, p + ".to_k.bias"))
    v = F.linear(y, _t(sd, p + ".to_v.weight"), _t(sd, p + ".to_v.bias"))
    nh = C // head_dim
    q, k, v = (z.view(B, -1, nh, head_dim).transpose(1, 2) for z in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)                              # scale = head_dim ** -0.5
    o = o.transpose(1, 2).reshape(B, -1, C)
    o = F.linear(o, _t(sd, p + ".to_out.0.weight"), _t(sd, p + ".to_out.0.bias"))
    return o.transpose(1, 2).reshape(B, C, W, H) + x


def time_embedding(sd, cfg: UNetConfig, timestep, batch):
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.long)
    elif timestep.dim() == 0:
        timestep = timestep[None]
    t = timestep * torch.ones(batch, dtype=timestep.dtype)
    e = ops.timestep_embedding(t, cfg.block_out_channels[0], flip_sin_to_cos=cfg.flip_sin_to_cos, freq_shift=cfg.freq_shift)
    e = F.linear(e, _t(sd, "time_embedding.linear_1.weight"), _t(sd, "time_embedding.linear_1.bias"))
    return F.linear(F.silu(e), _t(sd, "time_embedding.linear_2.weight"), _t(sd, "time_embedding.linear_2.bias"))


@torch.no_grad()
def unet_forward(sd, cfg: UNetConfig, sample, timestep, taps=None):
    """sample (B, C_in, W, H) fp32, timestep int / 0-d / 1-d(B) -> (B, C_out, W, H).  `taps` (dict) records
    intermediate activations for block-level parity tests."""
    G, eps, hd = cfg.norm_num_groups, cfg.norm_eps, cfg.attention_head_dim
    boc = cfg.block_out_channels
    emb = time_embedding(sd, cfg, timestep, sample.shape[0])
    h = ops.circ_conv2d(sample, _t(sd, "conv_in.weight"), _t(sd, "conv_in.bias"))
    skips = [h]
    for i, btype in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, emb, G, eps)
            if btype == "AttnDownBlock2D":
                h = attention_block(sd, f"down_blocks.{i}.attentions.{j}", h, G, eps, hd)
            skips.append(h)
        if i != len(boc) - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            h = ops.downsample_unet(h, _t(sd, p + ".weight"), _t(sd, p + ".bias"))
            skips.append(h)
        if taps is not None:
            taps[f"down{i}"] = h
    h = resnet_block(sd, "mid_block.resnets.0", h, emb, G, eps)
    if cfg.add_attention:
        h = attention_block(sd, "mid_block.attentions.0", h, G, eps, hd)
    h = resnet_block(sd, "mid_block.resnets.1", h, emb, G, eps)
    if taps is not None:
        taps["mid"] = h
    for i, btype in enumerate(cfg.up_block_types):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, emb, G, eps)
            if btype == "AttnUpBlock2D":
                h = attention_block(sd, f"up_blocks.{i}.attentions.{j}", h, G, eps, hd)
        if i != len(boc) - 1:
            p = f"up_blocks.{i}.upsamplers.0.conv"
            h = ops.upsample_conv(h, _t(sd, p + ".weight"), _t(sd, p + ".bias"))
        if taps is not None:
            taps[f"up{i}"] = h
    assert not skips
    h = ops.group_norm_silu(h, _t(sd, "conv_norm_out.weight"), _t(sd, "conv_norm_out.bias"), G, eps)
    return ops.circ_conv2d(h, _t(sd, "conv_out.weight"), _t(sd, "conv_out.bias"))


class OracleUNet:
    """Duck-types the surface pipelines use: unet(x, t).sample, .config, .dtype, .device."""

    class _Out:
        def __init__(self, s):
            self.sample = s

    def __init__(self, cfg: UNetConfig, state_dict):
        self.config = cfg
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(v)).float() for k, v in state_dict.items()}
        self.dtype = torch.float32
        self.device = torch.device("cpu")

    def __call__(self, sample, timestep):
        return self._Out(unet_forward(self.sd, self.config, sample.float(), timestep))
